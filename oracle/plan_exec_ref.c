/*
 * plan_exec_ref.c -- CPU reference executor for device plans (nanorq_amd/csrc/plan.h).
 *
 * TEST INFRASTRUCTURE ONLY (part of the oracle library).  It applies a plan to whole symbol
 * rows with straightforward loops and table-based GF(256) arithmetic, independently of the HIP
 * kernels' strip layout, xtime/Horner tricks and chunk scheduling.  tests/ use it to check a
 * planner on the CPU against the reference-equivalent oracle (rq_oracle.c) and as the executable
 * specification of what solve.hip must compute from the same plan.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../nanorq_amd/csrc/plan.h"

void orc_gf_tables(uint8_t *exp510, uint8_t *log256, uint8_t *inv256);
int orc_hdpc(uint32_t K, uint8_t *out);

static uint8_t E_[510], L_[256], I_[256];

static void xor_row(uint8_t *d, const uint8_t *s, size_t n) { for (size_t k = 0; k < n; k++) d[k] ^= s[k]; }
static void fma_row(uint8_t *d, const uint8_t *s, size_t n, uint8_t b) {
  if (!b) return;
  for (size_t k = 0; k < n; k++)
    if (s[k]) d[k] ^= E_[L_[b] + L_[s[k]]];
}

/* Din: M x T (pitch T) with the received/source symbols already placed in their rows, zero
 * elsewhere.  C out: L x T.  returns 1 ok, 0 if the plan is marked singular, <0 on a bad plan. */
int orc_plan_exec(const uint8_t *plan, uint8_t *Din, uint32_t T, uint8_t *C) {
  const nrq_plan_hdr *h = (const nrq_plan_hdr *)plan;
  if (h->magic != NRQ_PLAN_MAGIC) return -1;
  if (h->status) return 0;
  orc_gf_tables(E_, L_, I_);
  /* work matrix: the M slots plus the r2 scratch rows E_p the op stream accumulates into */
  uint8_t *D = (uint8_t *)calloc((size_t)(h->M + h->r2 + 1) * T, 1);
  memcpy(D, Din, (size_t)h->M * T);
  const uint32_t *ops = (const uint32_t *)(plan + h->off_ops);
  const uint16_t *pivslot = (const uint16_t *)(plan + h->off_pivslot);
  const uint16_t *pivcol = (const uint16_t *)(plan + h->off_pivcol);
  const uint32_t *wt = (const uint32_t *)(plan + h->off_wt);
  const uint16_t *pivx = (const uint16_t *)(plan + h->off_pivx);
  const uint32_t *fbits = (const uint32_t *)(plan + h->off_fbits);
  const uint8_t *mh = plan + h->off_mh;
  const uint16_t *freex = (const uint16_t *)(plan + h->off_freex);
  const uint8_t *hinv = plan + h->off_hinv;
  const uint16_t *colslot = (const uint16_t *)(plan + h->off_colslot);
  const uint16_t *uslot = (const uint16_t *)(plan + h->off_uslot);
  const uint32_t H = h->H, n_hd = h->Kp + h->S;
#define ROW(r) (D + (size_t)(r) * T)
  /* 1+2: forward substitution through X, then the leftover rows */
  size_t nops = (size_t)h->nrows * NRQ_ROW;
  for (size_t e = 0; e < nops; e++) {
    if (NRQ_OP_IS_NOP(ops[e])) continue;
    xor_row(ROW(NRQ_OP_DST(ops[e])), ROW(NRQ_OP_SRC(ops[e])), T);
  }
  /* 3: HDPC right-hand sides */
  uint8_t *G = (uint8_t *)malloc((size_t)H * n_hd);
  orc_hdpc(h->K, G);
  for (uint32_t k = 0; k < h->npiv; k++)
    for (uint32_t q = 0; q < H; q++) fma_row(ROW(h->S + q), ROW(pivslot[k]), T, G[(size_t)q * n_hd + pivcol[k]]);
  free(G);
  /* 4: the binary combinations E_p (rows M+p) = XOR of the leftover rows the bit matrix names (lpr == 0: they were
   * ops of the stream, accumulated into rows M+p above) */
  uint8_t *E = D + (size_t)h->M * T;
  if (h->lpr) {
    const uint32_t *augt = (const uint32_t *)(plan + h->off_augt);
    const uint16_t *lowslot = (const uint16_t *)(plan + h->off_lowslot);
    for (uint32_t p = 0; p < h->r2; p++)
      for (uint32_t j = 0; j < h->nlow; j++)
        if ((augt[(size_t)(j >> 5) * h->aug_stride + p] >> (j & 31)) & 1u) xor_row(E + (size_t)p * T, ROW(lowslot[j]), T);
  }
  /* 5: fold the resolved columns out of the HDPC rows */
  for (uint32_t q = 0; q < H; q++)
    for (uint32_t p = 0; p < h->r2; p++) fma_row(ROW(h->S + q), E + (size_t)p * T, T, mh[(size_t)q * h->r2 + p]);
  /* 6: free columns */
  uint8_t *Cu = (uint8_t *)calloc((size_t)(h->u ? h->u : 1) * T, 1);
  for (uint32_t f = 0; f < h->nfree; f++)
    for (uint32_t q = 0; q < H; q++) fma_row(Cu + (size_t)freex[f] * T, ROW(h->S + q), T, hinv[(size_t)f * H + q]);
  /* 7: columns resolved by binary rows */
  for (uint32_t p = 0; p < h->r2; p++) {
    uint8_t *dst = Cu + (size_t)pivx[p] * T;
    memcpy(dst, E + (size_t)p * T, T);
    for (uint32_t f = 0; f < h->nfree; f++)
      if ((fbits[p] >> f) & 1u) xor_row(dst, Cu + (size_t)freex[f] * T, T);
  }
  /* 8: back substitution with the fill-in matrix W */
  for (uint32_t k = 0; k < h->npiv; k++)
    for (uint32_t x = 0; x < h->u; x++)
      if ((wt[(size_t)(x >> 5) * h->npiv_pad + k] >> (x & 31)) & 1u) xor_row(ROW(pivslot[k]), Cu + (size_t)x * T, T);
  /* 9+10: homes, gather */
  for (uint32_t x = 0; x < h->u; x++) memcpy(ROW(uslot[x]), Cu + (size_t)x * T, T);
  for (uint32_t c = 0; c < h->L; c++) memcpy(C + (size_t)c * T, ROW(colslot[c]), T);
  free(Cu); free(D);
#undef ROW
  return 1;
}
