/*
 * solve_emu.cpp -- CPU emulation of the gfx950 solve workgroup (TEST SUPPORT ONLY).
 *
 * Compiles nanorq_amd/csrc/solve_body.h -- the exact per-thread phase functions the HIP kernel
 * runs -- with g++ and executes the 256 threads of each phase sequentially, with the workgroup
 * barriers of nrq_solve_kernel turned into loop boundaries.  It lets the non-GPU test tier
 * exercise strip indexing, the packed GF(256) arithmetic, the HDPC Horner evaluation and the
 * chunk/sync schedule against the oracle.  It is not part of the product and is never shipped in
 * libnanorq_hip.so.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

static uint32_t g_dense_shared_min_nt = 512; /* workgroup size from which the dense fold shares the multiples (kernel: 512) */
#define NRQ_DENSE_SHARED_MIN_NT g_dense_shared_min_nt
#include "../../nanorq_amd/csrc/solve_body.h"
extern "C" void emu_set_dense_shared_min_nt(uint32_t v) { g_dense_shared_min_nt = v; }
static int g_hdpc_regs = 0; /* 1: the HDPC phase in the form of the big workgroup (register accumulators) */
extern "C" void emu_set_hdpc_regs(int v) { g_hdpc_regs = v; }

/* The forward passes in the order the kernel's wave 0 issues them (plan.h): step q applies row q-NRQ_PIPE,
 * then reads the sources of row q -- so a plan that puts dependent rows closer than NRQ_PIPE rows apart
 * produces wrong symbols here, exactly as it would on the GPU. */
template <int WB> static bool emu_forward(const StripCtx<WB> &c) {
  const uint32_t *ops = c.template arr<uint32_t>(c.h->off_ops);
  const uint32_t nrows = c.h->nrows, P = NRQ_PIPE;
  if (c.h->pipe != NRQ_PIPE) return false;
  std::vector<SV<WB>> v((size_t)(P + 1) * NRQ_ROW);
  for (uint32_t q = 0; q < nrows + P; q++) {
    if (q >= P) {
      const SV<WB> *vr = &v[(size_t)((q - P) % (P + 1)) * NRQ_ROW];
      for (uint32_t l = 0; l < NRQ_ROW; l++) ph_row_apply<WB>(c, ops[NRQ_OP_INDEX(q - P, l)], vr[l]);
    }
    if (q < nrows) {
      SV<WB> *vr = &v[(size_t)(q % (P + 1)) * NRQ_ROW];
      for (uint32_t l = 0; l < NRQ_ROW; l++) vr[l] = ph_row_read<WB>(c, ops[NRQ_OP_INDEX(q, l)]);
    }
  }
  return true;
}

template <int WB> static int run_strip(const nrq_job &job, uint32_t T, uint32_t strip, const uint8_t *kc, std::vector<uint8_t> &ostage) {
  const uint32_t NT = 256;
  StripCtx<WB> c;
  c.job = &job;
  c.plan = reinterpret_cast<const uint8_t *>(job.plan);
  c.h = reinterpret_cast<const nrq_plan_hdr *>(c.plan);
  if (c.h->status) return 0;
  c.kc = kc;
  c.lay = nrq_lds_plan(c.h, WB);
  std::vector<uint8_t> lds(c.lay.total + 64, 0xA5); /* garbage-filled like real LDS */
  c.lds = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~(uintptr_t)15);
  c.T = T;
  c.strip = strip;
  uint32_t rem = T - strip * WB;
  c.valid = rem < (uint32_t)WB ? rem : (uint32_t)WB;
#define PHASE(fn) for (uint32_t t = 0; t < NT; t++) fn<WB>(c, t, NT)
  { /* the way the persistent kernel fills the image: the gathering threads bring the whole line group of this
     * strip into the per-strip staging buffers, then all threads copy this strip's buffer into the image */
    constexpr uint32_t SPL = nrq_group_strips(WB);
    const size_t stride = ((size_t)c.h->M * WB + 255u) & ~(size_t)255u;
    std::vector<uint8_t> stage(stride * SPL + 64, 0x5A);
    GroupSrc<WB> g;
    g.rowsrc = gptr<uint32_t>(c.job->rowsrc); g.src = gptr<uint8_t>(c.job->src); g.rep = gptr<uint8_t>(c.job->rep);
    g.M = c.h->M; g.T = T; g.strip0 = (strip / SPL) * SPL; g.nstrips = (T + WB - 1) / WB; g.lsub = __builtin_ctz(SPL);
    const uint32_t np = NT - NRQ_ROW, units = g.M * SPL, half = units / 2;
    for (uint32_t p = 0; p < np; p++) pf_gather<WB>(g, stage.data(), stride, 0, half, p, np);   /* in two portions, */
    for (uint32_t p = 0; p < np; p++) pf_gather<WB, 1, true>(g, stage.data(), stride, half, units, p, np); /* as the kernel does */
    for (uint32_t t = 0; t < NT; t++) pf_commit<WB>(c, stage.data() + (size_t)(strip % SPL) * stride, 0u, t, NT);
    PHASE(ph_clear);
  }
  if (!emu_forward<WB>(c)) return -7;
  if (g_hdpc_regs) { for (uint32_t t = 0; t < NT; t++) ph_hdpc<WB, 1, true>(c, t, NT); } else PHASE(ph_hdpc);
  PHASE(ph_hdpc_reduce);
  for (uint32_t w0 = 0; w0 < c.h->lpr; w0 += low_table_words<WB>(c)) {
    for (uint32_t t = 0; t < NT; t++) ph_low_tables<WB>(c, w0, t, NT);
    for (uint32_t t = 0; t < NT; t++) { uint32_t cb[NRQ_COMBINE_WU]; ph_combine_fetch<WB>(c, w0, t, NT, cb); ph_combine<WB>(c, w0, t, NT, cb); }
  }
  if (c.h->lpr) PHASE(ph_clear_x);
  for (uint32_t t = 0; t < NT; t++) ph_dense_fold<WB, 1, 8>(c, t, NT); /* (the form of the 256-thread workgroup) */
  if (dense_fold_shared(NT)) PHASE(ph_hdpc_reduce);
  for (uint32_t t = 0; t < NT; t++) ph_dense_free<WB, 1, true>(c, t, NT);
  for (uint32_t t = 0; t < NT; t++) ph_dense_cu<WB, 1, true>(c, t, NT);
  PHASE(ph_tables);
  PHASE(ph_backsub);
  PHASE(ph_park);
  { /* results: staged per strip, then scattered to the symbol rows a line group at a time */
    constexpr uint32_t SPL = nrq_group_strips(WB);
    const uint32_t nstrips = (T + WB - 1) / WB, ne = out_elems<WB>(c.job, c.h);
    const size_t ostride = ((size_t)ne * WB + 255u) & ~(size_t)255u;
    if (strip % SPL == 0) ostage.assign(ostride * SPL + 64, 0x3C);
    for (uint32_t t = 0; t < NT; t++) ph_store<WB>(c, ostage.data() + (size_t)(strip % SPL) * ostride, t, NT);
    if (strip % SPL == SPL - 1 || strip + 1 == nstrips) {
      GroupDst<WB> g;
      g.inter = gptr_w<uint8_t>(c.job->inter); g.out = gptr_w<uint8_t>(c.job->out); g.orow = gptr<uint32_t>(c.job->out_row);
      g.ni = c.job->inter ? c.h->L : 0u; g.nout = c.job->nout; g.T = T; g.strip0 = (strip / SPL) * SPL; g.nstrips = nstrips; g.lsub = __builtin_ctz(SPL);
      const uint32_t np = NT - NRQ_ROW, units = ne * SPL, cut = units / 3;
      for (uint32_t p = 0; p < np; p++) pf_scatter<WB>(g, ostage.data(), ostride, 0, cut, p, np);
      for (uint32_t p = 0; p < np; p++) pf_scatter<WB, 1, true>(g, ostage.data(), ostride, cut, units, p, np);
    }
  }
#undef PHASE
  return 1;
}

extern "C" uint32_t emu_lds_bytes(const uint8_t *plan, uint32_t wb) {
  return nrq_lds_plan(reinterpret_cast<const nrq_plan_hdr *>(plan), wb).total;
}

/* every pointer in `job` is a host pointer here */
extern "C" int emu_solve(const nrq_job *job, uint32_t T, uint32_t wb, const uint8_t *kc) {
  const uint32_t nstrips = (T + wb - 1) / wb;
  int r = 1;
  std::vector<uint8_t> ostage;
  for (uint32_t s = 0; s < nstrips && r; s++) {
    switch (wb) {
      case 16: r = run_strip<16>(*job, T, s, kc, ostage); break;
      case 12: r = run_strip<12>(*job, T, s, kc, ostage); break;
      case 8: r = run_strip<8>(*job, T, s, kc, ostage); break;
      case 4: r = run_strip<4>(*job, T, s, kc, ostage); break;
      case 2: r = run_strip<2>(*job, T, s, kc, ostage); break;
      default: return -1;
    }
  }
  return r;
}
