/*
 * planner_emu.cpp -- CPU emulation of the GPU planner workgroup (TEST SUPPORT ONLY).
 * Runs nanorq_amd/csrc/planner_body.h -- the per-thread phase code of the HIP planner kernel -- with
 * the 256 threads of each phase executed sequentially and barriers as loop boundaries, in the phase
 * order of planner_seq.h.  Not part of the product.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

static uint32_t g_gj_block_min = 4; /* words of the GF(2) matrix per thread from which the Gauss-Jordan runs in panels (kernel: 4) */
#define PL_GJ_BLOCK_MIN g_gj_block_min
extern "C" void emu_plan_set_gj_block_min(uint32_t v) { g_gj_block_min = v; }
static uint32_t g_gj_wave = 1; /* the 32 columns of a panel in one phase (pl_gjp_wave; kernel: on) */
#define PL_GJ_WAVE g_gj_wave
extern "C" void emu_plan_set_gj_wave(uint32_t v) { g_gj_wave = v; }
#include "../../nanorq_amd/csrc/planner_body.h"

/* tests/test_planner_emu.py::test_plan_check_finds_damaged_books: damage done to the peel's books right before the plan check
 * (planner_seq.h PL_SABOTAGE_HOOK) -- what a lost race of the device-only peeling forms would leave behind */
static uint32_t g_sabotage = 0;
extern "C" void emu_plan_set_sabotage(uint32_t kind) { g_sabotage = kind; }
static void emu_sabotage(PlanCtx &c) {
  pl_shared *sh = c.sh;
  const uint32_t np = sh->npiv;
  if (!g_sabotage || np < 8u) return;
  switch (g_sabotage) {
    case 1: c.pivslot[np / 2u] = c.pivslot[np / 2u + 1u]; break;                    /* two pivots share a row */
    case 2: { const uint16_t t = c.pivcol[1]; c.pivcol[1] = c.pivcol[2]; c.pivcol[2] = t; } break; /* pivot columns swapped */
    case 3: { /* a pivot row claimed too early: the deepest pivot pretends to be on level 0 */
      uint32_t best = 0, bl = 0;
      for (uint32_t k = 0; k < np; k++) { const uint32_t l = c.rowinfo[c.pivslot[k]] & PL_LEVEL_MASK; if (l >= bl) { bl = l; best = k; } }
      c.rowinfo[c.pivslot[best]] &= ~PL_LEVEL_MASK;
    } break;
    case 4: c.colinfo[c.pivcol[np - 1u]] = 0u; break;                                 /* a column left in V */
    case 5: { const uint32_t u = c.p.L - np; if (u >= 2u) c.colinfo[c.ucol[1]] = (PL_ST_INACT << 30) | 0u; } break; /* two inactive columns, one W bit */
    case 6: { /* a pivot row that does not hold its pivot column: two pivots trade rows */
      const uint16_t t = c.pivslot[np - 1u]; c.pivslot[np - 1u] = c.pivslot[np - 2u]; c.pivslot[np - 2u] = t;
    } break;
    default: break;
  }
}
#define PL_SABOTAGE_HOOK emu_sabotage(c)
/* trace of the peel for tools/peel_trace.py: frontier width (rows with one V column) and open columns of every round */
static std::vector<uint32_t> g_trace;
static uint32_t g_trace_on = 0;
extern "C" void emu_plan_trace(uint32_t on) { g_trace_on = on; g_trace.clear(); }
extern "C" uint32_t emu_plan_trace_read(uint32_t *out, uint32_t cap) {
  const uint32_t n = (uint32_t)g_trace.size() < cap ? (uint32_t)g_trace.size() : cap;
  for (uint32_t i = 0; i < n; i++) out[i] = g_trace[i];
  return (uint32_t)g_trace.size();
}
#define PL_TRACE_ROUND(nq, nv) do { if (g_trace_on) { g_trace.push_back(nq); g_trace.push_back(nv); } } while (0)

extern "C" uint32_t emu_plan_arena_bound(uint32_t K, const uint8_t *kc, uint32_t overhead_cap, uint32_t nlost_cap) {
  const nrq_kconst_hdr *kh = reinterpret_cast<const nrq_kconst_hdr *>(kc);
  (void)K;
  return pl_arena_bound(kh->L, kh->L + overhead_cap, kh->P + 768u, kh->nnz + (nlost_cap + overhead_cap) * 40u, nlost_cap);
}

/* PL_WFAST_RUN on the CPU: the op-stream rows in the order the kernel's wave 0 issues them (step q applies row
 * q-NRQ_PIPE, then reads the sources of row q) on the wb-byte slot image at the start of the dynamic region. */
static void emu_wfast_run(PlanCtx &c, uint32_t wb) {
  const uint32_t wpl = wb / 4u, nrows = pl_wfast_rows(c), P = NRQ_PIPE;
  uint32_t *img = reinterpret_cast<uint32_t *>(c.lds_dyn);
  const uint32_t *ops = reinterpret_cast<const uint32_t *>(c.arena + c.sh->off_ops);
  std::vector<uint32_t> v((size_t)(P + 1) * NRQ_ROW * wpl);
  for (uint32_t q = 0; q < nrows + P; q++) {
    if (q >= P) {
      const uint32_t *vr = &v[(size_t)((q - P) % (P + 1)) * NRQ_ROW * wpl];
      for (uint32_t l = 0; l < NRQ_ROW; l++)
        for (uint32_t k = 0; k < wpl; k++) img[(size_t)(ops[NRQ_OP_INDEX(q - P, l)] & 0xFFFFu) * wpl + k] ^= vr[l * wpl + k];
    }
    if (q < nrows) {
      uint32_t *vr = &v[(size_t)(q % (P + 1)) * NRQ_ROW * wpl];
      for (uint32_t l = 0; l < NRQ_ROW; l++)
        for (uint32_t k = 0; k < wpl; k++) vr[l * wpl + k] = img[(size_t)(ops[NRQ_OP_INDEX(q, l)] >> 16) * wpl + k];
    }
  }
}

/* PL_MHREV_RUN on the CPU: the op stream transposed, rows from the last to the first in the order the kernel's wave issues them
 * (step for row q: apply row q + NRQ_PIPE -- slot(src) ^= what was read from slot(dst) --, then read row q's dst slots) */
static void emu_mhrev_run(PlanCtx &c, uint32_t wb) {
  const uint32_t wpl = wb / 4u, nrows = pl_wfast_rows(c), P = NRQ_PIPE;
  uint32_t *img = reinterpret_cast<uint32_t *>(c.lds_dyn);
  const uint32_t *ops = reinterpret_cast<const uint32_t *>(c.arena + c.sh->off_ops);
  std::vector<uint32_t> v((size_t)(P + 1) * NRQ_ROW * wpl);
  for (int64_t q = (int64_t)nrows - 1; q >= -(int64_t)P; q--) {
    if (q + P < (int64_t)nrows) {
      const uint32_t r = (uint32_t)(q + P);
      const uint32_t *vr = &v[(size_t)(r % (P + 1)) * NRQ_ROW * wpl];
      for (uint32_t l = 0; l < NRQ_ROW; l++)
        for (uint32_t k = 0; k < wpl; k++) img[(size_t)(ops[NRQ_OP_INDEX(r, l)] >> 16) * wpl + k] ^= vr[l * wpl + k];
    }
    if (q >= 0) {
      uint32_t *vr = &v[(size_t)((uint32_t)q % (P + 1)) * NRQ_ROW * wpl];
      for (uint32_t l = 0; l < NRQ_ROW; l++)
        for (uint32_t k = 0; k < wpl; k++) vr[l * wpl + k] = img[(size_t)(ops[NRQ_OP_INDEX((uint32_t)q, l)] & 0xFFFFu) * wpl + k];
    }
  }
}

/* returns 0 and fills arena (status in its header); job_out receives the solve job (host pointers) */
/* nrq_wpass_kernel on the CPU: every 2-byte strip of the W rows through the op stream, rows applied in stream order
 * (what fwd_rows<2> computes: a row never reads what it or its predecessor writes) */
static void emu_wpass_strips(PlanCtx &c) {
  pl_shared *sh = c.sh;
  const uint32_t M = sh->M, wpr = sh->wpr, nrows = sh->spare_base;
  const uint32_t *ops = reinterpret_cast<const uint32_t *>(c.arena + sh->off_ops);
  uint16_t *rows16 = reinterpret_cast<uint16_t *>(c.wrows);
  std::vector<uint16_t> img(M + NRQ_SCRATCH);
  for (uint32_t strip = 0; strip < wpr * 2u; strip++) {
    for (uint32_t e = 0; e < M + NRQ_SCRATCH; e++) img[e] = e >= NRQ_SCRATCH ? rows16[(size_t)(e - NRQ_SCRATCH) * wpr * 2u + strip] : 0;
    for (uint32_t r = 0; r < nrows; r++) {
      uint16_t v[NRQ_ROW];
      for (uint32_t l = 0; l < NRQ_ROW; l++) v[l] = img[ops[NRQ_OP_INDEX(r, l)] >> 16];
      for (uint32_t l = 0; l < NRQ_ROW; l++) img[ops[NRQ_OP_INDEX(r, l)] & 0xFFFFu] ^= v[l];
    }
    for (uint32_t r = 0; r < M; r++) rows16[(size_t)r * wpr * 2u + strip] = img[r + NRQ_SCRATCH];
  }
}

static uint32_t g_mode = 0; /* nrq_planjob::mode of the next emu_plan call (1 = encode plan) */
static uint32_t g_qcap = PL_QCAP, g_lowcap = PL_LOWCAP; /* capacities of the arrays behind pl_shared (small blocks get small ones) */
extern "C" void emu_plan_set_caps(uint32_t qcap, uint32_t lowcap) { g_qcap = qcap ? qcap : PL_QCAP; g_lowcap = lowcap ? lowcap : PL_LOWCAP; }
static uint32_t g_split = 0; /* run the phase sequence in its two parts (what big blocks do on the GPU) */
static uint32_t g_nopk = 0;  /* keep the peeling state in the workspace alone (no compact copy in LDS) when it does not fit the LDS */
static uint32_t g_went = 0;  /* with g_split: part 1 cut once more, the entry pass by two "workgroups" in between (nrq_wentry_kernel) */
extern "C" void emu_plan_set_mode(uint32_t mode) { g_mode = mode & 0xFFu; g_split = (mode >> 8) & 1u; g_nopk = (mode >> 9) & 1u; g_went = (mode >> 10) & 1u; }
extern "C" int emu_plan(uint32_t K, uint32_t Kp_hint, const uint8_t *kc, const uint32_t *lost, uint32_t nlost,
                        const uint32_t *rep_esi, uint32_t nrep, uint32_t nrep_avail, uint8_t *arena,
                        uint32_t arena_cap, uint32_t lds_dyn_bytes, nrq_job *job_out) {
  rq_params p;
  if (!rq_params_init(Kp_hint ? Kp_hint : K, &p)) return -1;
  p.K = K;
  const nrq_kconst_hdr *kh = reinterpret_cast<const nrq_kconst_hdr *>(kc);
  if (nrep_avail < nrep) nrep_avail = nrep;
  const uint32_t ohcap = nrep_avail > nlost ? nrep_avail - nlost : 0;
  const uint32_t Mcap = kh->L + ohcap + PL_EXTRA_ROWS + 8, npcap = nrep_avail + PL_EXTRA_ROWS + 8, ucap = kh->P + 768u;
  pl_work_layout wl = pl_work_plan(kh->L, Mcap, npcap, ucap, kh->nnz + npcap * PL_PATCH_STRIDE);
  std::vector<uint8_t> work(wl.total + 64, 0xCC), dyn(lds_dyn_bytes + 64, 0xDD);
  const uint32_t shb = pl_shared_bytes(g_qcap, g_lowcap, PL_NT);
  std::vector<uint8_t> shmem(shb + 64, 0xEE); /* pl_shared and the arrays behind it */
  pl_shared *sh = reinterpret_cast<pl_shared *>(shmem.data());
  nrq_planjob job;
  memset(&job, 0, sizeof(job));
  job.lost = (uint64_t)(uintptr_t)lost;
  job.rep_esi = (uint64_t)(uintptr_t)rep_esi;
  job.work = (uint64_t)(uintptr_t)work.data();
  job.arena = (uint64_t)(uintptr_t)arena;
  job.nlost = nlost; job.nrep = nrep; job.arena_cap = arena_cap; job.nrep_avail = nrep_avail;
  job.mode = g_mode | (g_split << 8) | ((g_split && g_went) ? 0x200u : 0u);
  PlanCtx c;
  pl_ctx_setup(c, p, kc, job, sh, lds_dyn_bytes ? dyn.data() : nullptr, lds_dyn_bytes, Mcap, npcap, ucap, job_out, g_qcap, g_lowcap, PL_NT);
  if (g_nopk) c.pk_cnt = c.pk_un = c.pk_pa = c.pk_vb = nullptr;
#define PL_PHASE(fn) do { for (uint32_t t_ = 0; t_ < PL_NT; t_++) fn<1>(c, t_, PL_NT); } while (0)
#define PL_PHASE1(fn, a) do { for (uint32_t t_ = 0; t_ < PL_NT; t_++) fn<1>(c, (a), t_, PL_NT); } while (0)
#define PL_PHASE1_CLAIM(fn, a) PL_PHASE1(fn, a)
#define PL_WFAST_RUN(wb) emu_wfast_run(c, (wb))
#define PL_MHREV_RUN(wb) emu_mhrev_run(c, (wb))
#define PL_SEG g_seg
#define PL_STEER_SYNC do { } while (0)
#define PL_NT_ PL_NT
#define PL_Z 1u
  const uint32_t segs_[3] = {(g_split && g_went) ? 3u : 1u, (g_split && g_went) ? 4u : 2u, 2u};
  for (uint32_t pass_ = 0; pass_ < (!g_split ? 1u : g_went ? 3u : 2u); pass_++) {
    const uint32_t g_seg = g_split ? segs_[pass_] : 0u;
    c.cls_glob = g_seg == 3u ? reinterpret_cast<uint32_t *>(c.work + c.wl.cls_g) : nullptr;
    if (g_seg == 4u) { /* nrq_wentry_kernel: the entry pass dealt out over two workgroups, counters in the workspace */
      PL_PHASE(pl_sh_restore);
      if (sh->status == 0 && sh->nV == 0) {
        c.cls_glob = reinterpret_cast<uint32_t *>(c.work + c.wl.cls_g);
        c.nrec_ptr = &c.wentry[0];
        c.own_ptr = &c.wentry[3];
        for (uint32_t part = 0; part < 2u; part++)
          for (uint32_t t_ = 0; t_ < PL_NT; t_++) pl_w_init_part<1>(c, part, 2u, t_, PL_NT);
        PL_PHASE(pl_wentry_report);
        c.cls_glob = nullptr;
        c.nrec_ptr = &sh->nrec;
        c.own_ptr = &sh->own_hits;
      }
      memset(sh, 0xEE, shb);
      PL_PHASE(pl_sh_restore);
    }
    if (g_seg == 2u) { /* what the helper kernels do between the parts */
      PL_PHASE(pl_sh_restore);
      if (sh->status == 0 && sh->nV == 0) {
        emu_wpass_strips(c); /* nrq_wpass_kernel: the op stream on 2-byte strips of the W rows */
        const uint32_t ntiles = (sh->npiv + PL_MH_TILE - 1u) / PL_MH_TILE;
        for (uint32_t part = 0; part < 2u; part++) {
          if (part == 0) PL_PHASE(pl_mh_init); else PL_PHASE(pl_mh_part_zero);
          for (uint32_t tl_ = part; tl_ < ntiles; tl_ += 2u) { PL_PHASE1(pl_mh_load, tl_); PL_PHASE1(pl_mh_acc, tl_); }
          PL_PHASE(pl_mh_part_flush);
        }
      }
    }
#include "../../nanorq_amd/csrc/planner_seq.h"
    if (g_seg == 1u || g_seg == 4u) PL_PHASE(pl_mh_ext_clear);
    if (g_seg == 1u || g_seg == 3u || g_seg == 4u) { PL_PHASE(pl_sh_save); memset(sh, 0xEE, shb); }
  }
#undef PL_SEG
#undef PL_STEER_SYNC
#undef PL_NT_
  if (g_split) pl_wt_fill(arena, reinterpret_cast<const uint32_t *>(work.data() + wl.wrows), 0u, 1u); /* = nrq_wt_kernel */
#undef PL_PHASE
#undef PL_PHASE1
#undef PL_PHASE1_CLAIM
#undef PL_WFAST_RUN
#undef PL_MHREV_RUN
  return 0;
}
