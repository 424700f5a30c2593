/*
 * planner_body.h -- the symbolic stage ON THE GPU: one 256-thread workgroup turns one source
 * block's reception pattern into a device plan (plan.h), the same format planner_host.cpp emits.
 * Written, like solve_body.h, as per-thread phase functions: planner kernel (nrq_device.hip) =
 * phases + `__syncthreads()`; tests/emu runs the 256 threads of a phase in a loop on the CPU.
 *
 * Replaces the reference's patch_precode_matrix + precode_matrix_invert for decoding
 * (nanorq.c:527-547, precode.c:99-377) -- sort/precond/choose/swap_cols/update_nnz, make_U,
 * fwd_GE, fill_HDPC, solve_gf2/solve_gf256, backsolve -- by a parallel formulation:
 *   peeling   breadth-first rounds: every row with ONE unresolved V column claims it with an atomic
 *             compare-and-swap; a column leaving V updates its rows' (count, id-sum) word with one
 *             atomic subtract; when a round has no claimant the sparsest open row is found by a
 *             workgroup-wide atomic min and all but one of its columns are inactivated;
 *   W         fill-in bit rows by level (groups of 8 lanes per row);
 *   dense     GF(2) Gauss-Jordan in LDS on the leftover rows (pivot row = atomic min over the
 *             candidates of a column), then the H HDPC rows over GF(256) for what is left, with
 *             log/antilog tables staged in LDS;
 *   emission  XOR op stream by dependency level (scattered inside a level so that neighbouring
 *             lanes rarely share a target), slot maps, the block's job descriptor.
 * Any valid elimination order yields the same intermediate symbols (unique solution of A*C = D),
 * so plans differ from the host planner's while results are bit-identical.
 */
#ifndef NRQ_PLANNER_BODY_H
#define NRQ_PLANNER_BODY_H

#include <stdint.h>
#include <string.h>

#include "plan.h"
#include "rq_math.h"
#include "solve_body.h"

#ifndef PL_NT
#define PL_NT 1024u     /* threads of a planner workgroup: big blocks */
#endif
#define PL_NT_MIN 256u  /* small blocks (several workgroups per CU) */
#ifndef PL_NT_TINY
#define PL_NT_TINY 128u /* the smallest blocks (six or more workgroups per CU) */
#endif
#define PL_QCAP 2048u          /* frontier / claim queue capacity */
#define PL_UNASSIGNED 0x80000000u /* rowinfo bit 31: row has no pivot column (yet) */
#define PL_PATCHED 0x40000000u    /* rowinfo bit 30: this block replaced the base row (its base CSC entries are void) */
#define PL_LEVEL_MASK 0x3FFFFFFFu /* rowinfo bits 0..29: dependency level (so far) */
#define PL_ST_V 0u
#define PL_ST_PIVOT 1u
#define PL_ST_INACT 2u
#define PL_ST_CLAIM 3u
#define PL_NONE 0xFFFFFFFFu
#define PL_RING_EMPTY 0xFFFFFFFFu
#define PL_EMIT_TILES_MAX 24u   /* tiles of the op stream's permutation pass through LDS (pl_ops_emit_tiled) ... */
#define PL_EMIT_TILE_WORDS 12288u /* ... and op words a tile holds at most (48 KB) */
#define PL_PCHEAD 4u
#define PL_MAXH 16u
#define PL_PATCH_STRIDE RQ_MAX_LT_COLS
#define PL_MH_TILE 256u
#define PL_EXTRA_ROWS 8u   /* repair symbols a block may take beyond nrep when rank deficient */
#define PL_SPARE_ROWS 8u /* op rows reserved for the rows added that way */
#define PL_LOWCAP 1344u /* leftover rows the dense stage can take (>= inactive-column cap 1280 + 32) */
/* LDS kept for the dense stage when the peeling state is in LDS too: 36 KB for big blocks, less for small ones (whose
 * planner workgroups then share a CU) */
/* (what the dense stage takes is 16 bytes per inactive column plus the larger of Mb and the HDPC fold's tiles -- 4 KB + 1 KB per
 * 32 inactive columns: pl_low_b checks --; the smallest blocks, a few dozen inactive columns, get by with 6 KB + 6 L, and one
 * more of their workgroups fits a CU) */
SB_HD uint32_t pl_dense_reserve(uint32_t L) { const uint32_t r = (L < 1200u ? 6u : 8u) * 1024u + L * 6u; return r < 36u * 1024u ? (r + 15u) & ~15u : 36u * 1024u; }

/* Where a block's peeling state (rowstate, rowinfo, colinfo: 12 bytes per row) lives, given the dynamic LDS region of its
 * workgroup: 1 = in LDS next to the dense-stage reserve; 2 = in LDS with the dense stage taking over the rowstate image, which
 * is dead once peeling is over (blocks of ~8500 to ~11000 symbols: 12 bytes x 10300 rows + 36 KB do not fit, 12 x 10300 do, and
 * the image of 41 KB holds the reserve -- the level tables and class counters that live there between peeling and the W pass
 * are done with before the dense stage starts); 0 = in the workspace (HBM), with the compact copy in LDS.  One rule for
 * pl_ctx_setup, the launch (kernel instance, segmented run) and the tests. */
SB_HD uint32_t pl_state_in_lds(uint32_t L, uint32_t Mcap, uint32_t dyn_bytes) {
  const uint32_t img = (Mcap * 4u + 15u) & ~15u, need = 2u * img + ((L * 4u + 15u) & ~15u);
  if (need + pl_dense_reserve(L) <= dyn_bytes) return 1u;
#ifndef PL_NO_DENSE_OVERLAY
  if (need <= dyn_bytes && pl_dense_reserve(L) <= img) return 2u;
#endif
  return 0u;
}

/* what the host hands the planner for one block */
typedef struct nrq_planjob {
  uint64_t lost;     /* u32[nlost]: missing source ESIs, ascending */
  uint64_t rep_esi;  /* u32[nrep]: ESIs of the received repair symbols, arrival order */
  uint64_t work;     /* this block's workspace (nrq_planwork_bytes) */
  uint64_t arena;    /* plan arena out (capacity arena_cap bytes) */
  uint64_t src, rep, inter; /* forwarded into the solve job: symbol buffers of the block */
  uint32_t nlost, nrep; /* nrep: repair symbols to use up front (>= nlost) */
  uint32_t arena_cap;
  uint32_t nrep_avail;  /* >= nrep: further symbols the planner may take, one at a time, if the system is rank deficient */
  uint32_t mode;        /* 0 = decode; 1 = encode plan: no symbol is missing, the system is the encoder's (nlost = nrep = 0) */
  uint32_t pad_;
  uint64_t hdr_out;     /* 0, or where a second copy of the plan header goes: the headers of a batch side by side, so that they come
                         * back to the host as ONE copy (8192 headers out of 8192 arenas were a strided copy of ~150 us) */
} nrq_planjob;

/* ---- atomics: device intrinsics / plain memory in the emulator ---- */
#if defined(__HIP_DEVICE_COMPILE__)
#define PL_ATOM_ADD(p, v) atomicAdd((p), (v))
#define PL_ATOM_SUB(p, v) atomicSub((p), (v))
#define PL_ATOM_MAX(p, v) atomicMax((p), (v))
#define PL_ATOM_MIN(p, v) atomicMin((p), (v))
#define PL_ATOM_OR(p, v) atomicOr((p), (v))
#define PL_ATOM_XOR(p, v) atomicXor((p), (v))
#define PL_ATOM_CAS(p, c, v) atomicCAS((p), (c), (v))
/* minimum over the wave (every lane must call it), so that one lane per wave goes to the shared word */
#define PL_WAVE_MIN(v) __reduce_min_sync(~0ull, (unsigned int)(v))
#define PL_WAVE_MAX(v) __reduce_max_sync(~0ull, (unsigned int)(v))
#define PL_WAVE_LEADER(tid) (((tid) & 63u) == 0u)
/* `take` of the wave's lanes each want the next value of the counter *p: one atomic for all of them (every lane must call it;
 * lanes without `take` get garbage) -- thousands of single increments of one LDS word are served one after the other */
__device__ __forceinline__ uint32_t pl_wave_take(uint32_t *p, bool take) {
  const unsigned long long m = __ballot(take);
  if (!m) return 0u;
  const uint32_t lane = __lane_id(), leader = (uint32_t)__ffsll((long long)m) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(p, (uint32_t)__popcll(m));
  base = __shfl(base, (int)leader);
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
#define PL_WAVE_TAKE(p, take) pl_wave_take((p), (take))
/* compact peeling state (below): updates of the HBM copies that nobody waits for -- GLOBAL instructions without a return value
 * (a FLAT one would also count against the LDS counter of the wave and every LDS result would wait for it) -- and loads
 * that must see what other waves' atomics left in L2 */
#define PL_G32(p) ((__attribute__((address_space(1))) uint32_t *)(p))
/* (WORKGROUP scope: one workgroup owns a block's peeling state from the first to the last round, so its atomics may be
 * performed in the XCD's L2; at device scope -- the default of atomicSub() and friends -- gfx950 sends them past the L2) */
#ifndef PL_PEEL_SCOPE
#define PL_PEEL_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
#define PL_GSUB_NR(p, v) ((void)__hip_atomic_fetch_sub(PL_G32(p), (v), __ATOMIC_RELAXED, PL_PEEL_SCOPE))
#define PL_GMAX_NR(p, v) ((void)__hip_atomic_fetch_max(PL_G32(p), (v), __ATOMIC_RELAXED, PL_PEEL_SCOPE))
#define PL_GLOAD(p) __hip_atomic_load(PL_G32(p), __ATOMIC_RELAXED, PL_PEEL_SCOPE)
#define PL_GSTORE(p, v) (*PL_G32(p) = (v))
/* two words from two places in one trip (as atomic loads the compiler waits for the first before it asks for the second) */
__device__ __forceinline__ void pl_gload2(const uint32_t *a, const uint32_t *b, uint32_t &va, uint32_t &vb) {
  asm volatile("global_load_dword %0, %2, off sc0\n\tglobal_load_dword %1, %3, off sc0\n\ts_waitcnt vmcnt(0)"
               : "=&v"(va), "=&v"(vb) : "v"(a), "v"(b) : "memory");
}
#else
static inline void pl_gload2(const uint32_t *a, const uint32_t *b, uint32_t &va, uint32_t &vb) { va = *a; vb = *b; }
#define PL_GSUB_NR(p, v) ((void)pl_sub_((p), (v)))
#define PL_GMAX_NR(p, v) ((void)pl_max_((p), (v)))
#define PL_GLOAD(p) (*(p))
#define PL_GSTORE(p, v) (*(p) = (v))
#define PL_WAVE_TAKE(p, take) ((take) ? pl_add_((p), 1u) : 0u)
#define PL_WAVE_MIN(v) (v)
#define PL_WAVE_MAX(v) (v)
#define PL_WAVE_LEADER(tid) true
static inline uint32_t pl_add_(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
static inline uint32_t pl_sub_(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o - v; return o; }
static inline uint32_t pl_max_(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
static inline uint32_t pl_min_(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
static inline uint32_t pl_or_(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
static inline uint32_t pl_xor_(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o ^ v; return o; }
static inline uint32_t pl_cas_(uint32_t *p, uint32_t c, uint32_t v) { uint32_t o = *p; if (o == c) *p = v; return o; }
#define PL_ATOM_ADD(p, v) pl_add_((p), (v))
#define PL_ATOM_SUB(p, v) pl_sub_((p), (v))
#define PL_ATOM_MAX(p, v) pl_max_((p), (v))
#define PL_ATOM_MIN(p, v) pl_min_((p), (v))
#define PL_ATOM_OR(p, v) pl_or_((p), (v))
#define PL_ATOM_XOR(p, v) pl_xor_((p), (v))
#define PL_ATOM_CAS(p, c, v) pl_cas_((p), (c), (v))
#endif

SB_HD uint32_t pl_r16(uint32_t x) { return (x + 15u) & ~15u; }
/* for (i = tid; i < n; i += nt) store(i, load(i)) with the loads of PL_BATCH items in flight together.  Written the plain way
 * such a loop is a chain: the compiler may not move item i+1's load above item i's store (they could alias), so every item
 * pays its own trip to L2 / HBM -- 56 trips per thread and loop at K'=56403 (init and final phases: 6 M clocks per block). */
#ifndef PL_BATCH
#define PL_BATCH 8u
#endif
template <class LOAD, class STORE> SB_HD void pl_for_batched(uint32_t tid, uint32_t nt, uint32_t n, LOAD load, STORE store) {
  for (uint32_t i0 = tid; i0 < n; i0 += PL_BATCH * nt) {
    decltype(load(0u)) v[PL_BATCH];
#pragma unroll
    for (uint32_t j = 0; j < PL_BATCH; j++) { const uint32_t i = i0 + j * nt; v[j] = load(i < n ? i : i0); } /* (beyond the end: item i0 again, unused) */
#pragma unroll
    for (uint32_t j = 0; j < PL_BATCH; j++) { const uint32_t i = i0 + j * nt; if (i < n) store(i, v[j]); }
  }
}

/* A pointer held in PlanCtx is generic as far as the compiler can tell, and every access through it a FLAT instruction:
 * slower than a DS one, and ordered (vmcnt) behind the phase's outstanding global stores.  The workgroup state
 * (pl_shared) and the dense-stage region always live in LDS, the peeling arrays when they fit: say so. */
#if defined(__HIP_DEVICE_COMPILE__)
#define PL_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void *)(p)))
#else
#define PL_ASSUME_LDS(p) ((void)0)
#endif

#if defined(PL_STAMP) && defined(__HIP_DEVICE_COMPILE__)
#define PL_ST(c, i) do { if ((c).st_on) { const unsigned long long t_ = (unsigned long long)clock64(); \
    atomicAdd(&(c).sh->st_acc[i], t_ - (c).st_prev); (c).st_prev = t_; } } while (0)
#else
#define PL_ST(c, i) ((void)0)
#endif
#define PL_LIKELY(x) __builtin_expect(!!(x), 1)
#define PL_UNLIKELY(x) __builtin_expect(!!(x), 0)
/* Which form of the peeling phases an instance Z of the phase functions carries (PL_PEEL_DISPATCH3).  On the device the kernel
 * instance says where the state is -- launch_plan_kernel starts Z = 0 for blocks whose state fits the LDS by pl_ctx_setup's
 * rule, Z = 1 (compact state in LDS, arrays in HBM) for the others -- so each instance carries ONE form, not two or three: the
 * peeling loop's code is fetched round after round, and with the forms that never run compiled out the headline planner went
 * 3.19 -> 2.89 ms per 256 blocks.  The emulator (Z = 1) decides at run time, as does a PL_NO_COMPACT build. */
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(PL_NO_COMPACT) && PL_NO_COMPACT)
#define PL_PEEL_LDS(Z) ((Z) == 0)
#define PL_PEEL_PK(Z) ((Z) != 0)
#define PL_PEEL_FORM_OK(Z) ((Z) == 0 ? pl_peel_in_lds(c) : (!pl_peel_in_lds(c) && c.pk_cnt != nullptr))
#else
#define PL_PEEL_LDS(Z) pl_peel_in_lds(c)
#define PL_PEEL_PK(Z) ((Z) != 0 && c.pk_cnt)
#define PL_PEEL_FORM_OK(Z) true
#endif
/* ---- workgroup-shared scalars and small arrays (LDS) ---- */
/* status: 0 ok, 1 = not decodable (too few symbols / rank deficient), 2 = a planner capacity was
 * exceeded (queues, inactive-column cap, arena, LDS): the caller re-plans that block on the host */
#define PL_FAIL_SINGULAR 1u
#define PL_FAIL_CAPACITY 2u

typedef struct pl_shared {
#ifdef PL_STAMP
  unsigned long long st_acc[32]; /* (24..31: the chained peel's counters, PL_STAMP) */
#endif
  uint32_t status, fail_site; /* fail_site: source line that raised PL_FAIL_CAPACITY (diagnostics) */
  uint32_t defer_wt;          /* segmented run: W is transposed by nrq_wt_kernel, not by pl_final_c */
  uint32_t M, overhead, npatch, wpr, lpr, rowlen;
  uint32_t nV, npiv, ninact, nlev;
  uint32_t nq[2], nclaim[2]; /* frontier / claim counts, indexed by round parity */
  uint32_t ncand[2]; /* stack of open rows with two V columns: [0] entries, [1] scratch of the search (new top) */
  uint32_t best;
  uint32_t nlow, r2, nfree, cand[3];
  uint32_t arena_top, nrows, nrec, opbase; /* nrec: op records written by pl_w_init */
  uint32_t uslot_fill, tmp0, tmp1;
  uint32_t off_ops;
  uint32_t lv_in_lds, opq_group[2];
  uint32_t tmp_mhoff; /* byte offset of Mb inside the dense LDS region, behind MhT (fixed once nlow is known) */
  uint32_t dense_ok, nextra, spare_base, spare_fill, xcol, xrow[48]; /* W pass: level tables staged in LDS; which group each op buffer holds */
  uint32_t ndone;             /* chained peeling (pl_round_chain): claims of the current list that have been dropped */
  uint32_t own_hits;          /* plan check: entries (pivot row, its own pivot column) met by the entry pass (pl_w_entries) -- must be npiv */
  uint32_t chk_piv, chk_inact;/* plan check (pl_check_b): columns found in state pivot / inactive */
  uint32_t bkt_n[PL_EMIT_TILES_MAX]; /* pl_ops_emit_tiled: ops sorted out to every tile of the stream so far */
  uint32_t ev_n, ev_min, ev_none; /* inactivation event, peeling state in LDS (pl_event_*): open rows with two V columns listed, the
                                   * sparsest of the others, "no open row is left" */
  uint32_t off_augt, aug_stride, mhrev; /* mhrev: the HDPC fold's z rows are in the workspace (pl_mhrev_store) */
  uint32_t bin_ct[NRQ_LANE_CLASSES]; /* ops of the GF(2) combination group per lane class of the target */
  uint32_t gj_A[32];   /* blocked Gauss-Jordan: pivot row b at its pivot step = XOR of the panel-start rows gj_pr[k], k in gj_A[b] */
  uint16_t gj_pr[32];  /* pivot row of bit b of the current panel (PL_NONE16: the column is free) */
  /* The arrays whose size depends on the launch follow the struct in LDS (pl_tail_*): frontier queues queue[2][qcap],
   * claim lists claim_l[qcap] / claim_c[qcap] (columns claimed this round: level + 1 of the pivot, column),
   * partial[nt] (per-thread scratch), gj_flag[lowcap] / gj_used[lowcap] (Gauss-Jordan: bit of the current column /
   * row already a pivot).  Small blocks get small ones, so that four planner workgroups share a CU. */
  uint32_t freex[NRQ_MAX_FREE];
  uint8_t gf_exp[512], gf_log[256];
  uint8_t aug[PL_MAXH * (NRQ_MAX_FREE + PL_MAXH)];
  uint8_t solver[NRQ_MAX_FREE];
  uint8_t taken[PL_MAXH];
  uint8_t colf[PL_MAXH];
} pl_shared;
SB_HD uint32_t pl_shared_bytes(uint32_t qcap, uint32_t lowcap, uint32_t nt) {
  return pl_r16((uint32_t)sizeof(pl_shared)) + pl_r16(qcap * 8u) + pl_r16(nt * 4u) + pl_r16(lowcap * 2u);
}

SB_HD bool pl_bin_in_stream(uint32_t L) { return L < NRQ_AUG_MATRIX_MIN_L; } /* (plan.h: the GF(2) combinations as ops of the stream) */

/* ---- per-block workspace in HBM (offsets from job.work) ---- */
typedef struct pl_work_layout {
  uint32_t rowstate, rowinfo, colinfo, patch_of, patch_cols, patch_len, pc_ptr, pc_fill, pc_rows, pc_head, ucol, wrows,
      lev_ops, lev_base, lev_fill, pivdeg, lowdeg, lev_fin, red_row, red_x, rec_word, rec_idx, rec_g, cand, sh_save, mh_ext, cls_g, wentry, emit_bkt, chk, total;
} pl_work_layout;

/* nnzcap: entries of the base structure plus the patch rows (bounds the number of row ops) */
SB_HD pl_work_layout pl_work_plan(uint32_t L, uint32_t Mcap, uint32_t npcap, uint32_t ucap, uint32_t nnzcap) {
  pl_work_layout w;
  uint32_t o = 0;
  const uint32_t wprcap = (ucap + 31u) / 32u;
  w.rowstate = o;   o = pl_r16(o + Mcap * 4u);
  w.rowinfo = o;    o = pl_r16(o + Mcap * 4u);
  w.colinfo = o;    o = pl_r16(o + L * 4u);
  w.patch_of = o;   o = pl_r16(o + Mcap * 2u);
  w.patch_cols = o; o = pl_r16(o + npcap * PL_PATCH_STRIDE * 2u);
  w.patch_len = o;  o = pl_r16(o + npcap);
  w.pc_ptr = o;     o = pl_r16(o + (L + 1u) * 4u);
  w.pc_fill = o;    o = pl_r16(o + (L + 1u) * 4u);
  w.pc_rows = o;    o = pl_r16(o + npcap * PL_PATCH_STRIDE * 2u);
  w.pc_head = o;    o = pl_r16(o + L * PL_PCHEAD * 2u); /* the first PL_PCHEAD patch rows of every column (pl_pcsc_fill; the chained peel reads them with pc_fill, the column's count) */
  w.ucol = o;       o = pl_r16(o + ucap * 2u);
  w.wrows = o;      o = pl_r16(o + Mcap * wprcap * 4u); /* W rows by SLOT (pivot rows and leftover rows) */
  w.lev_ops = o;    o = pl_r16(o + (L + 2u) * 4u);
  w.lev_base = o;   o = pl_r16(o + (L + 2u) * 4u);
  w.lev_fill = o;   o = pl_r16(o + (L + 2u) * 4u);
  w.pivdeg = o;     o = pl_r16(o + L * 4u);
  w.lowdeg = o;     o = pl_r16(o + Mcap * 4u);
  w.lev_fin = o;    o = pl_r16(o + (L + 2u) * 4u);
  w.red_row = o;    o = pl_r16(o + ucap * 4u);
  w.red_x = o;      o = pl_r16(o + ucap * 4u);
  /* op records (pl_w_init -> pl_ops_emit): op word, place inside its group, group */
  w.rec_word = o;   o = pl_r16(o + nnzcap * 4u);
  w.rec_idx = o;    o = pl_r16(o + nnzcap * 4u);
  w.rec_g = o;      o = pl_r16(o + nnzcap * 2u);
  w.cand = o;       o = pl_r16(o + Mcap * 2u); /* stack of open rows with exactly two V columns (pl_inact_find) */
  w.sh_save = o;    o = pl_r16(o + pl_shared_bytes(PL_QCAP, PL_LOWCAP, PL_NT)); /* pl_shared (with its arrays) between the parts of a segmented run (planner_seq.h) */
  w.mh_ext = o;     o = pl_r16(o + ucap * PL_MAXH);               /* MhT as nrq_mh_kernel leaves it (16 bytes per inactive column) */
  w.cls_g = o;      o = pl_r16(o + (L + 2u) * 32u);               /* the class counters (PL_CLS_BYTES per level group) while nrq_wentry_kernel's workgroups count into them */
  w.wentry = o;     o = pl_r16(o + 16u);                           /* ... and their record counter / failure report */
  w.emit_bkt = o;   o = pl_r16(o + 3u * nnzcap * 8u);             /* (place inside the tile, op word), a list per tile of the stream: pl_ops_emit_tiled */
  w.chk = o;        o = pl_r16(o + Mcap * 2u);                     /* pl_check_a / _b: the pivot that owns every slot */
  w.total = o;
  return w;
}

/* upper bound of a plan arena (header + arrays + rowsrc + output lists) for a block of this size */
SB_HD uint32_t pl_arena_bound(uint32_t L, uint32_t Mcap, uint32_t ucap, uint32_t nnz, uint32_t nlost_cap) {
  const uint32_t wprcap = (ucap + 31u) / 32u, npad = (L + 63u) & ~63u;
  /* op stream: the ops themselves, one partly filled row plus NRQ_PIPE-1 spacer rows per level, the spare and
   * the lead/padding rows */
  uint32_t ops = (2u * nnz + nnz / 2u) + NRQ_ROW * (NRQ_PIPE * (L / 2u + 8u) + PL_SPARE_ROWS + NRQ_PAD_ROWS + 4u);
  if (pl_bin_in_stream(L)) ops += 2u * ucap * 64u;             /* (the GF(2) combinations as ops of the stream) */
  else ops += ((ucap + 3u) & ~3u) * ((PL_LOWCAP + 31u) / 32u) + 16u; /* (or as words of a bit matrix) */
  uint32_t b = 256u + L * 2u * 3u + (L + 16u) * 2u + ucap * 2u * 4u + ucap * 4u * 2u + PL_MAXH * ucap +
               NRQ_MAX_FREE * PL_MAXH + wprcap * npad * 4u + ops * 4u +
               Mcap * 4u + (nlost_cap + 1u) * 8u + nlost_cap * PL_PATCH_STRIDE * 2u + 1024u;
  return pl_r16(b);
}

/* ---- context ---- */
struct PlanCtx {
  /* inputs */
  rq_params p;
  const uint8_t *kc;
  const nrq_kconst_hdr *kh;
  const uint32_t *b_rptr, *b_cptr, *b_state;
  const uint16_t *b_cidx, *b_ridx, *b_erow, *b_chead;
  const uint8_t *G, *GT;
  nrq_planjob job;
  const uint32_t *lost, *rep_esi;
  pl_shared *sh;
  uint32_t qcap, lowcap, sh_bytes; /* capacities of the arrays behind pl_shared (pl_shared_bytes) */
  uint16_t *qmem;    /* queue[2][qcap], claim_l[qcap], claim_c[qcap] */
  uint32_t *partial; /* [nt] */
  uint8_t *gjmem;    /* gj_flag[lowcap], gj_used[lowcap] */
  SB_MEM uint16_t *queue(uint32_t pq) const { uint16_t *q = qmem + (size_t)pq * qcap; PL_ASSUME_LDS(q); return q; }
  SB_MEM uint16_t *claim_l() const { uint16_t *q = qmem + 2u * (size_t)qcap; PL_ASSUME_LDS(q); return q; }
  SB_MEM uint16_t *claim_c() const { uint16_t *q = qmem + 3u * (size_t)qcap; PL_ASSUME_LDS(q); return q; }
  /* the chained peel's list: the same bytes as qcap words, column | level + 1 << 16 (pl_round_chain_dev) */
  SB_MEM uint32_t *ring() const { uint32_t *q = reinterpret_cast<uint32_t *>(qmem + 2u * (size_t)qcap); PL_ASSUME_LDS(q); return q; }
  SB_MEM uint32_t *part() const { uint32_t *q = partial; PL_ASSUME_LDS(q); return q; }
  SB_MEM uint8_t *gj_flag() const { uint8_t *q = gjmem; PL_ASSUME_LDS(q); return q; }
  SB_MEM uint8_t *gj_used() const { uint8_t *q = gjmem + lowcap; PL_ASSUME_LDS(q); return q; }
  /* during peeling the same bytes hold one bit per column: "has entries in the patch CSC" (pl_init_a clears, pl_init_b sets,
   * the chained drop phase reads -- a column without the bit needs no trip to pc_ptr); nullptr when L has more bits than that */
  SB_MEM uint32_t *pcbits() const { if (p.L > 16u * lowcap) return nullptr; uint32_t *q = reinterpret_cast<uint32_t *>(gjmem); PL_ASSUME_LDS(q); return q; }
  uint8_t *lds_dyn; /* dynamic LDS region: peeling state (if it fits) and the dense stage (Mb, Mh) */
  uint32_t lds_dyn_bytes;
  uint8_t *dense_lds;
  uint32_t dense_bytes;
  uint8_t *aux_lds; /* >= 32 KiB of LDS that is idle between the end of peeling and pl_low_c */
  uint32_t aux_bytes;
  uint32_t Mcap, npcap, ucap;
  /* workspace views */
  pl_work_layout wl;
  uint8_t *work;
  uint32_t *rowstate, *rowinfo, *colinfo;
  /* compact peeling state in LDS (big blocks, whose three arrays above stay in HBM): what a peeling round DECIDES on --
   * the number of V columns of every row (a byte), "row has no pivot yet", "row was replaced by this block", "column has
   * left V" (a bit each): 79 KB at K'=56403.  The HBM arrays are kept up to date by stores and atomics nobody waits for
   * (later phases read them); a round then waits for three trips to memory instead of six (pl_round_claim_k). */
  uint32_t *pk_cnt, *pk_un, *pk_pa, *pk_vb;
#ifdef PL_STAMP /* diagnostic build: thread 0 of block 0 accumulates shader clocks between points of a peeling round (pl_shared::st_acc) */
  unsigned long long st_prev;
  bool st_on;
  bool st_blk; /* every thread of block 0 */
#endif
  /* the entry pass on many workgroups (nrq_wentry_kernel, big blocks; job.mode bit 9): the column levels stay in HBM whatever the
   * LDS would hold, and while the helper's workgroups count, the class counters and the record counter are the workspace's */
  bool collev_hbm;
  uint32_t *cls_glob;  /* non-null: pl_cls() is this array in HBM */
  uint32_t *nrec_ptr;  /* the record counter: &sh->nrec, or the workspace's */
  uint32_t *wentry;    /* workspace: [0] records, [1] status, [2] fail site, [3] own-column entries met (plan check) */
  uint32_t *own_ptr;   /* the plan check's counter of (pivot row, own pivot column) entries: &sh->own_hits, or the workspace's */
  uint16_t *chk;       /* workspace: pivot that owns slot r (pl_check_a / _b) */
  uint16_t *patch_of, *patch_cols, *pc_rows, *pc_head, *ucol;
  uint32_t *emit_bkt;
  bool pcfill_lds; /* pc_fill (patch entries per column) lives in the dense stage's LDS region, idle until peeling is over */
  uint8_t *patch_len;
  uint32_t *pc_ptr, *pc_fill, *wrows, *lev_ops, *lev_base, *lev_fill, *pivdeg, *lowdeg, *lev_fin, *red_row,
      *red_x, *rec_word, *rec_idx;
  uint16_t *rec_g, *cand;
  uint32_t reccap;
  /* arena views (fixed part laid out up front) */
  uint8_t *arena;
  nrq_plan_hdr *hdr;
  uint16_t *pivslot, *pivcol, *colslot, *pivof, *uslot, *lowslot, *pivx, *freex_out;
  uint32_t *fbits;
  uint8_t *mh, *hinv;
  uint32_t off_pivslot, off_pivcol, off_colslot, off_pivof, off_uslot, off_lowslot, off_pivx, off_fbits, off_mh,
      off_freex, off_hinv, fixed_end;
  nrq_job *jobout;
};

/* a pointer read from a job record is generic to the compiler (FLAT accesses); these all point into HBM */
#if defined(__HIP_DEVICE_COMPILE__)
#define PL_HBM(T, v) ((T *)(__attribute__((address_space(1))) T *)(uintptr_t)(v))
#else
#define PL_HBM(T, v) (reinterpret_cast<T *>(v))
#endif
SB_HD void pl_ctx_setup(PlanCtx &c, const rq_params &prm, const uint8_t *kc, const nrq_planjob &job, pl_shared *sh,
                        uint8_t *lds_dyn, uint32_t lds_dyn_bytes, uint32_t Mcap, uint32_t npcap, uint32_t ucap,
                        nrq_job *jobout, uint32_t qcap = PL_QCAP, uint32_t lowcap = PL_LOWCAP, uint32_t nt = PL_NT) {
  c.qcap = qcap; c.lowcap = lowcap;
  c.sh_bytes = pl_shared_bytes(qcap, lowcap, nt);
  {
    uint8_t *t = reinterpret_cast<uint8_t *>(sh) + pl_r16((uint32_t)sizeof(pl_shared));
    c.qmem = reinterpret_cast<uint16_t *>(t); t += pl_r16(qcap * 8u);
    c.partial = reinterpret_cast<uint32_t *>(t); t += pl_r16(nt * 4u);
    c.gjmem = t;
  }
  c.kc = kc;
  c.kh = reinterpret_cast<const nrq_kconst_hdr *>(kc);
  c.p = prm;
  c.b_rptr = reinterpret_cast<const uint32_t *>(kc + c.kh->off_rptr);
  c.b_cidx = reinterpret_cast<const uint16_t *>(kc + c.kh->off_cidx);
  c.b_cptr = reinterpret_cast<const uint32_t *>(kc + c.kh->off_cptr);
  c.b_ridx = reinterpret_cast<const uint16_t *>(kc + c.kh->off_ridx);
  c.b_chead = reinterpret_cast<const uint16_t *>(kc + c.kh->off_chead);
  c.b_erow = reinterpret_cast<const uint16_t *>(kc + c.kh->off_erow);
  c.b_state = reinterpret_cast<const uint32_t *>(kc + c.kh->off_state);
  c.G = kc + c.kh->off_g;
  c.GT = kc + c.kh->off_gt;
  c.job = job;
  c.lost = PL_HBM(const uint32_t, job.lost);
  c.rep_esi = PL_HBM(const uint32_t, job.rep_esi);
  c.sh = sh;
  c.lds_dyn = lds_dyn;
  c.lds_dyn_bytes = lds_dyn_bytes;
  c.Mcap = Mcap; c.npcap = npcap; c.ucap = ucap;
  c.reccap = c.kh->nnz + npcap * PL_PATCH_STRIDE;
  c.wl = pl_work_plan(c.p.L, Mcap, npcap, ucap, c.reccap);
  c.work = PL_HBM(uint8_t, job.work);
  uint8_t *w = c.work;
  c.rowstate = reinterpret_cast<uint32_t *>(w + c.wl.rowstate);
  c.rowinfo = reinterpret_cast<uint32_t *>(w + c.wl.rowinfo);
  c.colinfo = reinterpret_cast<uint32_t *>(w + c.wl.colinfo);
  /* the three hot peeling arrays live in LDS when they fit next to the dense-stage reserve; the
   * dense stage (Mb, Mh) uses what is left of the dynamic region */
  c.dense_lds = lds_dyn;
  c.dense_bytes = lds_dyn_bytes;
  c.aux_lds = lds_dyn;
  c.aux_bytes = lds_dyn_bytes;
  {
    uint32_t need = pl_r16(Mcap * 4u) * 2u + pl_r16(c.p.L * 4u);
    const uint32_t where = lds_dyn ? pl_state_in_lds(c.p.L, Mcap, lds_dyn_bytes) : 0u;
    if (where) {
      c.rowstate = reinterpret_cast<uint32_t *>(lds_dyn);
      c.rowinfo = reinterpret_cast<uint32_t *>(lds_dyn + pl_r16(Mcap * 4u));
      c.colinfo = reinterpret_cast<uint32_t *>(lds_dyn + 2u * pl_r16(Mcap * 4u));
      c.aux_bytes = pl_r16(Mcap * 4u); /* the rowstate image, dead once peeling is over */
      if (where == 1u) {
        c.dense_lds = lds_dyn + need;
        c.dense_bytes = lds_dyn_bytes - need;
      } else { /* the dense stage takes over the rowstate image (= the aux region: pl_cls_place / pl_collev_in_lds see one region) */
        c.dense_lds = lds_dyn;
        c.dense_bytes = c.aux_bytes;
      }
    }
    c.collev_hbm = (job.mode & 0x200u) != 0u;
    c.cls_glob = nullptr;
    c.nrec_ptr = &sh->nrec;
    c.own_ptr = &sh->own_hits;
    c.wentry = reinterpret_cast<uint32_t *>(w + c.wl.wentry);
    c.chk = reinterpret_cast<uint16_t *>(w + c.wl.chk);
    c.pk_cnt = c.pk_un = c.pk_pa = c.pk_vb = nullptr;
    const uint32_t pk_cnt_b = pl_r16(Mcap + 4u), pk_row_b = pl_r16((Mcap + 31u) / 32u * 4u), pk_col_b = pl_r16((c.p.L + 31u) / 32u * 4u);
#ifndef PL_NO_COMPACT
#define PL_NO_COMPACT 0
#endif
    if (!PL_NO_COMPACT && lds_dyn && c.rowstate != reinterpret_cast<uint32_t *>(lds_dyn) && pk_cnt_b + 2u * pk_row_b + pk_col_b <= lds_dyn_bytes) {
      /* (the dynamic region is idle during peeling when the state is not in it: the dense stage and the level tables come later) */
      c.pk_cnt = reinterpret_cast<uint32_t *>(lds_dyn);
      c.pk_un = reinterpret_cast<uint32_t *>(lds_dyn + pk_cnt_b);
      c.pk_pa = reinterpret_cast<uint32_t *>(lds_dyn + pk_cnt_b + pk_row_b);
      c.pk_vb = reinterpret_cast<uint32_t *>(lds_dyn + pk_cnt_b + 2u * pk_row_b);
    }
  }
  c.patch_of = reinterpret_cast<uint16_t *>(w + c.wl.patch_of);
  c.patch_cols = reinterpret_cast<uint16_t *>(w + c.wl.patch_cols);
  c.patch_len = w + c.wl.patch_len;
  c.pc_ptr = reinterpret_cast<uint32_t *>(w + c.wl.pc_ptr);
  c.pc_fill = reinterpret_cast<uint32_t *>(w + c.wl.pc_fill);
  /* (the counts of patch entries per column -- counted and handed out by atomics while the patch CSC is built, read by the chained
   * peel for every column with patch rows: in the dense stage's region when that is idle until peeling is over, i.e. when the
   * peeling state has its own place in LDS) */
  c.pcfill_lds = lds_dyn && c.dense_lds != c.aux_lds && c.dense_lds != lds_dyn && (c.p.L + 1u) * 4u <= c.dense_bytes;
  if (c.pcfill_lds) c.pc_fill = reinterpret_cast<uint32_t *>(c.dense_lds);
  c.pc_rows = reinterpret_cast<uint16_t *>(w + c.wl.pc_rows);
  c.pc_head = reinterpret_cast<uint16_t *>(w + c.wl.pc_head);
  c.ucol = reinterpret_cast<uint16_t *>(w + c.wl.ucol);
  c.wrows = reinterpret_cast<uint32_t *>(w + c.wl.wrows);
  c.lev_ops = reinterpret_cast<uint32_t *>(w + c.wl.lev_ops);
  c.lev_base = reinterpret_cast<uint32_t *>(w + c.wl.lev_base);
  c.lev_fill = reinterpret_cast<uint32_t *>(w + c.wl.lev_fill);
  c.pivdeg = reinterpret_cast<uint32_t *>(w + c.wl.pivdeg);
  c.lowdeg = reinterpret_cast<uint32_t *>(w + c.wl.lowdeg);
  c.lev_fin = reinterpret_cast<uint32_t *>(w + c.wl.lev_fin);
  c.red_row = reinterpret_cast<uint32_t *>(w + c.wl.red_row);
  c.red_x = reinterpret_cast<uint32_t *>(w + c.wl.red_x);
  c.rec_word = reinterpret_cast<uint32_t *>(w + c.wl.rec_word);
  c.rec_idx = reinterpret_cast<uint32_t *>(w + c.wl.rec_idx);
  c.rec_g = reinterpret_cast<uint16_t *>(w + c.wl.rec_g);
  c.cand = reinterpret_cast<uint16_t *>(w + c.wl.cand);
  c.emit_bkt = reinterpret_cast<uint32_t *>(w + c.wl.emit_bkt);
  /* arena: header, then the arrays whose size is bounded by (L, ucap) */
  c.arena = PL_HBM(uint8_t, job.arena);
  c.hdr = reinterpret_cast<nrq_plan_hdr *>(c.arena);
  uint32_t o = pl_r16((uint32_t)sizeof(nrq_plan_hdr));
  const uint32_t L = c.p.L, n_hd = c.p.Kp + c.p.S;
  c.off_pivslot = o; o = pl_r16(o + L * 2u);
  c.off_pivcol = o;  o = pl_r16(o + L * 2u);
  c.off_colslot = o; o = pl_r16(o + L * 2u);
  c.off_pivof = o;   o = pl_r16(o + ((n_hd + 7u) & ~7u) * 2u);
  c.off_uslot = o;   o = pl_r16(o + ucap * 2u);
  c.off_lowslot = o; o = pl_r16(o + ucap * 2u + 64u);
  c.off_pivx = o;    o = pl_r16(o + ucap * 2u);
  c.off_fbits = o;   o = pl_r16(o + ucap * 4u);
  c.off_mh = o;      o = pl_r16(o + PL_MAXH * ucap);
  c.off_freex = o;   o = pl_r16(o + NRQ_MAX_FREE * 2u);
  c.off_hinv = o;    o = pl_r16(o + NRQ_MAX_FREE * PL_MAXH);
  c.fixed_end = o;
  c.pivslot = reinterpret_cast<uint16_t *>(c.arena + c.off_pivslot);
  c.pivcol = reinterpret_cast<uint16_t *>(c.arena + c.off_pivcol);
  c.colslot = reinterpret_cast<uint16_t *>(c.arena + c.off_colslot);
  c.pivof = reinterpret_cast<uint16_t *>(c.arena + c.off_pivof);
  c.uslot = reinterpret_cast<uint16_t *>(c.arena + c.off_uslot);
  c.lowslot = reinterpret_cast<uint16_t *>(c.arena + c.off_lowslot);
  c.pivx = reinterpret_cast<uint16_t *>(c.arena + c.off_pivx);
  c.fbits = reinterpret_cast<uint32_t *>(c.arena + c.off_fbits);
  c.mh = c.arena + c.off_mh;
  c.freex_out = reinterpret_cast<uint16_t *>(c.arena + c.off_freex);
  c.hinv = c.arena + c.off_hinv;
  c.jobout = jobout;
}

/* columns of constraint row r: base structure unless this block patched the row */
SB_HD uint32_t pl_row(const PlanCtx &c, uint32_t r, const uint16_t **cols) {
  uint32_t pi = c.patch_of[r];
  if (pi != 0xFFFFu) {
    *cols = c.patch_cols + (size_t)pi * PL_PATCH_STRIDE;
    return c.patch_len[pi];
  }
  if (r >= c.p.L) { *cols = c.patch_cols; return 0; }
  uint32_t a = c.b_rptr[r];
  *cols = c.b_cidx + a;
  return c.b_rptr[r + 1] - a;
}

SB_HD uint8_t pl_gfmul(const pl_shared *sh, uint8_t a, uint8_t b) {
  return (a && b) ? sh->gf_exp[(uint32_t)sh->gf_log[a] + sh->gf_log[b]] : 0;
}

/* =============================== phase 0: inputs, patch rows, state ========================== */
SB_HD bool pl_peel_in_lds(const PlanCtx &c);
template <int Z> SB_HD void pl_init_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const rq_params &p = c.p;
  if (tid == 0) {
    uint32_t st = 0;
    const uint32_t nl = c.job.nlost, nr = c.job.nrep;
    if ((nl == 0 && (c.job.mode & 0xFFu) != 1u) || nr < nl) st = PL_FAIL_SINGULAR;
    uint32_t oh = st ? 0 : nr - nl;
    if (!st && (p.L + oh + PL_EXTRA_ROWS > c.Mcap || nr + PL_EXTRA_ROWS > c.npcap || p.L + oh + PL_EXTRA_ROWS > 65534u))
      st = PL_FAIL_CAPACITY;
    /* (the kernel instances carry one form of the peeling phases each, PL_PEEL_DISPATCH3: a block whose state is not where its
     * instance expects it goes back to the host planner) */
    if (!st && !PL_PEEL_FORM_OK(Z)) { st = PL_FAIL_CAPACITY; sh->fail_site = __LINE__; } else sh->fail_site = 0;
    sh->status = st;
    sh->defer_wt = c.job.mode >> 8; /* (bit 8 of the job's mode: set by the host for segmented runs) */
    sh->overhead = oh;
    sh->M = p.L + oh;
    sh->npatch = st ? 0 : nr;
    sh->nV = p.W; sh->npiv = 0; sh->ninact = 0; sh->nlev = 0;
    sh->nq[0] = sh->nq[1] = 0; sh->nclaim[0] = sh->nclaim[1] = 0; sh->best = PL_NONE;
    sh->ncand[0] = sh->ncand[1] = 0;
    sh->nlow = 0; sh->r2 = 0; sh->nfree = 0; sh->cand[0] = sh->cand[1] = sh->cand[2] = PL_NONE;
    sh->nrows = 0; sh->nrec = 0; sh->uslot_fill = 0;
    sh->dense_ok = 0; sh->nextra = 0; sh->spare_base = 0; sh->spare_fill = 0; sh->xcol = PL_NONE; sh->mhrev = 0; sh->ndone = 0; sh->ev_n = 0; sh->ev_min = PL_NONE; sh->ev_none = 0;
  }
  /* GF(256) tables into LDS (RFC 6330 section 5.7): generated by one thread, 255 steps */
  if (tid == 1 % nt) {
    uint32_t x = 1;
    for (uint32_t e = 0; e < 255; e++) {
      sh->gf_exp[e] = (uint8_t)x;
      sh->gf_log[x] = (uint8_t)e;
      x <<= 1;
      if (x & 0x100u) x ^= 0x11Du;
    }
    for (uint32_t e = 255; e < 512; e++) sh->gf_exp[e] = sh->gf_exp[e - 255];
    sh->gf_log[0] = 0;
  }
  const uint32_t L = p.L;
  pl_for_batched(tid, nt, c.Mcap, [&](uint32_t r) { return r < L ? c.b_state[r] : 0u; },
                 [&](uint32_t r, uint32_t st) { c.rowstate[r] = st; c.rowinfo[r] = PL_UNASSIGNED; c.patch_of[r] = 0xFFFFu; });
  for (uint32_t col = tid; col < L; col += nt) {
    c.colinfo[col] = col < p.W ? 0u : ((PL_ST_INACT << 30) | (col - p.W));
    c.pc_fill[col] = 0;
  }
  for (uint32_t x = tid; x < p.P; x += nt) c.ucol[x] = (uint16_t)(p.W + x);
  if (uint32_t *pcb = c.pcbits())
    for (uint32_t j = tid; j < c.lowcap / 2u; j += nt) pcb[j] = 0u; /* (gj_flag / gj_used: theirs again from pl_lev_a on) */
  for (uint32_t j = tid; j < c.qcap; j += nt) c.ring()[j] = PL_RING_EMPTY; /* (claim_l and claim_c as one list of words: pl_round_chain polls it) */
  if (Z != 0 && c.pk_cnt) {
    uint32_t *cnt = c.pk_cnt, *un = c.pk_un, *pa = c.pk_pa, *vb = c.pk_vb;
    PL_ASSUME_LDS(cnt); PL_ASSUME_LDS(un); PL_ASSUME_LDS(pa); PL_ASSUME_LDS(vb);
    for (uint32_t w = tid; w * 4u < c.Mcap; w += nt) { /* four rows' counts per word */
      uint32_t v = 0;
      for (uint32_t q = 0; q < 4u; q++) {
        const uint32_t r = w * 4u + q;
        if (r < L) v |= (c.b_state[r] >> 24) << (8u * q);
      }
      cnt[w] = v;
    }
    for (uint32_t w = tid; w * 32u < c.Mcap; w += nt) { un[w] = 0xFFFFFFFFu; pa[w] = 0u; }
    for (uint32_t w = tid; w * 32u < L; w += nt) { /* bit set = the column is not (or no longer) in V */
      const uint32_t lo = w * 32u;
      vb[w] = lo >= p.W ? 0xFFFFFFFFu : (lo + 32u <= p.W ? 0u : ~((1u << (p.W - lo)) - 1u));
    }
  }
}

/* validate the inputs and expand the patched rows (thread per received repair symbol) */
template <int Z> SB_HD void pl_init_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const rq_params &p = c.p;
  const uint32_t nl = c.job.nlost, pad = p.Kp - p.K;
  uint32_t *pcb = c.pcbits();
  for (uint32_t i = tid; i < sh->npatch; i += nt) {
    uint32_t esi = c.rep_esi[i], row;
    bool bad = esi < p.K || esi >= (1u << 24);
    if (i < nl) {
      uint32_t e = c.lost[i];
      if (e >= p.K || (i && e <= c.lost[i - 1])) bad = true;
      row = p.S + p.H + (bad ? 0u : e);
    } else {
      row = p.L + (i - nl);
    }
    if (bad) { sh->status = PL_FAIL_SINGULAR; continue; }
    uint32_t cols[RQ_MAX_LT_COLS];
    uint32_t n = rq_lt_columns(&p, esi + pad, cols);
    uint16_t *dst = c.patch_cols + (size_t)i * PL_PATCH_STRIDE;
    uint32_t cnt = 0, sum = 0;
    for (uint32_t k = 0; k < n; k++) {
      dst[k] = (uint16_t)cols[k];
      if (cols[k] < p.W) { cnt++; sum += cols[k]; }
      PL_ATOM_ADD(&c.pc_fill[cols[k]], 1u);
      if (pcb) PL_ATOM_OR(&pcb[cols[k] >> 5], 1u << (cols[k] & 31u));
    }
    c.patch_len[i] = (uint8_t)n;
    c.patch_of[row] = (uint16_t)i;
    c.rowstate[row] = (cnt << 24) | sum;
    c.rowinfo[row] = PL_UNASSIGNED | PL_PATCHED;
    if (Z != 0 && c.pk_cnt) { /* (pl_init_a wrote the base row's count: replace the byte) */
      uint32_t *pc = c.pk_cnt, *pa = c.pk_pa;
      PL_ASSUME_LDS(pc); PL_ASSUME_LDS(pa);
      const uint32_t sh8 = (row & 3u) * 8u, old = (pc[row >> 2] >> sh8) & 0xFFu;
      PL_ATOM_ADD(&pc[row >> 2], (cnt - old) << sh8); /* (wraps inside the byte's lane of the sum: cnt - old may be negative) */
      PL_ATOM_OR(&pa[row >> 5], 1u << (row & 31u));
    }
  }
}

/* exclusive scan of pc_fill[0..L) into pc_ptr[0..L]; three steps */
template <int Z> SB_HD void pl_scan_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  const uint32_t L = c.p.L, per = (L + nt - 1) / nt;
  uint32_t a = tid * per, b = a + per < L ? a + per : L, s = 0;
  for (uint32_t k = a; k < b; k++) s += c.pc_fill[k];
  c.part()[tid] = s;
}
template <int Z> SB_HD void pl_scan_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  /* one wave: a lane sums nt / 64 partial sums, the lanes' sums are scanned across the wave, the lane writes its run back.  (One
   * thread walking all nt words was a chain of nt LDS round trips: ~130 k clocks at 1024 threads, 3 % of the planner at K=8192.) */
  if (tid >= 64u) return;
  uint32_t *part = c.part();
  const uint32_t per = (nt + 63u) / 64u, a = tid * per, b = a + per < nt ? a + per : nt;
  uint32_t sum = 0;
  for (uint32_t t = a; t < b; t++) sum += part[t];
  uint32_t incl = sum;
#pragma unroll
  for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (tid >= d) incl += o; }
  uint32_t run = incl - sum;
  for (uint32_t t = a; t < b; t++) { const uint32_t v = part[t]; part[t] = run; run += v; }
  if (tid == 63u) c.pc_ptr[c.p.L] = incl;
#else
  if (tid != 0) return;
  uint32_t run = 0;
  for (uint32_t t = 0; t < nt; t++) { uint32_t v = c.part()[t]; c.part()[t] = run; run += v; }
  c.pc_ptr[c.p.L] = run;
#endif
}
template <int Z> SB_HD void pl_scan_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  const uint32_t L = c.p.L, per = (L + nt - 1) / nt;
  uint32_t a = tid * per, b = a + per < L ? a + per : L, run = c.part()[tid];
  for (uint32_t k0 = a; k0 < b; k0 += PL_BATCH) { /* (the counts of a batch first, then the running sum over them) */
    uint32_t v[PL_BATCH];
#pragma unroll
    for (uint32_t j = 0; j < PL_BATCH; j++) v[j] = c.pc_fill[k0 + j < b ? k0 + j : k0];
#pragma unroll
    for (uint32_t j = 0; j < PL_BATCH; j++)
      if (k0 + j < b) { c.pc_ptr[k0 + j] = run; run += v[j]; c.pc_fill[k0 + j] = 0; }
  }
}
SB_HD bool pl_peel_in_lds(const PlanCtx &c);
/* fill the patch CSC; seed the first frontier with the rows that already have one V column */
template <int Z> SB_HD void pl_pcsc_fill(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const rq_params &p = c.p;
  const uint32_t nl = c.job.nlost;
  for (uint32_t i = tid; i < sh->npatch; i += nt) {
    uint32_t row = i < nl ? p.S + p.H + c.lost[i] : p.L + (i - nl);
    const uint16_t *cols = c.patch_cols + (size_t)i * PL_PATCH_STRIDE;
    const uint32_t n = c.patch_len[i];
    for (uint32_t k0 = 0; k0 < n; k0 += PL_BATCH) { /* (a batch's places asked for together: entry by entry each was a trip of its own) */
      uint32_t col[PL_BATCH], base[PL_BATCH], pos[PL_BATCH];
#pragma unroll
      for (uint32_t j = 0; j < PL_BATCH; j++) col[j] = cols[k0 + j < n ? k0 + j : k0];
#pragma unroll
      for (uint32_t j = 0; j < PL_BATCH; j++) base[j] = c.pc_ptr[col[j]];
#pragma unroll
      for (uint32_t j = 0; j < PL_BATCH; j++) pos[j] = k0 + j < n ? PL_ATOM_ADD(&c.pc_fill[col[j]], 1u) : 0u;
#pragma unroll
      for (uint32_t j = 0; j < PL_BATCH; j++)
        if (k0 + j < n) {
          c.pc_rows[base[j] + pos[j]] = (uint16_t)row;
          if (pos[j] < PL_PCHEAD) c.pc_head[col[j] * PL_PCHEAD + pos[j]] = (uint16_t)row;
        }
    }
  }
  const bool stack2 = !pl_peel_in_lds(c);
  pl_for_batched(tid, nt, sh->M, [&](uint32_t r) { return c.rowstate[r] >> 24; }, [&](uint32_t r, uint32_t cnt) {
    /* (half of all rows start with two V columns: their places on the stack are taken a wave at a time -- 28 k single
     * increments of ONE LDS word were 0.8 M clocks at K'=56403) */
    const bool two = cnt == 2u && stack2;
    const uint32_t at = PL_WAVE_TAKE(&sh->ncand[0], two);
    if (cnt == 1u) {
      uint32_t j = PL_ATOM_ADD(&sh->nq[0], 1u);
      if (j < c.qcap) c.queue(0u)[j] = (uint16_t)r; else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    } else if (two) {
      c.cand[at] = (uint16_t)r; /* (a row enters the stack once: at most M entries) */
    }
  });
}

/* The peeling state (rowstate / rowinfo / colinfo) lives in LDS when it fits and in the block's HBM workspace
 * otherwise, so PlanCtx holds it behind generic pointers -- and every access through one is a FLAT instruction:
 * slower than a DS one, and ordered (vmcnt) behind the phase's outstanding global stores.  The peeling phases are
 * therefore compiled twice; the LDS instance tells the compiler where its three pointers point. */
struct PlPeel { uint32_t *rowstate, *rowinfo, *colinfo; };
template <bool LDS> SB_HD PlPeel pl_peel_state(const PlanCtx &c) {
  PlPeel s{c.rowstate, c.rowinfo, c.colinfo};
  if (LDS) { PL_ASSUME_LDS(s.rowstate); PL_ASSUME_LDS(s.rowinfo); PL_ASSUME_LDS(s.colinfo); }
  return s;
}
#define PL_PEEL_DISPATCH(fn, ...) do { if (pl_peel_in_lds(c)) fn<true>(__VA_ARGS__); else fn<false>(__VA_ARGS__); } while (0)
/* three ways: the state in LDS | the compact state in LDS with the HBM arrays kept up to date (fn##k) | the HBM arrays alone.
 * Z: template argument of the phase functions -- 0 compiles the compact form out (the kernel instance for blocks whose state
 * fits the LDS: with the extra code in it that instance ran 11 % slower, 3.19 -> 3.55 ms per 256 blocks of K=8192) */
#define PL_PEEL_DISPATCH3(fn, ...) do { if (PL_PEEL_LDS(Z)) fn##t<true>(__VA_ARGS__); else if (PL_PEEL_PK(Z)) fn##k(__VA_ARGS__); \
                                        else fn##t<false>(__VA_ARGS__); } while (0)
struct PlPk { uint32_t *cnt, *un, *pa, *vb; };
SB_HD PlPk pl_pk(const PlanCtx &c) {
  PlPk k{c.pk_cnt, c.pk_un, c.pk_pa, c.pk_vb};
  PL_ASSUME_LDS(k.cnt); PL_ASSUME_LDS(k.un); PL_ASSUME_LDS(k.pa); PL_ASSUME_LDS(k.vb);
  return k;
}
SB_HD uint32_t pk_count(const PlPk &k, uint32_t r) { return (k.cnt[r >> 2] >> ((r & 3u) * 8u)) & 0xFFu; }
SB_HD bool pk_bit(const uint32_t *b, uint32_t i) { return ((b[i >> 5] >> (i & 31u)) & 1u) != 0u; }

/* column `col` leaves V: one atomic subtract per row that contains it; rows that drop to a single V
 * column join the next frontier (queue of parity `np`).  `lvl1` (pivot level + 1) is folded into the
 * rows' level-so-far; 0 for an inactivated column.  A group of `lanes` lanes strides over the row list. */
template <bool LDS> SB_HD void pl_drop_column(PlanCtx &c, const PlPeel &s, uint32_t col, uint32_t lvl1, uint32_t np, uint32_t lane0, uint32_t lanes) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t dec = (1u << 24) | col;
  uint16_t *nextq = c.queue(np);
#ifdef PL_STAMP
  if (col + lvl1 != 0x7FFFFFF1u) PL_ST(c, 17);
  const uint32_t a = c.b_cptr[col];
  if (a != 0x7FFFFFF1u) PL_ST(c, 18);
  const uint32_t nb = c.b_cptr[col + 1 + (a == 0x7FFFFFF1u ? 1u : 0u)] - a;
#else
  const uint32_t a = c.b_cptr[col], nb = c.b_cptr[col + 1] - a;
#endif
#ifdef PL_STAMP
  if (a + nb != 0x7FFFFFF1u) PL_ST(c, 16);
  const uint32_t col2 = col + (a == 0xFFFFFFF1u ? 1u : 0u);
  const uint32_t pa = c.pc_ptr[col2], npc = c.pc_ptr[col2 + 1] - pa;
#else
  const uint32_t pa = c.pc_ptr[col], npc = c.pc_ptr[col + 1] - pa;
#endif
  if (nb + npc > lane0) PL_ST(c, 10);
  for (uint32_t e = lane0; e < nb + npc; e += lanes) {
    const bool base = e < nb;
    const uint32_t r = base ? c.b_ridx[a + e] : c.pc_rows[pa + (e - nb)];
    if (e == lane0 && r != 0xFFFFFFFFu) PL_ST(c, 11);
    const uint32_t info = s.rowinfo[r]; /* flags change in other phases only: stable here */
    if (base && (info & PL_PATCHED)) continue; /* base entry of a row this block replaced */
    if (e == lane0 && info != 0x12345u) PL_ST(c, 12);
    if (lvl1 && (info & PL_UNASSIGNED)) PL_ATOM_MAX(&s.rowinfo[r], (info & ~PL_LEVEL_MASK) | lvl1);
    const uint32_t old = PL_ATOM_SUB(&s.rowstate[r], dec);
    if (e == lane0 && old != 0x12345u) PL_ST(c, 13);
    if ((old >> 24) == 2u && (info & PL_UNASSIGNED)) {
      const uint32_t j = PL_ATOM_ADD(&sh->nq[np], 1u);
      if (PL_LIKELY(j < c.qcap)) nextq[j] = (uint16_t)r; else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    } else if (!LDS && (old >> 24) == 3u && (info & PL_UNASSIGNED)) { /* two V columns left: a candidate of the next inactivation */
      const uint32_t j = PL_ATOM_ADD(&sh->ncand[0], 1u);
      if (j < c.Mcap) c.cand[j] = (uint16_t)r; else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    }
  }
}

/* the same on the compact state: after the two trips for the column's row list everything the round waits for is in LDS --
 * the count (a byte of a word: the atomic subtracts 1 in the byte's place; a count never goes below zero, so nothing is
 * borrowed from the neighbour), the two flags; the HBM row state and level follow by atomics without a return value */
SB_HD void pl_drop_column_k(PlanCtx &c, const PlPk &k, uint32_t col, uint32_t lvl1, uint32_t np, uint32_t lane0, uint32_t lanes) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t dec = (1u << 24) | col;
  uint16_t *nextq = c.queue(np);
  const uint32_t a = c.b_cptr[col], nb = c.b_cptr[col + 1] - a;
  const uint32_t pa = c.pc_ptr[col], npc = c.pc_ptr[col + 1] - pa;
  for (uint32_t e = lane0; e < nb + npc; e += lanes) {
    const bool base = e < nb;
    const uint32_t r = base ? c.b_ridx[a + e] : c.pc_rows[pa + (e - nb)];
    const bool patched = pk_bit(k.pa, r), open = pk_bit(k.un, r); /* flags change in other phases only: stable here */
    if (base && patched) continue; /* base entry of a row this block replaced */
    if (lvl1 && open) PL_GMAX_NR(&c.rowinfo[r], PL_UNASSIGNED | (patched ? PL_PATCHED : 0u) | lvl1);
    PL_GSUB_NR(&c.rowstate[r], dec);
    const uint32_t sh8 = (r & 3u) * 8u;
    const uint32_t old = (PL_ATOM_SUB(&k.cnt[r >> 2], 1u << sh8) >> sh8) & 0xFFu;
    if (old == 2u && open) {
      const uint32_t j = PL_ATOM_ADD(&sh->nq[np], 1u);
      if (PL_LIKELY(j < c.qcap)) nextq[j] = (uint16_t)r; else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    } else if (old == 3u && open) { /* two V columns left: a candidate of the next inactivation */
      const uint32_t j = PL_ATOM_ADD(&sh->ncand[0], 1u);
      if (j < c.Mcap) c.cand[j] = (uint16_t)r; else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    }
  }
}

/* =============================== phase 1: peeling rounds ==================================== */
/* Round `rd`: frontier = queue[rd&1], next frontier = queue[(rd+1)&1].
 * A: every frontier row that still has exactly one V column tries to claim it (compare-and-swap on the
 *    column); the winner becomes a pivot at the level its earlier column drops accumulated. */
template <bool LDS, bool RING = false> SB_HD void pl_round_claim_t(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPeel s = pl_peel_state<LDS>(c);
  const uint32_t pq = rd & 1u;
  const uint16_t *fq = c.queue(pq);
  const uint32_t nf = sh->nq[pq] < c.qcap ? sh->nq[pq] : c.qcap, npiv0 = sh->npiv;
  if (nf) PL_ST(c, 3);
  for (uint32_t t = tid; t < nf; t += nt) {
    const uint32_t r = fq[t];
    const uint32_t st = s.rowstate[r], info = s.rowinfo[r];
    if ((st >> 24) != 1u || !(info & PL_UNASSIGNED)) continue;
    PL_ST(c, 4);
    const uint32_t col = st & 0xFFFFFFu;
    if (PL_ATOM_CAS(&s.colinfo[col], 0u, (PL_ST_CLAIM << 30) | r) != 0u) continue;
    PL_ST(c, 5);
    const uint32_t lv = info & PL_LEVEL_MASK;
    /* ONE counter per claim: the place in the round's claim list; the pivot number is that place behind the pivots of the
     * rounds before, and the drop phase adds the round's claims to the pivot count and takes them off the count of open
     * columns (three wave-aggregated atomics per claiming wave were ~35 instructions and a trip of its critical path) */
    const uint32_t i = PL_ATOM_ADD(&sh->nclaim[pq], 1u);
    const uint32_t k = npiv0 + i;
    PL_ST(c, 6);
    s.rowinfo[r] = (info & PL_PATCHED) | lv; /* assigned: bit 31 cleared */
    s.colinfo[col] = (PL_ST_PIVOT << 30) | k;
    if (PL_LIKELY(i < c.qcap)) {
      if (RING) c.ring()[i] = ((lv + 1u) << 16) | col;
      else { c.claim_l()[i] = (uint16_t)(lv + 1u); c.claim_c()[i] = (uint16_t)col; }
    } else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    c.pivslot[k] = (uint16_t)r; /* (HBM; read after peeling) */
    c.pivcol[k] = (uint16_t)col;
  }
  if (tid == 0) { sh->nq[pq ^ 1u] = 0; sh->best = PL_NONE; }
  PL_ST(c, 7);
}
/* compact state: whether a frontier row still has exactly one V column and no pivot is in LDS; WHICH column that is, is the
 * sum left in the HBM row state (all of last round's subtractions have arrived: the barrier waited for them) -- one trip,
 * together with the row's level so far; the claim itself is a bit in LDS */
SB_HD void pl_round_claim_k(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPk k = pl_pk(c);
  const uint32_t pq = rd & 1u;
  const uint16_t *fq = c.queue(pq);
  const uint32_t nf = sh->nq[pq] < c.qcap ? sh->nq[pq] : c.qcap, npiv0 = sh->npiv;
  for (uint32_t t = tid; t < nf; t += nt) {
    const uint32_t r = fq[t];
    if (pk_count(k, r) != 1u || !pk_bit(k.un, r)) continue;
    uint32_t st, info;
    pl_gload2(&c.rowstate[r], &c.rowinfo[r], st, info);
    const uint32_t col = st & 0xFFFFFFu, cbit = 1u << (col & 31u);
    if (PL_ATOM_OR(&k.vb[col >> 5], cbit) & cbit) continue; /* another row of this round took the column */
    const uint32_t lv = info & PL_LEVEL_MASK;
    const uint32_t i = PL_ATOM_ADD(&sh->nclaim[pq], 1u);
    const uint32_t kk = npiv0 + i; /* (pl_round_claim_t) */
    PL_ATOM_XOR(&k.un[r >> 5], 1u << (r & 31u)); /* assigned (the bit was set: only this thread clears it) */
    PL_GSTORE(&c.rowinfo[r], (info & PL_PATCHED) | lv);
    PL_GSTORE(&c.colinfo[col], (PL_ST_PIVOT << 30) | kk);
    if (PL_LIKELY(i < c.qcap)) { c.claim_l()[i] = (uint16_t)(lv + 1u); c.claim_c()[i] = (uint16_t)col; } else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    c.pivslot[kk] = (uint16_t)r; /* (HBM; read after peeling) */
    c.pivcol[kk] = (uint16_t)col;
  }
  if (tid == 0) { sh->nq[pq ^ 1u] = 0; sh->best = PL_NONE; }
}
SB_HD bool pl_chained(uint32_t Z);
template <int Z> SB_HD void pl_round_claim(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  if (pl_chained((uint32_t)Z)) { pl_round_claim_t<true, true>(c, rd, tid, nt); return; } /* (the chained drop phase takes the claims as words) */
  PL_PEEL_DISPATCH3(pl_round_claim_, c, rd, tid, nt);
}
#ifndef PL_DROP_LG_MAX
#define PL_DROP_LG_MAX 4u
#endif
/* B: the claimed columns leave V.  A group of 8..64 lanes per column -- as many as the round's claim count leaves
 * (most rounds claim a dozen columns; each trip over a column's row list is a dependent HBM/L2 round trip) */
template <bool LDS> SB_HD void pl_round_drop_t(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPeel s = pl_peel_state<LDS>(c);
  const uint32_t pq = rd & 1u;
  const uint32_t nc = sh->nclaim[pq] < c.qcap ? sh->nclaim[pq] : c.qcap;
  uint32_t lg = 3;
  while (lg < PL_DROP_LG_MAX && (nc << (lg + 1u)) <= nt) lg++;
  const uint32_t grp = tid >> lg, lane = tid & ((1u << lg) - 1u), ngrp = nt >> lg;
  if (nc) PL_ST(c, 9);
#ifdef PL_STAMP
  if (nc && grp + lane + ngrp != 0x7FFFFFF1u) PL_ST(c, 19);
#endif
  for (uint32_t i = grp; i < nc; i += ngrp) {
#ifdef PL_STAMP
    const uint32_t cc_ = c.claim_c()[i];
    if (cc_ != 0x7FFFFFF1u) PL_ST(c, 20);
    const uint32_t cl_ = c.claim_l()[i];
    if (cl_ != 0x7FFFFFF1u) PL_ST(c, 21);
    pl_drop_column<LDS>(c, s, cc_, cl_, pq ^ 1u, lane, 1u << lg);
#else
    pl_drop_column<LDS>(c, s, c.claim_c()[i], c.claim_l()[i], pq ^ 1u, lane, 1u << lg);
#endif
  }
  if (tid == 0) { sh->nclaim[pq ^ 1u] = 0; sh->npiv += nc; sh->nV -= nc; } /* (the round's claims: pl_round_claim_t) */
  PL_ST(c, 14);
}
SB_HD void pl_round_drop_k(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPk k = pl_pk(c);
  const uint32_t pq = rd & 1u;
  const uint32_t nc = sh->nclaim[pq] < c.qcap ? sh->nclaim[pq] : c.qcap;
  uint32_t lg = 3;
  while (lg < PL_DROP_LG_MAX && (nc << (lg + 1u)) <= nt) lg++;
  const uint32_t grp = tid >> lg, lane = tid & ((1u << lg) - 1u), ngrp = nt >> lg;
  for (uint32_t i = grp; i < nc; i += ngrp) pl_drop_column_k(c, k, c.claim_c()[i], c.claim_l()[i], pq ^ 1u, lane, 1u << lg);
  if (tid == 0) { sh->nclaim[pq ^ 1u] = 0; sh->npiv += nc; sh->nV -= nc; }
}
/* B, chained (peeling state in LDS, device): the claimed columns leave V, and a row that is left with ONE V column by that is
 * claimed on the spot -- by the lane whose subtraction brought it there -- and joins the list this phase is working through,
 * instead of waiting in the next frontier for the next round's claim phase.  The rounds between two inactivation events (a claim
 * phase and a drop phase each, two barriers and ~4.1 k clocks per dependency level: 643 rounds at K=8192) become ONE phase whose
 * chain is a claim's row list (two trips to L2), the subtraction, the compare-and-swap and the list entry: ~1.7 k clocks per
 * level.  A row's level is the maximum over its dropped columns as before (each drop folds its level into the row before it
 * subtracts, so the lane that brings the count to one sees them all), pivots are numbered by their place in the list, and who
 * wins a column two rows are left with is decided by the compare-and-swap as before: same levels, an equally valid plan.
 * List protocol: nclaim[pq] counts entries handed out; an entry is there when its column is not 0xFFFF (level first, column
 * last, by the one lane); group g of 16 lanes takes entries g, g + groups, ...; ndone counts entries dropped; the phase is over
 * when every entry handed out has been dropped (nobody is working, so nothing can be added).  The list is a ring of qcap
 * entries.  The emulator (threads one after the other) keeps the round form. */
#ifndef PL_CHAIN
#define PL_CHAIN 1
#endif
/* How the chained peel publishes a claim (pl_chain_claim): 1 = release / acquire atomics at workgroup scope, as the memory
 * model has it; 0 = plain volatile LDS accesses kept in place by a compiler barrier, resting on the LDS taking a wave's
 * instructions in order (what rounds 4 and 5 shipped).  Either way the plan check behind the peel (pl_check_*) catches a
 * claim that went wrong and sends the block to the host planner. */
#ifndef PL_CHAIN_RELEASE
#define PL_CHAIN_RELEASE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
/* (volatile accesses to LDS as DS instructions: through a generic pointer they become FLAT ones, whose waits count the
 * outstanding HBM stores as well -- a trip to memory on the chain for every claim) */
#define PL_VOL32(p) (*(volatile __attribute__((address_space(3))) uint32_t *)(uintptr_t)(p))
#define PL_VOL16(p) (*(volatile __attribute__((address_space(3))) uint16_t *)(uintptr_t)(p))
/* a row left with one V column by this lane's subtraction (old: what the subtraction found, now: the row's flags and level behind
 * it): claim that column, number the pivot, hand the column to the list */
__device__ __forceinline__ void pl_chain_claim(PlanCtx &c, const PlPeel &s, uint32_t r, uint32_t old, uint32_t now, uint32_t dec, uint32_t pq, uint32_t npiv0) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t col2 = (old - dec) & 0xFFFFFFu;
  if (PL_ATOM_CAS(&s.colinfo[col2], 0u, (PL_ST_CLAIM << 30) | r) != 0u) return; /* another row got the column first */
  const uint32_t lv = now & PL_LEVEL_MASK;
  const uint32_t i2 = PL_ATOM_ADD(&sh->nclaim[pq], 1u), k = npiv0 + i2;
  /* "assigned" before the list entry: whoever takes the entry meets this row in the column's list and must find it done (the
   * LDS takes a wave's instructions in order: no wait, but the compiler must keep the two where they are).  The entry before
   * everything else -- the chain waits for it; a slot that is not empty: the ring has come round on an entry nobody has dropped
   * yet (qcap entries in flight) */
#if PL_CHAIN_RELEASE
  /* the memory model's way of saying it: the row's "assigned" is a relaxed atomic store, the list entry a RELEASE at workgroup
   * scope (the reader's ring load is the acquire, pl_round_chain_dev) -- the compiler puts the wait for the LDS store in front
   * of the compare-and-swap.  Measured against the in-order form below: profiles/r6_planner_release.txt. */
  __hip_atomic_store((__attribute__((address_space(3))) uint32_t *)(uintptr_t)&s.rowinfo[r], (now & PL_PATCHED) | lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  uint32_t was = PL_RING_EMPTY;
  (void)__hip_atomic_compare_exchange_strong((__attribute__((address_space(3))) uint32_t *)(uintptr_t)&c.ring()[i2 & (c.qcap - 1u)], &was,
                                             ((lv + 1u) << 16) | col2, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  PL_VOL32(&s.rowinfo[r]) = (now & PL_PATCHED) | lv;
  __asm__ volatile("" ::: "memory");
  const uint32_t was = PL_ATOM_CAS(&c.ring()[i2 & (c.qcap - 1u)], PL_RING_EMPTY, ((lv + 1u) << 16) | col2);
#endif
  s.colinfo[col2] = (PL_ST_PIVOT << 30) | k;
  c.pivslot[k] = (uint16_t)r; /* (HBM; read after peeling) */
  c.pivcol[k] = (uint16_t)col2;
  if (was != PL_RING_EMPTY) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
#ifdef PL_STAMP
  if (c.st_blk) atomicAdd(&sh->st_acc[27], 1ull);
#endif
}
/* one row of a column that leaves V (the chained form of pl_drop_column's loop body).  Four LDS instructions, two of them waited
 * for (the flags; the subtraction with the level behind it) -- and a claim's three atomics, each waited for: measured on the
 * stamped build (-DPL_STAMP) a read costs ~130 clocks, an atomic with a result ~300, and these trips, not the one to memory
 * (~350 for the rows), are the chain: ~2.2 k clocks from taking an entry to having dropped it at K=8192.  (Base and patch row of a
 * lane side by side, one pass for both, was no faster: the second row's instructions are issued for every column then.) */
__device__ __forceinline__ void pl_chain_row(PlanCtx &c, const PlPeel &s, uint32_t r, bool base, uint32_t dec, uint32_t lvl1, uint32_t pq, uint32_t npiv0) {
#ifdef PL_STAMP
  const bool st_ = c.st_blk && (threadIdx.x & 15u) == 0u;
  const unsigned long long ta_ = clock64();
#endif
  const uint32_t info = PL_VOL32(&s.rowinfo[r]);
#ifdef PL_STAMP
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long tb_ = clock64();
  if (st_) atomicAdd(&c.sh->st_acc[30], tb_ - ta_);
#endif
  if (base && (info & PL_PATCHED)) return; /* base entry of a row this block replaced */
  if (!(info & PL_UNASSIGNED)) { (void)PL_ATOM_SUB(&s.rowstate[r], dec); return; } /* (the column's own pivot row, or a row that has its pivot) */
  if (lvl1) PL_ATOM_MAX(&s.rowinfo[r], (info & ~PL_LEVEL_MASK) | lvl1);
  /* the subtraction and, behind it in the LDS queue, the row's level as every earlier drop left it: one trip for both */
  const uint32_t old = PL_ATOM_SUB(&s.rowstate[r], dec);
  const uint32_t now = PL_VOL32(&s.rowinfo[r]);
#ifdef PL_STAMP
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long tc_ = clock64();
  if (st_) atomicAdd(&c.sh->st_acc[31], tc_ - tb_);
#endif
  /* one V column left, and every other column's level is in the row by now: claim it */
  if ((old >> 24) == 2u) pl_chain_claim(c, s, r, old, now, dec, pq, npiv0);
#ifdef PL_STAMP
  if (st_) atomicAdd(&c.sh->st_acc[22], (unsigned long long)clock64() - tc_);
#endif
}
/* column `col` leaves V, by a group of 16 lanes: the rows come from the column's head in the constants (NRQ_CHEAD = 16 entries:
 * the lane's own, ONE trip -- pointer-then-list is two, on a chain that is little else) and, for a column the block's patch rows
 * touch (a bit in LDS says so), from the block's patch heads with the column's count beside them, in flight with the first.
 * Columns with more than 16 base rows or PL_PCHEAD patch rows (a few per block) go on through the pointers. */
__device__ __forceinline__ void pl_drop_column_chain(PlanCtx &c, const PlPeel &s, uint32_t col, uint32_t lvl1, uint32_t pq, uint32_t npiv0, uint32_t lane0) {
  const uint32_t dec = (1u << 24) | col;
  const uint16_t *hd = c.b_chead + (size_t)col * NRQ_CHEAD;
  const uint32_t r0 = hd[lane0], last = hd[NRQ_CHEAD - 1u];
  uint32_t pcnt = 0, r1 = 0xFFFFu;
  const uint32_t *pcb = c.pcbits();
  if (!pcb || ((pcb[col >> 5] >> (col & 31u)) & 1u)) { /* (per block: far away) */
    pcnt = c.pcfill_lds ? (uint32_t)PL_VOL32(&c.pc_fill[col]) : c.pc_fill[col]; /* (a DS read when the counts are in LDS: pl_ctx_setup) */
    if (lane0 < PL_PCHEAD) r1 = c.pc_head[col * PL_PCHEAD + lane0];
  }
#ifdef PL_STAMP
  const unsigned long long t0_ = clock64();
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (c.st_blk && lane0 == 0u) atomicAdd(&c.sh->st_acc[26], (unsigned long long)clock64() - t0_);
#endif
  if (r0 != 0xFFFFu) pl_chain_row(c, s, r0, true, dec, lvl1, pq, npiv0);
  if (lane0 < PL_PCHEAD && lane0 < pcnt) pl_chain_row(c, s, r1, false, dec, lvl1, pq, npiv0);
  if (last != 0xFFFFu) { /* (the head is full: there may be more) */
    const uint32_t a = c.b_cptr[col], nb = c.b_cptr[col + 1] - a;
    for (uint32_t e = NRQ_CHEAD + lane0; e < nb; e += 16u) pl_chain_row(c, s, c.b_ridx[a + e], true, dec, lvl1, pq, npiv0);
  }
  if (pcnt > PL_PCHEAD) {
    const uint32_t pa = c.pc_ptr[col];
    for (uint32_t e = PL_PCHEAD + lane0; e < pcnt; e += 16u) pl_chain_row(c, s, c.pc_rows[pa + e], false, dec, lvl1, pq, npiv0);
  }
}
__device__ __forceinline__ void pl_round_chain_dev(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPeel s = pl_peel_state<true>(c);
  const uint32_t pq = rd & 1u, npiv0 = sh->npiv;
  constexpr uint32_t LG = 4u;
  /* (consecutive entries to different waves: a wave's four groups move in lockstep, and an entry that arrives just after its
   * neighbour would wait out the wave's whole trip) */
  const uint32_t nwave = nt >> 6, grp = (tid >> 6) + nwave * ((tid >> LG) & 3u), lane = tid & ((1u << LG) - 1u), ngrp = nt >> LG;
  const uint32_t qm = c.qcap - 1u; /* (a power of two: 256 .. 2048, nrq_plan_launch) */
  uint32_t i = grp, idle = 0;
#ifdef PL_STAMP
  const unsigned long long tp_ = clock64();
#endif
  for (uint32_t guard = 0; guard < (1u << 24); guard++) {
#if PL_CHAIN_RELEASE
    const uint32_t v = __hip_atomic_load((__attribute__((address_space(3))) uint32_t *)(uintptr_t)&c.ring()[i & qm], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    const uint32_t v = PL_VOL32(&c.ring()[i & qm]); /* (the group's next entry: there as soon as its claimant has written it) */
#endif
    if (v != PL_RING_EMPTY) {
#ifdef PL_STAMP
      const unsigned long long t0_ = clock64();
#endif
      pl_drop_column_chain(c, s, v & 0xFFFFu, v >> 16, pq, npiv0, lane);
#ifdef PL_STAMP
      if (c.st_blk && lane == 0u) { atomicAdd(&sh->st_acc[24], 1ull); atomicAdd(&sh->st_acc[25], (unsigned long long)clock64() - t0_); }
#endif
      if (lane == 0u) { PL_VOL32(&c.ring()[i & qm]) = PL_RING_EMPTY; (void)PL_ATOM_ADD(&sh->ndone, 1u); }
      i += ngrp; idle = 0;
      continue;
    }
    if ((++idle & 3u) != 0u) continue;
    /* nothing for this group for a while: over when everything handed out has been dropped (then nobody can add anything) */
    const uint32_t d = PL_VOL32(&sh->ndone), n = PL_VOL32(&sh->nclaim[pq]);
    if ((d == n && i >= n) || PL_VOL32(&sh->status)) break;
    __builtin_amdgcn_s_sleep(1);
  }
#ifdef PL_STAMP
  if (c.st_blk && tid == 0u) { atomicAdd(&sh->st_acc[28], (unsigned long long)clock64() - tp_); atomicAdd(&sh->st_acc[29], 1ull); }
#endif
}
#endif
template <int Z> SB_HD void pl_round_drop(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (PL_CHAIN && Z == 0) { pl_round_chain_dev(c, rd, tid, nt); return; } /* (the instance for peeling state in LDS; the books: pl_round_chain_end) */
#endif
  PL_PEEL_DISPATCH3(pl_round_drop_, c, rd, tid, nt);
}
/* behind the barrier that ends a chained drop phase: the round's claims (the claim phase's and the chained ones) into the counts */
template <int Z> SB_HD void pl_round_chain_end(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid != 0) return;
  const uint32_t pq = rd & 1u, nc = sh->nclaim[pq];
  if (sh->ndone != nc && !sh->status) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); /* (the phase gave up on an entry that never came: cannot happen; the host planner takes the block) */
  sh->nclaim[pq ^ 1u] = 0; sh->npiv += nc; sh->nV -= nc; sh->ndone = 0;
  (void)nt;
}
SB_HD bool pl_chained(uint32_t Z) {
#if defined(__HIP_DEVICE_COMPILE__)
  return PL_CHAIN && Z == 0u;
#else
  (void)Z;
  return false;
#endif
}
#define PL_EVENT_ROWS NRQ_MULTI_INACT
/* (small blocks: fewer rows an event -- with six at once a block of a few hundred symbols ends up with more inactive columns than
 * row after row, K=256: the decode launch lost a workgroup per CU to the larger dense stage) */
SB_HD uint32_t pl_event_rows(const PlanCtx &c) { const uint32_t n = c.p.W / 128u; return n < 2u ? 2u : n < PL_EVENT_ROWS ? n : PL_EVENT_ROWS; }
/* all but one V column of row r leave V for the inactive set: the one with the fewest entries stays (heuristic) */
template <bool LDS> SB_HD void pl_event_row(PlanCtx &c, const PlPeel &s, uint32_t r, uint32_t pq) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint16_t *cols;
  const uint32_t n = pl_row(c, r, &cols);
  uint32_t keep = PL_NONE, keepdeg = PL_NONE;
  constexpr uint32_t CB = 8; /* (eight entries at a time with all their loads in flight together -- entry, column state, four list bounds) */
  for (uint32_t k0 = 0; k0 < n; k0 += CB) {
    uint32_t col[CB], inf[CB], dg[CB];
#pragma unroll
    for (uint32_t q = 0; q < CB; q++) col[q] = k0 + q < n ? cols[k0 + q] : 0u;
#pragma unroll
    for (uint32_t q = 0; q < CB; q++) {
      inf[q] = s.colinfo[col[q]];
      dg[q] = (c.b_cptr[col[q] + 1] - c.b_cptr[col[q]]) + (c.pc_ptr[col[q] + 1] - c.pc_ptr[col[q]]);
    }
#pragma unroll
    for (uint32_t q = 0; q < CB; q++)
      if (k0 + q < n && inf[q] == 0u && dg[q] < keepdeg) { keepdeg = dg[q]; keep = col[q]; }
  }
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t col = cols[k];
    if (col == keep || s.colinfo[col] != 0u) continue;
    if (PL_ATOM_CAS(&s.colinfo[col], 0u, (PL_ST_CLAIM << 30) | r) != 0u) continue; /* (another row of the event took it) */
    const uint32_t x = c.p.P + PL_ATOM_ADD(&sh->ninact, 1u), m = PL_ATOM_ADD(&sh->nclaim[pq], 1u);
    if (x >= c.ucap || m >= c.qcap) { (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); continue; }
    s.colinfo[col] = (PL_ST_INACT << 30) | x;
    c.ucol[x] = (uint16_t)col;
    c.claim_c()[m] = (uint16_t)col; /* "columns to drop" */
    (void)PL_ATOM_SUB(&sh->nV, 1u);
  }
}
/* the same on the compact state: "in V" is a bit in LDS (its atomic OR is the claim), the column's state word in HBM is stored */
SB_HD void pl_event_row_k(PlanCtx &c, const PlPk &k, uint32_t r, uint32_t pq) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint16_t *cols;
  const uint32_t n = pl_row(c, r, &cols);
  uint32_t keep = PL_NONE, keepdeg = PL_NONE;
  constexpr uint32_t CB = 8;
  for (uint32_t k0 = 0; k0 < n; k0 += CB) {
    uint32_t col[CB], dg[CB];
#pragma unroll
    for (uint32_t q = 0; q < CB; q++) col[q] = k0 + q < n ? cols[k0 + q] : 0u;
#pragma unroll
    for (uint32_t q = 0; q < CB; q++)
      dg[q] = (c.b_cptr[col[q] + 1] - c.b_cptr[col[q]]) + (c.pc_ptr[col[q] + 1] - c.pc_ptr[col[q]]);
#pragma unroll
    for (uint32_t q = 0; q < CB; q++)
      if (k0 + q < n && !pk_bit(k.vb, col[q]) && dg[q] < keepdeg) { keepdeg = dg[q]; keep = col[q]; }
  }
  for (uint32_t e = 0; e < n; e++) {
    const uint32_t col = cols[e], bit = 1u << (col & 31u);
    if (col == keep || pk_bit(k.vb, col)) continue;
    if (PL_ATOM_OR(&k.vb[col >> 5], bit) & bit) continue; /* (another row of the event took it; also: a column listed twice) */
    const uint32_t x = c.p.P + PL_ATOM_ADD(&sh->ninact, 1u), m = PL_ATOM_ADD(&sh->nclaim[pq], 1u);
    if (x >= c.ucap || m >= c.qcap) { (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); continue; }
    c.colinfo[col] = (PL_ST_INACT << 30) | x;
    c.ucol[x] = (uint16_t)col;
    c.claim_c()[m] = (uint16_t)col; /* "columns to drop" */
    (void)PL_ATOM_SUB(&sh->nV, 1u);
  }
}
/* No claimant in the frontier: find an open row with the fewest V columns.  Almost always that is a row with two; those
 * are kept on a stack as they come up (pl_drop_column pushes a row when its count drops to two; rows that start with
 * two are pushed by pl_pcsc_fill), and the search looks at the top nt entries only -- the rows that reached two most
 * recently, i.e. the ones most likely still open -- instead of all M rows (K'=56403: 41 k clocks per search, 264
 * searches; half of all rows start with two V columns, so a list that is scanned whole is no better).  Entries above
 * the highest valid one are popped; a chunk without a valid entry is popped whole and the search repeats
 * (planner_seq.h).  Only when the stack runs empty are all rows scanned (pl_inact_find_b). */
template <bool LDS> SB_HD void pl_inact_find_t(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPeel s = pl_peel_state<LDS>(c);
  const uint32_t rd = rdrep & 0xFFFFFFu, rep = rdrep >> 24; /* rep-th row of this inactivation event */
  const uint32_t n = sh->ncand[0], lo = n > nt ? n - nt : 0u, i = lo + tid;
  uint32_t best = PL_NONE, hi = 0;
  bool valid = false;
  uint32_t vr = 0;
  if (i < n) {
    const uint32_t r = c.cand[i];
    if ((s.rowinfo[r] & PL_UNASSIGNED) && (s.rowstate[r] >> 24) == 2u) { best = (2u << 16) | r; hi = i + 1u; valid = true; vr = r; }
  }
  { /* (the event's rows: the first few valid entries, pl_event_rows() of them are used -- pl_inact_apply_a) */
    const uint32_t at = PL_WAVE_TAKE(&sh->ev_n, valid);
    if (valid && at < PL_EVENT_ROWS) c.queue(rd & 1u)[at] = (uint16_t)vr;
  }
  best = PL_WAVE_MIN(best);
  hi = PL_WAVE_MAX(hi);
  if (best != PL_NONE && PL_WAVE_LEADER(tid)) { PL_ATOM_MIN(&sh->best, best); PL_ATOM_MAX(&sh->ncand[1], hi); }
  if (tid == 0 && rep == 0) { sh->nq[(rd & 1u) ^ 1u] = 0; sh->nclaim[rd & 1u] = 0; }
}
SB_HD void pl_inact_find_k(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPk k = pl_pk(c);
  const uint32_t rd = rdrep & 0xFFFFFFu, rep = rdrep >> 24;
  const uint32_t n = sh->ncand[0], lo = n > nt ? n - nt : 0u, i = lo + tid;
  uint32_t best = PL_NONE, hi = 0;
  bool valid = false;
  uint32_t vr = 0;
  if (i < n) {
    const uint32_t r = c.cand[i];
    if (pk_bit(k.un, r) && pk_count(k, r) == 2u) { best = (2u << 16) | r; hi = i + 1u; valid = true; vr = r; }
  }
  { /* (the event's rows: pl_inact_find_t) */
    const uint32_t at = PL_WAVE_TAKE(&sh->ev_n, valid);
    if (valid && at < PL_EVENT_ROWS) c.queue(rd & 1u)[at] = (uint16_t)vr;
  }
  best = PL_WAVE_MIN(best);
  hi = PL_WAVE_MAX(hi);
  if (best != PL_NONE && PL_WAVE_LEADER(tid)) { PL_ATOM_MIN(&sh->best, best); PL_ATOM_MAX(&sh->ncand[1], hi); }
  if (tid == 0 && rep == 0) { sh->nq[(rd & 1u) ^ 1u] = 0; sh->nclaim[rd & 1u] = 0; }
}
template <int Z> SB_HD void pl_inact_find(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  PL_PEEL_DISPATCH3(pl_inact_find_, c, rdrep, tid, nt);
}
/* pop what the search found invalid: everything above the highest valid entry, or the whole chunk */
template <int Z> SB_HD void pl_inact_find_c(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid != 0) return;
  const uint32_t n = sh->ncand[0];
  sh->ncand[0] = sh->best != PL_NONE ? sh->ncand[1] : (n > nt ? n - nt : 0u);
  sh->ncand[1] = 0;
  (void)rdrep;
}
/* the stack is empty and nothing was found: the sparsest of all open rows (workgroup-wide atomic min) */
template <bool LDS> SB_HD void pl_inact_find_b_t(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPeel s = pl_peel_state<LDS>(c);
  uint32_t best = PL_NONE;
  /* (a batch of rows' words asked for together: row by row, flags then count, each was a trip of its own -- 3.8 k clocks a search
   * for nine rows per thread at K=8192, 78 searches) */
  struct IS { uint32_t info, st; };
  pl_for_batched(tid, nt, sh->M, [&](uint32_t r) { return IS{s.rowinfo[r], s.rowstate[r]}; }, [&](uint32_t r, IS v) {
    const uint32_t cnt = v.st >> 24;
    if ((v.info & PL_UNASSIGNED) && cnt >= 2u) {
      const uint32_t key = (cnt << 16) | r; /* (ties by level-so-far give 424 -> 312 levels at the same u, but the kernel is
                                              * 2.5 % slower for it: measured, not adopted) */
      if (key < best) best = key;
    }
  });
  best = PL_WAVE_MIN(best);
  if (best != PL_NONE && PL_WAVE_LEADER(tid)) PL_ATOM_MIN(&sh->best, best);
  (void)rdrep;
}
SB_HD void pl_inact_find_b_k(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPk k = pl_pk(c);
  uint32_t best = PL_NONE;
  for (uint32_t r = tid; r < sh->M; r += nt) {
    if (!pk_bit(k.un, r)) continue;
    const uint32_t cnt = pk_count(k, r);
    if (cnt >= 2u) {
      const uint32_t key = (cnt << 16) | r;
      if (key < best) best = key;
    }
  }
  best = PL_WAVE_MIN(best);
  if (best != PL_NONE && PL_WAVE_LEADER(tid)) PL_ATOM_MIN(&sh->best, best);
  (void)rdrep;
}
template <int Z> SB_HD void pl_inact_find_b(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  PL_PEEL_DISPATCH3(pl_inact_find_b_, c, rdrep, tid, nt);
}
/* inactivate all but one V column of that row (or every remaining V column if no row is left) */
template <bool LDS> SB_HD void pl_inact_apply_a_t(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPeel s = pl_peel_state<LDS>(c);
  const rq_params &p = c.p;
  const uint32_t rd = rdrep & 0xFFFFFFu, rep = rdrep >> 24;
  if (sh->best == PL_NONE) {
    if (rep != 0) return; /* later rows of an event: nothing left to take, peeling resumes */
    for (uint32_t col = tid; col < p.W; col += nt) {
      if (s.colinfo[col] == 0u) {
        const uint32_t x = p.P + PL_ATOM_ADD(&sh->ninact, 1u);
        if (x < c.ucap) { s.colinfo[col] = (PL_ST_INACT << 30) | x; c.ucol[x] = (uint16_t)col; }
        else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
      }
    }
    return;
  }
  /* the event's rows (pl_inact_find listed them; none listed: the sparsest open row), a thread each: all but one V column of a
   * row go inactive -- pl_event_row; the columns end up in the claim list for pl_inact_apply_b */
  const uint32_t n2 = sh->ev_n < pl_event_rows(c) ? sh->ev_n : pl_event_rows(c);
  if (n2) { if (tid < n2) pl_event_row<LDS>(c, s, c.queue(rd & 1u)[tid], rd & 1u); }
  else if (tid == 0) pl_event_row<LDS>(c, s, sh->best & 0xFFFFu, rd & 1u);
  (void)p;
}
SB_HD void pl_inact_apply_a_k(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPk k = pl_pk(c);
  const rq_params &p = c.p;
  const uint32_t rd = rdrep & 0xFFFFFFu, rep = rdrep >> 24;
  if (sh->best == PL_NONE) {
    if (rep != 0) return; /* later rows of an event: nothing left to take, peeling resumes */
    for (uint32_t col = tid; col < p.W; col += nt) {
      if (!pk_bit(k.vb, col)) {
        const uint32_t x = p.P + PL_ATOM_ADD(&sh->ninact, 1u);
        if (x < c.ucap) { PL_ATOM_OR(&k.vb[col >> 5], 1u << (col & 31u)); c.colinfo[col] = (PL_ST_INACT << 30) | x; c.ucol[x] = (uint16_t)col; }
        else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
      }
    }
    return;
  }
  const uint32_t n2 = sh->ev_n < pl_event_rows(c) ? sh->ev_n : pl_event_rows(c); /* (pl_inact_apply_a_t) */
  if (n2) { if (tid < n2) pl_event_row_k(c, k, c.queue(rd & 1u)[tid], rd & 1u); }
  else if (tid == 0) pl_event_row_k(c, k, sh->best & 0xFFFFu, rd & 1u);
  (void)p;
}
template <int Z> SB_HD void pl_inact_apply_a(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  PL_PEEL_DISPATCH3(pl_inact_apply_a_, c, rdrep, tid, nt);
}
template <bool LDS> SB_HD void pl_inact_apply_b_t(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPeel s = pl_peel_state<LDS>(c);
  const uint32_t rd = rdrep & 0xFFFFFFu, rep = rdrep >> 24;
  const uint32_t pq = rd & 1u;
  if (sh->best == PL_NONE) {
    if (tid == 0 && rep == 0) sh->nV = 0;
    if (tid == 0) sh->tmp1 = 1u; /* event over */
    return;
  }
  if (sh->status) return; /* (a capacity gave out while the columns were listed: the list has holes) */
  const uint32_t grp = tid >> 5, lane = tid & 31u, ngrp = nt >> 5;
  for (uint32_t i = grp; i < sh->nclaim[pq]; i += ngrp) {
    pl_drop_column<LDS>(c, s, c.claim_c()[i], 0u, pq ^ 1u, lane, 32u);
#if defined(__HIP_DEVICE_COMPILE__) /* (the 32 lanes of a group run in lockstep there; the emulator's threads run one after the other and never chain) */
    if (lane == 0u) c.claim_c()[i] = 0xFFFFu; /* (a list entry that is not "no entry" is one pl_round_chain may take) */
#endif
  }
  if (tid == 0) { sh->nclaim[pq ^ 1u] = 0; sh->tmp1 = 0u; sh->ev_n = 0; }
}
SB_HD void pl_inact_apply_b_k(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const PlPk k = pl_pk(c);
  const uint32_t rd = rdrep & 0xFFFFFFu, rep = rdrep >> 24;
  const uint32_t pq = rd & 1u;
  if (sh->best == PL_NONE) {
    if (tid == 0 && rep == 0) sh->nV = 0;
    if (tid == 0) sh->tmp1 = 1u; /* event over */
    return;
  }
  if (sh->status) return;
  const uint32_t grp = tid >> 5, lane = tid & 31u, ngrp = nt >> 5;
  for (uint32_t i = grp; i < sh->nclaim[pq]; i += ngrp) pl_drop_column_k(c, k, c.claim_c()[i], 0u, pq ^ 1u, lane, 32u);
  if (tid == 0) { sh->nclaim[pq ^ 1u] = 0; sh->tmp1 = 0u; sh->ev_n = 0; }
}
template <int Z> SB_HD void pl_inact_apply_b(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  PL_PEEL_DISPATCH3(pl_inact_apply_b_, c, rdrep, tid, nt);
}
/* between two rows of one event: forget the previous choice */
template <int Z> SB_HD void pl_inact_next(PlanCtx &c, uint32_t rdrep, uint32_t tid, uint32_t nt) {
  if (tid == 0) { c.sh->best = PL_NONE; c.sh->nclaim[rdrep & 1u] = 0; }
}

/* ---- one inactivation event in three phases (peeling state in LDS) ----
 * Row after row -- search, choose, drop, a handful of barriers each, up to NRQ_MULTI_INACT rows an event -- an event was ~60 k
 * clocks of mostly barriers (78 rows in 14 events at K=8192: 0.86 M clocks).  Here the search lists up to PL_EVENT_ROWS open
 * rows with two V columns in one pass (which of the thousands there are does not matter: whichever waves come first), one
 * thread per listed row walks its row and inactivates all but one of its V columns (a compare-and-swap on the column: two rows
 * may hold the same one), and one drop phase takes all those columns out of V.  Rows of one event almost never share a column
 * (two of ~5000); when they do, a column more than necessary may go inactive, which costs nothing but that column. */
template <int Z> SB_HD void pl_event_scan(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (!PL_PEEL_LDS(Z)) return;
  const PlPeel s = pl_peel_state<true>(c);
  const uint32_t pq = rd & 1u;
  uint16_t *evrow = c.queue(pq); /* (the frontier is empty: that is why there is an event) */
  uint32_t best = PL_NONE;
  struct IS { uint32_t info, st; };
  pl_for_batched(tid, nt, sh->M, [&](uint32_t r) { return IS{s.rowinfo[r], s.rowstate[r]}; }, [&](uint32_t r, IS v) {
    const uint32_t cnt = v.st >> 24;
    const bool open = (v.info & PL_UNASSIGNED) && cnt >= 2u, two = open && cnt == 2u;
    const uint32_t at = PL_WAVE_TAKE(&sh->ev_n, two);
    if (two && at < PL_EVENT_ROWS) evrow[at] = (uint16_t)r; /* (pl_event_pick takes pl_event_rows() of them) */
    if (open && !two) { const uint32_t key = (cnt << 16) | r; if (key < best) best = key; }
  });
  best = PL_WAVE_MIN(best);
  if (best != PL_NONE && PL_WAVE_LEADER(tid)) PL_ATOM_MIN(&sh->ev_min, best);
  if (tid == 0) { sh->nq[pq ^ 1u] = 0; sh->nclaim[pq] = 0; }
}
template <int Z> SB_HD void pl_event_pick(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (!PL_PEEL_LDS(Z)) return;
  const PlPeel s = pl_peel_state<true>(c);
  const uint32_t pq = rd & 1u;
  const uint32_t n2 = sh->ev_n < pl_event_rows(c) ? sh->ev_n : pl_event_rows(c), lone = sh->ev_min;
  if (n2 == 0u && lone == PL_NONE) { /* no open row is left: what remains of V goes inactive, and peeling is over */
    for (uint32_t col = tid; col < c.p.W; col += nt) {
      if (s.colinfo[col] == 0u) {
        const uint32_t x = c.p.P + PL_ATOM_ADD(&sh->ninact, 1u);
        if (x < c.ucap) { s.colinfo[col] = (PL_ST_INACT << 30) | x; c.ucol[x] = (uint16_t)col; }
        else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
      }
    }
    if (tid == 0) sh->ev_none = 1u;
    return;
  }
  if (n2) { if (tid < n2) pl_event_row<true>(c, s, c.queue(pq)[tid], pq); }
  else if (tid == 0) pl_event_row<true>(c, s, lone & 0xFFFFu, pq); /* (no row with two: the sparsest one) */
}
template <int Z> SB_HD void pl_event_drop(PlanCtx &c, uint32_t rd, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (!PL_PEEL_LDS(Z)) return;
  const PlPeel s = pl_peel_state<true>(c);
  const uint32_t pq = rd & 1u;
  if (sh->status) return; /* (a capacity gave out while the columns were listed: the list has holes, and the block goes to the host planner) */
  if (sh->ev_none) { if (tid == 0) sh->nV = 0; return; }
  const uint32_t grp = tid >> 5, lane = tid & 31u, ngrp = nt >> 5, nc = sh->nclaim[pq] < c.qcap ? sh->nclaim[pq] : c.qcap;
  for (uint32_t i = grp; i < nc; i += ngrp) {
    pl_drop_column<true>(c, s, c.claim_c()[i], 0u, pq ^ 1u, lane, 32u);
#if defined(__HIP_DEVICE_COMPILE__) /* (the 32 lanes of a group run in lockstep there; the emulator's threads run one after the other and never chain) */
    if (lane == 0u) c.claim_c()[i] = 0xFFFFu; /* (a list entry that is not "no entry" is one pl_round_chain may take) */
#endif
  }
  if (tid == 0) { sh->nclaim[pq ^ 1u] = 0; sh->ev_n = 0; sh->ev_min = PL_NONE; }
}

/* =============================== phase 2: levels, W ========================================== */
/* Per-group counters of the op stream.  Group l of the stream holds
 *   - the "finishing" ops of level l: dst on level l, src on level l-1 (every row of the level has at least one), and
 *   - "early" ops: dst on a later level t > l, src final before l; such an op may run in any group of its window
 *     [level(src)+1, t-1] and is sent to one of them by a hash of (dst, column) -- it fills lanes that the narrow
 *     levels would leave empty (plan.h: a group needs NRQ_PIPE-1 rows after its finishing ops anyway).
 * Sixteen 16-bit counters per group -- finishing / early ops counted, by lane class of the target (plan.h "lane
 * placement") -- in LDS at the end of the aux region (the rowstate image, dead once peeling is over): thousands of rows
 * share a level, and that many atomics on ONE global address serialise in L2 (~1 M clocks per block measured).  The level
 * of every pivot column (what an op's window depends on) sits at the start of the aux region.  If LDS is too small for
 * either, every op counts as finishing op of its dst level, with the counters in HBM (lev_ops / lev_fill), one atomic
 * per row, not per op, and lanes handed out in counter order. */
SB_HD uint32_t pl_lev_words(const PlanCtx &c) { return c.sh->nlev + 2u; }
#define PL_CLS_BYTES (2u * NRQ_LANE_CLASSES * 2u) /* per group: [part 0 = finishing, 1 = early][class] x u16 */
static_assert(PL_CLS_BYTES == 32u, "pl_work_plan sizes the workspace copy of the class counters with 32 bytes per group");
/* Where the class counters go: the dense stage's region when the peeling state has its own place in LDS (that region is
 * idle until the HDPC fold), else the aux region -- behind the column levels if those fit there too.  Big blocks (peeling
 * state in the workspace, L * 2 bytes of levels beyond the LDS) keep the column levels in HBM: in the stack of open rows,
 * dead once peeling is over -- a trip to L2 per entry, with PL_WU entries in flight, instead of the row walks and
 * same-address global atomics of the path without counters. */
SB_HD bool pl_collev_in_lds(const PlanCtx &c) {
  if (c.collev_hbm) return false;
  const uint32_t bytes = pl_r16(pl_lev_words(c) * PL_CLS_BYTES), lv = pl_r16(c.p.L * 2u);
  if (!c.aux_lds || lv > c.aux_bytes) return false;
  if (c.dense_lds != c.aux_lds && bytes <= c.dense_bytes) return true; /* (the counters are elsewhere) */
  return lv + bytes <= c.aux_bytes;
}
SB_HD uint8_t *pl_cls_place(const PlanCtx &c) {
  const uint32_t bytes = pl_r16(pl_lev_words(c) * PL_CLS_BYTES);
  if (!c.aux_lds) return nullptr;
  if (c.dense_lds != c.aux_lds && bytes <= c.dense_bytes) return c.dense_lds;
  if (bytes <= c.aux_bytes) return c.aux_lds + c.aux_bytes - bytes; /* (the levels, if they are here too, lie at the start: pl_collev_in_lds) */
  return nullptr;
}
SB_HD uint16_t *pl_col_level(const PlanCtx &c) {
  if (!pl_cls_place(c)) return nullptr;
  if (pl_collev_in_lds(c)) return reinterpret_cast<uint16_t *>(c.aux_lds); /* (the level-by-level W pass stages its tables in the aux region later) */
  return c.cand; /* Mcap >= L entries */
}
/* the class counters (nullptr: the HBM counters are in use) */
SB_HD uint32_t *pl_cls(const PlanCtx &c) {
  if (c.cls_glob) return pl_cls_place(c) ? c.cls_glob : nullptr;
  uint32_t *q = reinterpret_cast<uint32_t *>(pl_cls_place(c));
  if (!q) return nullptr;
  PL_ASSUME_LDS(q);
  return q;
}
/* count one op of (group g, part, class d): its rank among them */
SB_HD uint32_t pl_cls_take(uint32_t *cls, uint32_t g, uint32_t part, uint32_t d) {
  const uint32_t i = (g * 2u + part) * NRQ_LANE_CLASSES + d, sh16 = (i & 1u) * 16u;
  return (PL_ATOM_ADD(&cls[i >> 1], 1u << sh16) >> sh16) & 0xFFFFu;
}
/* the counts of group g: cf[] finishing ops per class, ct[] all ops per class; returns the number of ops (nf: finishing ones) */
SB_HD uint32_t pl_cls_counts(const uint32_t *cls, uint32_t g, uint32_t *cf, uint32_t *ct, uint32_t *nf) {
  uint32_t n = 0, f = 0;
  const uint32_t *w = cls + (size_t)g * NRQ_LANE_CLASSES; /* (two counters per word: 8 words per group) */
#pragma unroll
  for (uint32_t d = 0; d < NRQ_LANE_CLASSES; d++) {
    const uint32_t a = (w[d >> 1] >> ((d & 1u) * 16u)) & 0xFFFFu;
    const uint32_t e = (w[(NRQ_LANE_CLASSES + d) >> 1] >> ((d & 1u) * 16u)) & 0xFFFFu;
    cf[d] = a; ct[d] = a + e; f += a; n += a + e;
  }
  *nf = f;
  return n;
}
/* without the class counters: [0] ops counted per group (= lev_ops), [2] ops placed (= lev_fill) */
SB_HD uint32_t *pl_lev_ctr(const PlanCtx &c, uint32_t which) { return which == 0 ? c.lev_ops : c.lev_fill; }
/* group of the op (dst row r on level t) <- (pivot column col): t itself for a finishing op */
SB_HD uint32_t pl_op_group(const uint16_t *collev, uint32_t t, uint32_t r, uint32_t col) {
  if (!collev) return t;
  const uint32_t e = (uint32_t)collev[col] + 1u;
  if (e >= t) return t;
  uint32_t h = r * 0x9E3779B1u ^ col * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return e + h % (t - e);
}
/* number of levels = deepest pivot + 1 (one reduction here instead of an atomic per claim) */
template <int Z> SB_HD void pl_lev_0(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  uint32_t m = 0;
  pl_for_batched(tid, nt, sh->npiv, [&](uint32_t k) { return (c.rowinfo[c.pivslot[k]] & PL_LEVEL_MASK) + 1u; },
                 [&](uint32_t, uint32_t lv) { if (lv > m) m = lv; });
  m = PL_WAVE_MAX(m);
  if (m && PL_WAVE_LEADER(tid)) PL_ATOM_MAX(&sh->nlev, m);
}
/* Behind a chained peel the pivots are numbered in the order the chain reached them, not level by level as the rounds number
 * them; everything after walks the pivots in that order, and with the levels mixed the op records of one wave scatter over the
 * whole stream (pl_w_init / pl_ops_emit: +0.35 M clocks at K=8192).  One counting sort by level puts the order back: the
 * thread keeps its pivots in registers across the barriers, so the lists are permuted in place.  Device only, as the chain is. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void pl_pivot_sort_dev(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  constexpr uint32_t PER = 16u;
  const uint32_t npiv = sh->npiv, nlev = sh->nlev; /* (levels 0 .. nlev-1) */
  if (sh->status || nlev > c.qcap || npiv > PER * nt) return; /* (only the order is at stake) */
  uint32_t *hist = reinterpret_cast<uint32_t *>(c.queue(0u)), *base = hist + c.qcap; /* (the queues and claim lists: 2 x qcap words) */
  PL_ASSUME_LDS(hist); PL_ASSUME_LDS(base);
  const uint32_t *rowinfo = c.rowinfo; PL_ASSUME_LDS(rowinfo);
  uint32_t *colinfo = c.colinfo; PL_ASSUME_LDS(colinfo);
  for (uint32_t l = tid; l < nlev; l += nt) hist[l] = 0u;
  __syncthreads();
  uint32_t slot[PER], col[PER], lv[PER], rk[PER];
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) {
    const uint32_t k = tid + j * nt;
    slot[j] = k < npiv ? c.pivslot[k] : 0u; col[j] = k < npiv ? c.pivcol[k] : 0u;
  }
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) lv[j] = rowinfo[slot[j]] & PL_LEVEL_MASK;
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) rk[j] = tid + j * nt < npiv ? PL_ATOM_ADD(&hist[lv[j]], 1u) : 0u;
  __syncthreads();
  for (uint32_t l = tid; l < nlev; l += nt) {
    uint32_t run = 0;
    for (uint32_t i = 0; i < l; i++) run += hist[i];
    base[l] = run;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < PER; j++) {
    if (tid + j * nt >= npiv) continue;
    const uint32_t k2 = base[lv[j]] + rk[j];
    c.pivslot[k2] = (uint16_t)slot[j]; c.pivcol[k2] = (uint16_t)col[j];
    colinfo[col[j]] = (PL_ST_PIVOT << 30) | k2;
  }
}
#endif
template <int Z> SB_HD void pl_pivot_sort(PlanCtx &c, uint32_t tid, uint32_t nt) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (PL_CHAIN && Z == 0) pl_pivot_sort_dev(c, tid, nt);
#else
  (void)c; (void)tid; (void)nt;
#endif
}
template <int Z> SB_HD void pl_lev_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid == 0) {
    const uint32_t u = c.p.L - sh->npiv;
    sh->wpr = u ? (u + 31u) / 32u : 1u;
    if (c.p.P + sh->ninact != u) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); /* cannot happen: every column is pivot or inactive */
  }
  for (uint32_t l = tid; l < sh->nlev + 2u; l += nt) { c.lev_ops[l] = 0; c.lev_fill[l] = 0; }
  for (uint32_t j = tid; j < c.lowcap; j += nt) c.gj_used()[j] = 0; /* (the bytes were the patch-column bits during peeling) */
}
/* Plan check behind the peel (the device-only peeling forms are race-based by design -- chained claims, batched
 * inactivation events, the pivot sort -- and a wrong plan decodes to silently wrong bytes).  Two cheap passes establish, for
 * every planner instance (device, big-block device, emulator), that the peel's books describe a permutation:
 *   a) pivot k's column says "pivot k" (so pivot columns are distinct), its row is marked assigned, and k is written to chk[row];
 *   b) chk[row of pivot k] still holds k (so pivot ROWS are distinct); every column is pivot or inactive, the pivot columns
 *      number npiv, every inactive column x has ucol[index] == x (distinct W bits) and they number L - npiv.
 * With the entry pass's two rules (pl_w_entries: a row op's source is final before the row's level; every pivot row holds
 * its own column, counted) the plan then IS a forward substitution of the block's matrix; what is violated raises
 * PL_FAIL_CAPACITY and the caller re-plans the block on the host (nrq_device.hip decode_device: capacity fallback).
 * Cost at K=8192: two passes over 8.4 k pivots / columns by 1024 threads, ~10 k of the planner's 3.7 M clocks. */
template <int Z> SB_HD void pl_check_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid == 0) { sh->own_hits = 0; sh->chk_piv = 0; sh->chk_inact = 0; } /* (a segmented run counts in wentry[3]: zeroed by pl_lev_b) */
  if (sh->status) return;
  struct PV { uint32_t slot, info, rinfo; };
  bool bad = false;
  pl_for_batched(tid, nt, sh->npiv,
                 [&](uint32_t k) { const uint32_t sl = c.pivslot[k]; return PV{sl, c.colinfo[c.pivcol[k]], sl < sh->M ? c.rowinfo[sl] : PL_UNASSIGNED}; },
                 [&](uint32_t k, PV v) {
                   if (v.info != ((PL_ST_PIVOT << 30) | k) || (v.rinfo & PL_UNASSIGNED) || v.slot >= sh->M) { bad = true; return; }
                   c.chk[v.slot] = (uint16_t)k;
                 });
  if (bad) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
}
template <int Z> SB_HD void pl_check_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  bool bad = false;
  pl_for_batched(tid, nt, sh->npiv, [&](uint32_t k) { return (uint32_t)c.chk[c.pivslot[k]]; },
                 [&](uint32_t k, uint32_t owner) { if (owner != (k & 0xFFFFu)) bad = true; });
  uint32_t np = 0, ni = 0;
  const uint32_t u = c.p.L - sh->npiv;
  for (uint32_t col = tid; col < c.p.L; col += nt) {
    const uint32_t info = c.colinfo[col], st = info >> 30, idx = info & 0x3FFFFFFFu;
    if (st == PL_ST_PIVOT) np++;
    else if (st == PL_ST_INACT) { ni++; if (idx >= u || c.ucol[idx] != (uint16_t)col) bad = true; }
    else bad = true; /* a column still in V (or claimed and never made a pivot) */
  }
  if (np) PL_ATOM_ADD(&sh->chk_piv, np);
  if (ni) PL_ATOM_ADD(&sh->chk_inact, ni);
  if (bad) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
}
template <int Z> SB_HD void pl_check_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  (void)nt;
  if (tid == 0 && !sh->status && (sh->chk_piv != sh->npiv || sh->chk_inact != c.p.L - sh->npiv)) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
}
template <int Z> SB_HD void pl_lev_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (uint32_t *l = pl_cls(c)) { /* nlev is final now */
    for (uint32_t k = tid; k < pl_lev_words(c) * (PL_CLS_BYTES / 4u); k += nt) l[k] = 0;
    if (c.cls_glob && tid < 4u) c.wentry[tid] = 0u;
    uint16_t *collev = pl_col_level(c);
    struct CL { uint32_t col, lev; };
    pl_for_batched(tid, nt, sh->npiv, [&](uint32_t k) { return CL{c.pivcol[k], c.rowinfo[c.pivslot[k]] & PL_LEVEL_MASK}; },
                   [&](uint32_t, CL v) { collev[v.col] = (uint16_t)v.lev; });
  }
  /* W rows are accumulated with XORs: start from zero */
  for (uint32_t e = tid; e < sh->M * sh->wpr; e += nt) c.wrows[e] = 0;
  for (uint32_t k = tid; k < sh->npiv; k += nt) c.pivdeg[k] = 0;
  for (uint32_t k = tid; k < sh->M; k += nt) c.lowdeg[k] = 0;
}
/* One row op found by pl_w_init (LDS-counter path): counted in its group -- the counter's old value is its place
 * among the group's finishing / early ops -- and recorded for pl_ops_emit, which then is a plain permutation pass
 * instead of a second walk over the rows (a walk is a chain of dependent trips to L2 per row). */
SB_HD void pl_record_op(PlanCtx &c, uint32_t *cls, const uint16_t *collev, uint32_t lev, uint32_t r,
                        uint32_t col, uint32_t info) {
  const uint32_t g = pl_op_group(collev, lev, r, col);
  const uint32_t early = g == lev ? 0u : 0x80000000u;
  const uint32_t src = c.pivslot[info & 0x3FFFFFFFu];
  const uint32_t word = NRQ_OP(r, src);
  const uint32_t idx = pl_cls_take(cls, g, early ? 1u : 0u, nrq_op_class(word));
  const uint32_t i = PL_ATOM_ADD(&c.sh->nrec, 1u);
  if (i >= c.reccap) { (c.sh->fail_site = __LINE__, c.sh->status = PL_FAIL_CAPACITY); return; }
  c.rec_word[i] = word;
  c.rec_idx[i] = early | idx;
  c.rec_g[i] = (uint16_t)g;
}
/* W starts as A restricted to the inactive columns, row by slot; op counts per row and per level group.
 * Ordinary rows: 8 lanes per row, sharing its entries.  The long LDPC rows (r < S): 64 lanes each (pl_w_init_b). */
/* one entry (row r, column col) of the constraint matrix: an inactive column toggles its bit of the row's W row,
 * a pivot column other than the row's own is a row op */
#ifndef PL_WU
#define PL_WU 4u
#endif
/* PL_WU: entries a thread has in flight: every step below is a trip to L2 or beyond (the planner's working
                   set, ~2 MB per block, does not stay in the 4 MB of L2 that 32 blocks share) */
SB_HD void pl_w_entries(PlanCtx &c, uint32_t *cls, const uint16_t *collev, const uint32_t (&r)[PL_WU],
                        const uint32_t (&col)[PL_WU], const bool (&use)[PL_WU], bool base) {
  uint32_t rinfo[PL_WU], info[PL_WU], src[PL_WU];
  bool on[PL_WU];
#pragma unroll
  for (uint32_t j = 0; j < PL_WU; j++) {
    rinfo[j] = use[j] ? c.rowinfo[r[j]] : 0u;
    info[j] = use[j] ? c.colinfo[col[j]] : 0u;
    on[j] = use[j] && !(base && (rinfo[j] & PL_PATCHED)); /* (not: a base entry of a row this block replaced) */
  }
#pragma unroll
  for (uint32_t j = 0; j < PL_WU; j++)
    src[j] = (on[j] && (info[j] >> 30) == PL_ST_PIVOT) ? c.pivslot[info[j] & 0x3FFFFFFFu] : PL_NONE;
#pragma unroll
  for (uint32_t j = 0; j < PL_WU; j++) {
    const uint32_t idx = info[j] & 0x3FFFFFFFu;
    const bool inact = on[j] && (info[j] >> 30) == PL_ST_INACT;
    const bool own = on[j] && !inact && src[j] == r[j]; /* the row's own pivot column */
    const bool op = on[j] && !inact && !own;
    if (inact) PL_ATOM_XOR(&c.wrows[(size_t)r[j] * c.sh->wpr + (idx >> 5)], 1u << (idx & 31u));
    (void)PL_WAVE_TAKE(c.own_ptr, own); /* (plan check: every pivot row must hold its pivot column -- counted, compared in pl_ops_layout_a) */
    const uint32_t i = PL_WAVE_TAKE(c.nrec_ptr, op); /* (by the whole wave: see there) */
    if (!op) continue;
    const uint32_t lev = (rinfo[j] & PL_UNASSIGNED) ? c.sh->nlev : (rinfo[j] & PL_LEVEL_MASK);
    /* plan check, the rule the forward passes rest on: the source of a row op -- the pivot row of another column of the row --
     * is final BEFORE the row's own level (peeling took this row when that column had left V).  A peel that claimed a row
     * too early (the race-based device forms: chained claims, batched events) shows here; the block goes to the host planner. */
    if (collev && (uint32_t)collev[col[j]] >= lev) { (c.sh->fail_site = __LINE__, c.sh->status = PL_FAIL_CAPACITY); continue; }
    const uint32_t g = pl_op_group(collev, lev, r[j], col[j]);
    const uint32_t early = g == lev ? 0u : 0x80000000u;
    const uint32_t word = NRQ_OP(r[j], src[j]);
    const uint32_t at = pl_cls_take(cls, g, early ? 1u : 0u, nrq_op_class(word));
    if (i >= c.reccap) { (c.sh->fail_site = __LINE__, c.sh->status = PL_FAIL_CAPACITY); continue; }
    c.rec_word[i] = word;
    c.rec_idx[i] = early | at;
    c.rec_g[i] = (uint16_t)g;
  }
}
/* `part` of `nparts`: the entries (collev path) are dealt out over nparts workgroups -- nrq_wentry_kernel; one workgroup: 0 of 1 */
template <int Z> SB_HD void pl_w_init_part(PlanCtx &c, uint32_t part, uint32_t nparts, uint32_t tid, uint32_t nt);
template <int Z> SB_HD void pl_w_init(PlanCtx &c, uint32_t tid, uint32_t nt) { pl_w_init_part<Z>(c, 0u, 1u, tid, nt); }
template <int Z> SB_HD void pl_w_init_part(PlanCtx &c, uint32_t part, uint32_t nparts, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t grp = tid >> 3, w8 = tid & 7u, ngrp = nt >> 3, wpr = sh->wpr, S = c.p.S;
  const uint32_t total = sh->npiv + sh->nlow;
  uint32_t *cntF = pl_lev_ctr(c, 0), *cls = pl_cls(c);
  const uint16_t *collev = pl_col_level(c);
  if (collev) {
    /* with the group counters in LDS every entry stands for itself: one pass over the entries of the base structure
     * (rows this block patched excepted) and of the patch rows -- coalesced, no walk from row to row */
    const uint32_t nnz = c.kh->nnz, nl = c.job.nlost, npq = sh->npatch * PL_PATCH_STRIDE;
    const uint32_t vt = part * nt + tid, vnt = nparts * nt; /* thread and thread count over all parts */
    for (uint32_t e0 = vt; e0 < nnz; e0 += PL_WU * vnt) {
      uint32_t r[PL_WU], col[PL_WU];
      bool use[PL_WU];
#pragma unroll
      for (uint32_t j = 0; j < PL_WU; j++) {
        const uint32_t e = e0 + j * vnt;
        use[j] = e < nnz;
        r[j] = use[j] ? c.b_erow[e] : 0u;
        col[j] = use[j] ? c.b_cidx[e] : 0u;
      }
      pl_w_entries(c, cls, collev, r, col, use, true);
    }
    for (uint32_t q0 = vt; q0 < npq; q0 += PL_WU * vnt) {
      uint32_t r[PL_WU], col[PL_WU];
      bool use[PL_WU];
#pragma unroll
      for (uint32_t j = 0; j < PL_WU; j++) {
        const uint32_t q = q0 + j * vnt, i = q / PL_PATCH_STRIDE, k = q - i * PL_PATCH_STRIDE;
        use[j] = q < npq && k < c.patch_len[i];
        r[j] = !use[j] ? 0u : i < nl ? c.p.S + c.p.H + c.lost[i] : c.p.L + (i - nl);
        col[j] = use[j] ? c.patch_cols[q] : 0u;
      }
      pl_w_entries(c, cls, collev, r, col, use, false);
    }
    return;
  }
  /* item i < npiv: pivot i; otherwise leftover row i - npiv (group nlev).  Each step of a row is a dependent
   * trip to L2/HBM (slot -> patch -> row pointers -> entries), so a lane group works on RB rows at once, stage
   * by stage, to have their loads in flight together. */
  constexpr uint32_t RB = 4;
  for (uint32_t i0 = grp; i0 < total; i0 += RB * ngrp) {
    uint32_t ii[RB], r[RB], own[RB], n[RB], e0[RB], e1[RB];
    const uint16_t *cols[RB];
    bool piv[RB], use[RB];
#pragma unroll
    for (uint32_t j = 0; j < RB; j++) {
      ii[j] = i0 + j * ngrp;
      use[j] = ii[j] < total;
      piv[j] = ii[j] < sh->npiv;
      r[j] = !use[j] ? 0u : piv[j] ? c.pivslot[ii[j]] : c.lowslot[ii[j] - sh->npiv];
      own[j] = (use[j] && piv[j]) ? c.pivcol[ii[j]] : PL_NONE;
    }
#pragma unroll
    for (uint32_t j = 0; j < RB; j++) {
      n[j] = 0; cols[j] = c.patch_cols;
      if (!use[j]) continue;
      if (r[j] < S) { /* a long LDPC row: listed for pl_w_init_b (the frontier queue is free by now; S <= 907 < PL_QCAP) */
        if (w8 == 0) c.queue(0u)[PL_ATOM_ADD(&sh->nq[0], 1u)] = (uint16_t)ii[j];
        use[j] = false;
        continue;
      }
      n[j] = pl_row(c, r[j], &cols[j]);
    }
#pragma unroll
    for (uint32_t j = 0; j < RB; j++) { /* the first two entries of every lane (covers rows of up to 16 entries) */
      e0[j] = w8 < n[j] ? cols[j][w8] : PL_NONE;
      e1[j] = w8 + 8u < n[j] ? cols[j][w8 + 8u] : PL_NONE;
    }
#pragma unroll
    for (uint32_t j = 0; j < RB; j++) {
      if (!use[j]) continue;
      /* the 8 lanes share the row's entries; inactive ones toggle their bit in the (zeroed) W row, the others count */
      uint32_t *dst = c.wrows + (size_t)r[j] * wpr;
      const uint32_t lev = piv[j] ? (c.rowinfo[r[j]] & PL_LEVEL_MASK) : sh->nlev;
      uint32_t deg = 0;
      for (uint32_t k = w8, t = 0; k < n[j]; k += 8u, t++) {
        const uint32_t col = t == 0 ? e0[j] : t == 1 ? e1[j] : (uint32_t)cols[j][k];
        const uint32_t info = c.colinfo[col], idx = info & 0x3FFFFFFFu;
        if ((info >> 30) == PL_ST_INACT) PL_ATOM_XOR(&dst[idx >> 5], 1u << (idx & 31u));
        else if (col != own[j]) {
          if (collev) pl_record_op(c, cls, collev, lev, r[j], col, info);
          else deg++;
        }
      }
      if (deg) { /* no LDS counters: the row's ops as one run of its level group */
        PL_ATOM_ADD(piv[j] ? &c.pivdeg[ii[j]] : &c.lowdeg[ii[j] - sh->npiv], deg);
        PL_ATOM_ADD(&cntF[lev], deg);
      }
    }
  }
}
/* the long rows (slots [0,S)) listed by pl_w_init: 64 lanes each; the destination words were zeroed in pl_lev_b */
template <int Z> SB_HD void pl_w_init_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpr = sh->wpr, lane = tid & 63u, nlong = sh->nq[0];
  uint32_t *cntF = pl_lev_ctr(c, 0), *cls = pl_cls(c);
  const uint16_t *collev = pl_col_level(c);
  for (uint32_t q = tid >> 6; q < nlong; q += nt >> 6) {
    const uint32_t i = c.queue(0u)[q];
    const bool piv = i < sh->npiv;
    const uint32_t r = piv ? c.pivslot[i] : c.lowslot[i - sh->npiv], own = piv ? c.pivcol[i] : PL_NONE;
    const uint32_t lev = piv ? (c.rowinfo[r] & PL_LEVEL_MASK) : sh->nlev;
    const uint16_t *cols;
    const uint32_t n = pl_row(c, r, &cols);
    uint32_t *dst = c.wrows + (size_t)r * wpr;
    uint32_t deg = 0;
    for (uint32_t k = lane; k < n; k += 64u) {
      const uint32_t col = cols[k];
      const uint32_t info = c.colinfo[col], idx = info & 0x3FFFFFFFu;
      if ((info >> 30) == PL_ST_INACT) PL_ATOM_XOR(&dst[idx >> 5], 1u << (idx & 31u));
      else if (col != own) {
        if (collev) pl_record_op(c, cls, collev, lev, r, col, info);
        else deg++;
      }
    }
    if (deg) {
      PL_ATOM_ADD(piv ? &c.pivdeg[i] : &c.lowdeg[i - sh->npiv], deg);
      PL_ATOM_ADD(&cntF[lev], deg);
    }
  }
}

#define PL_OPQ_WORDS 1024u /* op words a prefetch buffer holds (4 chunks) */
SB_HD uint32_t *pl_aux_lvbase(const PlanCtx &c) { return reinterpret_cast<uint32_t *>(c.aux_lds); }
SB_HD uint32_t *pl_aux_lvops(const PlanCtx &c) { return reinterpret_cast<uint32_t *>(c.aux_lds) + (c.sh->nlev + 2u); }
SB_HD uint32_t *pl_aux_opq(const PlanCtx &c, uint32_t which) {
  return reinterpret_cast<uint32_t *>(c.aux_lds + pl_r16((c.sh->nlev + 2u) * 8u)) + which * PL_OPQ_WORDS;
}
/* rows a level group of n ops occupies in the stream (plan.h); lev_base[] counts rows */
SB_HD uint32_t pl_group_rows(uint32_t n) { return (n + NRQ_ROW - 1u) / NRQ_ROW; }

/* stage the per-level op counts / chunk bases in LDS so that a level of the W pass starts without a trip
 * to HBM; prefetch the first group's ops */
template <int Z> SB_HD void pl_w_stage(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t need = pl_r16((sh->nlev + 2u) * 8u) + 2u * PL_OPQ_WORDS * 4u;
  const bool ok = need <= c.aux_bytes;
  if (tid == 0) { sh->lv_in_lds = ok ? 1u : 0u; sh->opq_group[0] = sh->opq_group[1] = PL_NONE; }
  if (!ok) return;
  for (uint32_t l = tid; l < sh->nlev + 2u; l += nt) { pl_aux_lvbase(c)[l] = c.lev_base[l]; pl_aux_lvops(c)[l] = c.lev_ops[l]; }
}

/* One level group of the op stream applied to W: W[dst] ^= W[src] for every op of the group -- the same
 * forward substitution the solve kernel performs on symbols, here on the bit rows (8 lanes per op; for
 * wpr > 8 each lane takes several words).  While a group is processed the op words of the next group are
 * fetched into the other LDS buffer. */
template <int Z> SB_HD void pl_w_group(PlanCtx &c, uint32_t group, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t grp = tid >> 3, w8 = tid & 7u, ngrp = nt >> 3, wpr = sh->wpr;
  const bool lds = sh->lv_in_lds != 0u;
  /* the group's rows [base, next): every place of them is looked at (padding ops are skipped), place e = (row base + e / 64, lane e % 64) */
  const uint32_t base = lds ? pl_aux_lvbase(c)[group] : c.lev_base[group];
  const uint32_t next = lds ? pl_aux_lvbase(c)[group + 1u] : c.lev_base[group + 1u];
  const uint32_t nops = (next - base) * NRQ_ROW;
  const uint32_t *gops = reinterpret_cast<const uint32_t *>(c.arena + sh->off_ops);
  const bool staged = lds && sh->opq_group[group & 1u] == group;
  const uint32_t *opq = pl_aux_opq(c, group & 1u);
  auto op_at = [&](uint32_t e) { return staged ? opq[e] : gops[NRQ_OP_INDEX(base + e / NRQ_ROW, e % NRQ_ROW)]; };
  /* issue the prefetch of the next group first: its loads overlap with this group's work */
  uint32_t pf[PL_OPQ_WORDS / PL_NT_TINY], pf_n = 0; /* (PL_OPQ_WORDS / nt of them are used) */
  if (lds && group + 1u <= sh->nlev) {
    const uint32_t b2 = pl_aux_lvbase(c)[group + 1u], n2 = (pl_aux_lvbase(c)[group + 2u] - b2) * NRQ_ROW;
    if (n2 <= PL_OPQ_WORDS) {
      pf_n = n2;
#pragma unroll
      for (uint32_t q = 0; q < PL_OPQ_WORDS / PL_NT_TINY; q++)
        if (q < PL_OPQ_WORDS / nt) {
          const uint32_t e = tid + q * nt;
          pf[q] = e < n2 ? gops[NRQ_OP_INDEX(b2 + e / NRQ_ROW, e % NRQ_ROW)] : 0u /* a padding op */;
        }
    }
  }
  if (wpr <= 8u) {
    /* one word per lane; 8 ops per lane group in flight: all op words first, then all source words */
    const uint32_t wsel = w8 < wpr ? w8 : 0u;
    for (uint32_t e0 = grp; e0 < nops; e0 += 8u * ngrp) {
      uint32_t op[8], v[8];
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) {
        const uint32_t e = e0 + q * ngrp;
        op[q] = e < nops ? op_at(e) : 0u;
      }
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) {
        const uint32_t srow = NRQ_OP_IS_NOP(op[q]) ? 0u : NRQ_OP_SRC(op[q]);
        v[q] = c.wrows[(size_t)srow * wpr + wsel];
      }
#pragma unroll
      for (uint32_t q = 0; q < 8; q++)
        if (!NRQ_OP_IS_NOP(op[q]) && w8 < wpr && v[q]) PL_ATOM_XOR(&c.wrows[(size_t)NRQ_OP_DST(op[q]) * wpr + w8], v[q]);
    }
  } else {
    /* up to 5 words per lane (wpr <= 40); 4 ops per lane group in flight: all op words, then all source words, then
     * the XORs -- one op at a time this is a chain of two dependent trips to L2 per op (K'=56403: 22 k clocks per level) */
    constexpr uint32_t OB = 4, WL = 5;
    for (uint32_t e0 = grp; e0 < nops; e0 += OB * ngrp) {
      uint32_t op[OB], v[OB][WL];
#pragma unroll
      for (uint32_t q = 0; q < OB; q++) {
        const uint32_t e = e0 + q * ngrp;
        op[q] = e < nops ? op_at(e) : 0u;
      }
#pragma unroll
      for (uint32_t q = 0; q < OB; q++) {
        const uint32_t *src = c.wrows + (size_t)(NRQ_OP_IS_NOP(op[q]) ? 0u : NRQ_OP_SRC(op[q])) * wpr;
#pragma unroll
        for (uint32_t j = 0; j < WL; j++) v[q][j] = w8 + 8u * j < wpr ? src[w8 + 8u * j] : 0u;
      }
#pragma unroll
      for (uint32_t q = 0; q < OB; q++) {
        if (NRQ_OP_IS_NOP(op[q])) continue;
        uint32_t *dst = c.wrows + (size_t)NRQ_OP_DST(op[q]) * wpr;
#pragma unroll
        for (uint32_t j = 0; j < WL; j++)
          if (v[q][j]) PL_ATOM_XOR(&dst[w8 + 8u * j], v[q][j]);
      }
    }
  }
  if (pf_n) {
    uint32_t *dstq = pl_aux_opq(c, (group + 1u) & 1u);
#pragma unroll
    for (uint32_t q = 0; q < PL_OPQ_WORDS / PL_NT_TINY; q++)
      if (q < PL_OPQ_WORDS / nt)
      if ((tid + q * nt) < pf_n) dstq[tid + q * nt] = pf[q];
  }
  if (tid == 0) sh->opq_group[(group + 1u) & 1u] = pf_n ? group + 1u : PL_NONE;
}

/* ---- W pass, fast path ----
 * W = X^-1 * A_U is the op stream applied to the bit rows -- the very computation the solve kernel performs on
 * symbol strips.  So the planner does it the same way: a strip of `wb` bytes (wb/4 words) of every W row is
 * brought into LDS as slot image (slot r at (r + NRQ_SCRATCH) * wb from LDS address 0, like the solve kernel),
 * ONE wave runs the row pipeline over the stream (fwd_rows; no barriers, LDS atomics), the strip goes back to
 * the HBM rows; ceil(wpr*4/wb) strips.  The image takes over the whole dynamic LDS region: the LDS-resident
 * peeling arrays that are still needed afterwards (rowinfo, colinfo) are spilled to their HBM homes first and
 * brought back at the end.  0 = does not fit even with 4-byte strips: the level-by-level pass on the HBM rows. */
SB_HD uint32_t pl_wfast_wb(const PlanCtx &c) {
  if (!c.lds_dyn) return 0u;
  for (uint32_t wb = 16u; wb >= 4u; wb >>= 1)
    if ((c.sh->M + NRQ_SCRATCH) * wb <= c.lds_dyn_bytes) return wb;
  return 0u;
}
SB_HD bool pl_peel_in_lds(const PlanCtx &c) { return reinterpret_cast<const uint8_t *>(c.rowinfo) == c.lds_dyn + pl_r16(c.Mcap * 4u); }
template <int Z> SB_HD void pl_wfast_spill(PlanCtx &c, uint32_t tid, uint32_t nt) {
  if (c.sh->status || !pl_peel_in_lds(c)) return;
  uint32_t *ri = reinterpret_cast<uint32_t *>(c.work + c.wl.rowinfo), *ci = reinterpret_cast<uint32_t *>(c.work + c.wl.colinfo);
  for (uint32_t k = tid; k < c.Mcap; k += nt) ri[k] = c.rowinfo[k];
  for (uint32_t k = tid; k < c.p.L; k += nt) ci[k] = c.colinfo[k];
}
template <int Z> SB_HD void pl_wfast_restore(PlanCtx &c, uint32_t tid, uint32_t nt) {
  if (c.sh->status || !pl_peel_in_lds(c)) return;
  const uint32_t *ri = reinterpret_cast<const uint32_t *>(c.work + c.wl.rowinfo), *ci = reinterpret_cast<const uint32_t *>(c.work + c.wl.colinfo);
  for (uint32_t k = tid; k < c.Mcap; k += nt) c.rowinfo[k] = ri[k];
  for (uint32_t k = tid; k < c.p.L; k += nt) c.colinfo[k] = ci[k];
}
template <int Z> SB_HD void pl_wfast_load(PlanCtx &c, uint32_t strip, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpl = pl_wfast_wb(c) / 4u, wpr = sh->wpr, n = (sh->M + NRQ_SCRATCH) * wpl;
  uint32_t *img = reinterpret_cast<uint32_t *>(c.lds_dyn);
  for (uint32_t e = tid; e < n; e += nt) {
    const uint32_t slot = e / wpl, k = e - slot * wpl, wd = strip * wpl + k;
    img[e] = (slot >= NRQ_SCRATCH && wd < wpr) ? c.wrows[(size_t)(slot - NRQ_SCRATCH) * wpr + wd] : 0u;
  }
}
template <int Z> SB_HD void pl_wfast_store(PlanCtx &c, uint32_t strip, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpl = pl_wfast_wb(c) / 4u, wpr = sh->wpr, n = sh->M * wpl;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(c.lds_dyn) + NRQ_SCRATCH * wpl;
  for (uint32_t e = tid; e < n; e += nt) {
    const uint32_t slot = e / wpl, k = e - slot * wpl, wd = strip * wpl + k;
    if (wd < wpr) c.wrows[(size_t)slot * wpr + wd] = img[e];
  }
}
/* the rows of the stream that exist at this point: the pivot levels and the leftover rows */
SB_HD uint32_t pl_wfast_rows(const PlanCtx &c) { return c.sh->spare_base; }

/* ---- HDPC fold through the transposed op stream ----
 * MhT[x] = G_U[x] ^ SUM_k W[k][x] * g_k  (g_k = the HDPC column of pivot k, 16 bytes).  With W = X^-1 * A_U that sum is
 * (g^T X^-1) A_U[:, x]: ONE pass of the op stream, transposed and in reverse, over a 16-byte value per slot (z <- g^T X^-1: rev_rows
 * of solve_body.h on the same LDS image the W pass uses), then MhT[x] ^= z_r for the rows r that have column x -- the column lists,
 * ~21 k entries at K=8192, most of them in the permanently inactive columns.  The fold over tiles of pivots (pl_mh_load /
 * pl_mh_acc: npiv x u masked 16-byte XORs, 0.59 M of the planner's 6.1 M clocks at K=8192) stays for the paths without an LDS
 * image (level-by-level W pass) and for the segmented runs of big blocks (nrq_mh_kernel).  The image holds wb bytes per slot, so
 * the 16 bytes go through in 16 / wb passes ("parts").  z is kept in the workspace (rec_word: dead since the op stream was
 * emitted), 16 bytes per slot, because the dense stage's LDS lies where the image was. */
#ifndef PL_MH_REV
#define PL_MH_REV 1
#endif
SB_HD uint8_t *pl_mhm(const PlanCtx &c); /* (below: MhT, 16 bytes per inactive column, at the start of the dense stage's LDS) */
SB_HD bool pl_mhrev_ok(const PlanCtx &c) { return PL_MH_REV && pl_wfast_wb(c) != 0u && c.sh->M * 4u <= c.reccap; }
template <int Z> SB_HD void pl_mhrev_load(PlanCtx &c, uint32_t part, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpl = pl_wfast_wb(c) / 4u, n = (sh->M + NRQ_SCRATCH) * wpl;
  uint32_t *img = reinterpret_cast<uint32_t *>(c.lds_dyn);
  for (uint32_t e = tid; e < n; e += nt) img[e] = 0u;
}
/* (after a barrier: the pivot rows' slots take their HDPC columns -- words [part * wpl, +wpl) of the 16 bytes) */
template <int Z> SB_HD void pl_mhrev_load_b(PlanCtx &c, uint32_t part, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpl = pl_wfast_wb(c) / 4u;
  uint32_t *img = reinterpret_cast<uint32_t *>(c.lds_dyn);
  struct SC { uint32_t slot, col; };
  pl_for_batched(tid, nt, sh->npiv, [&](uint32_t k) { return SC{c.pivslot[k], c.pivcol[k]}; },
                 [&](uint32_t, SC v) {
                   const uint32_t *g = reinterpret_cast<const uint32_t *>(c.GT + (size_t)v.col * 16u) + part * wpl;
                   for (uint32_t j = 0; j < wpl; j++) img[(size_t)(v.slot + NRQ_SCRATCH) * wpl + j] = g[j];
                 });
}
template <int Z> SB_HD void pl_mhrev_store(PlanCtx &c, uint32_t part, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpl = pl_wfast_wb(c) / 4u, n = sh->M * wpl;
  const uint32_t *img = reinterpret_cast<const uint32_t *>(c.lds_dyn) + NRQ_SCRATCH * wpl;
  uint32_t *z = c.rec_word; /* [M][4] words */
  for (uint32_t e = tid; e < n; e += nt) {
    const uint32_t slot = e / wpl, j = e - slot * wpl;
    z[(size_t)slot * 4u + part * wpl + j] = img[e];
  }
  if (tid == 0) sh->mhrev = 1u;
}
/* MhT[x] ^= z_r over the rows r of inactive column x (MhT = G_U from pl_mh_init); 32 lanes per column */
template <int Z> SB_HD void pl_mhrev_scatter(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t u = c.p.L - sh->npiv, lane = tid & 31u, grp = tid >> 5, ngrp = nt >> 5;
  const uint4 *z = reinterpret_cast<const uint4 *>(c.rec_word);
  uint32_t *MhT = reinterpret_cast<uint32_t *>(pl_mhm(c));
  for (uint32_t x = grp; x < u; x += ngrp) {
    const uint32_t col = c.ucol[x];
    const uint32_t a = c.b_cptr[col], nb = c.b_cptr[col + 1] - a, pa = c.pc_ptr[col], npc = c.pc_ptr[col + 1] - pa;
    uint4 acc; acc.x = acc.y = acc.z = acc.w = 0u;
    for (uint32_t e0 = lane; e0 < nb + npc; e0 += 4u * 32u) { /* four entries of the lane in flight: row, flags, z */
      uint32_t r[4];
      bool use[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t e = e0 + q * 32u;
        use[q] = e < nb + npc;
        r[q] = !use[q] ? 0u : e < nb ? (uint32_t)c.b_ridx[a + e] : (uint32_t)c.pc_rows[pa + (e - nb)];
      }
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t e = e0 + q * 32u;
        if (use[q] && e < nb && (c.rowinfo[r[q]] & PL_PATCHED)) use[q] = false; /* base entry of a row this block replaced */
      }
      uint4 v[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) v[q] = z[use[q] ? r[q] : 0u];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++)
        if (use[q]) { acc.x ^= v[q].x; acc.y ^= v[q].y; acc.z ^= v[q].z; acc.w ^= v[q].w; }
    }
    uint32_t *dst = MhT + (size_t)x * 4u;
    if (acc.x) PL_ATOM_XOR(&dst[0], acc.x);
    if (acc.y) PL_ATOM_XOR(&dst[1], acc.y);
    if (acc.z) PL_ATOM_XOR(&dst[2], acc.z);
    if (acc.w) PL_ATOM_XOR(&dst[3], acc.w);
  }
}

/* =============================== phase 3: leftover rows ====================================== */
template <int Z> SB_HD void pl_low_a(PlanCtx &c, uint32_t tid, uint32_t nt) { /* list them */
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const rq_params &p = c.p;
  for (uint32_t r = tid; r < sh->M; r += nt) {
    if (!(c.rowinfo[r] & PL_UNASSIGNED) || (r >= p.S && r < p.S + p.H)) continue;
    uint32_t j = PL_ATOM_ADD(&sh->nlow, 1u);
    if (j < c.ucap + 32u && j < c.lowcap) c.lowslot[j] = (uint16_t)r; else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
  }
}
template <int Z> SB_HD void pl_low_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid != 0) return;
  sh->nq[0] = 0; /* the frontier queue becomes pl_w_init's list of long rows */
  sh->lpr = (sh->nlow + PL_EXTRA_ROWS + 31u) / 32u; /* room for the rows a rank-deficient block may add */
  sh->rowlen = sh->wpr + sh->lpr;
  /* The dense stage's share of the dynamic LDS region from here on: MhT (16 bytes per inactive column) first, then
   * Mb (nlow x rowlen words).  The tiles of the HDPC fold (pl_mh_load / pl_mh_acc) lie over Mb's place: Mb is loaded
   * only after the fold (pl_low_c; planner_seq.h) -- with separate places K'=56403 at 20 % loss needed 141-147 KB of
   * the ~140 KB there are, and one block in eight went to the host planner for it. */
  if (sh->nlow + PL_EXTRA_ROWS > c.lowcap) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
  sh->tmp_mhoff = pl_r16(PL_MAXH * (c.p.L - sh->npiv)); /* byte offset of Mb (and of the tiles) behind MhT */
  const uint32_t mb_bytes = pl_r16((sh->nlow + PL_EXTRA_ROWS) * sh->rowlen * 4u), tile_bytes = PL_MH_TILE * 16u + PL_MH_TILE * sh->wpr * 4u;
  const uint32_t need = sh->tmp_mhoff + (mb_bytes > tile_bytes ? mb_bytes : tile_bytes);
  if (need > c.dense_bytes || sh->wpr > 40u) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
}
/* HDPC rows over the inactive columns, transposed: 16 bytes (one per HDPC row) per inactive column */
SB_HD uint8_t *pl_mhm(const PlanCtx &c) { uint8_t *q = c.dense_lds; PL_ASSUME_LDS(q); return q; }
SB_HD uint32_t *pl_mb(const PlanCtx &c) { uint32_t *q = reinterpret_cast<uint32_t *>(c.dense_lds + c.sh->tmp_mhoff); PL_ASSUME_LDS(q); return q; }
SB_HD uint8_t *pl_gtile(const PlanCtx &c) { uint8_t *q = c.dense_lds + c.sh->tmp_mhoff; PL_ASSUME_LDS(q); return q; }
SB_HD uint32_t *pl_wtile(const PlanCtx &c) { return reinterpret_cast<uint32_t *>(pl_gtile(c) + PL_MH_TILE * 16u); }

/* reduced coefficient rows of the leftover rows over the inactive columns (their W rows after the op
 * stream has run), with the augmented identity, into LDS */
template <int Z> SB_HD void pl_low_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpr = sh->wpr, rowlen = sh->rowlen;
  uint32_t *Mb = pl_mb(c);
  for (uint32_t e = tid; e < sh->nlow * rowlen; e += nt) {
    const uint32_t j = e / rowlen, wd = e - j * rowlen;
    Mb[e] = wd < wpr ? c.wrows[(size_t)c.lowslot[j] * wpr + wd] : ((wd - wpr) == (j >> 5) ? (1u << (j & 31u)) : 0u);
  }
}

/* =============================== phase 4: op stream layout =================================== */
/* sum of the first g 16-bit counts staged in the (free) frontier queues: g <= 2 * PL_QCAP */
SB_HD uint32_t pl_deg_prefix(const PlanCtx &c, uint32_t g) {
  const uint16_t *degq = c.queue(0u);
  uint32_t run = 0, i = 0;
  for (; i + 2u <= g; i += 2u) { /* two 16-bit lengths per word (the queues are 4-byte aligned) */
    const uint32_t w = *reinterpret_cast<const uint32_t *>(degq + i);
    run += (w & 0xFFFFu) + (w >> 16);
  }
  if (i < g) run += degq[i];
  return run;
}
/* rows of the stream that group l takes: the finishing ops first; the last NRQ_PIPE-1 rows hold early ops only (or
 * padding) -- also when the group is empty: early ops of the group before may complete rows that the group after reads */
SB_HD uint32_t pl_group_span(uint32_t l, uint32_t nf, uint32_t n) {
  if (!l) return 0u;
  const uint32_t need = pl_group_rows(nf) + (NRQ_PIPE - 1u), have = pl_group_rows(n);
  return have > need ? have : need;
}
/* rows of group l: with the class counters, what the lane placement needs (plan.h); else pl_group_span */
SB_HD uint32_t pl_group_span_of(const PlanCtx &c, uint32_t l, uint32_t *n_out, uint32_t *nf_out) {
  uint32_t nf, n;
  uint32_t span;
  if (const uint32_t *cls = pl_cls(c)) {
    uint32_t cf[NRQ_LANE_CLASSES], ct[NRQ_LANE_CLASSES];
    n = pl_cls_counts(cls, l, cf, ct, &nf);
    span = l ? nrq_group_span(n, cf) : 0u;
    for (uint32_t d = 0; d < NRQ_LANE_CLASSES; d++) /* (a 16-bit counter that came near its end: do not trust the block) */
      if (ct[d] >= 0x7FFFu) (c.sh->fail_site = __LINE__, c.sh->status = PL_FAIL_CAPACITY);
  } else {
    nf = n = pl_lev_ctr(c, 0)[l];
    span = pl_group_span(l, nf, n);
  }
  *n_out = n; *nf_out = nf;
  return span;
}
/* per group: op counts to the workspace, rows to LDS (the frontier queues are free by now) for the prefix sums of
 * pl_ops_layout -- when the groups are too many for that, pl_ops_layout walks them alone */
template <int Z> SB_HD void pl_ops_layout_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const bool in_lds = sh->nlev + 1u <= 2u * c.qcap;
  uint16_t *rowq = c.queue(0u);
  /* plan check (pl_check_a): the entry pass met the own pivot column of every pivot row exactly once */
  if (tid == 0 && !sh->status && pl_col_level(c) && sh->own_hits != sh->npiv) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
  for (uint32_t l = tid; l <= sh->nlev; l += nt) {
    uint32_t n, nf;
    const uint32_t span = pl_group_span_of(c, l, &n, &nf);
    c.lev_ops[l] = n;
    c.lev_fin[l] = nf;
    if (in_lds) {
      if (span > 0xFFFFu) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
      rowq[l] = (uint16_t)span;
    }
  }
}
template <int Z> SB_HD void pl_ops_layout(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  /* row base of every group (1..nlev-1: pivot levels, nlev: leftover rows); lev_base[nlev + 1] = the first row behind them */
  const bool in_lds = sh->nlev + 1u <= 2u * c.qcap;
  if (in_lds)
    for (uint32_t l = tid; l <= sh->nlev + 1u; l += nt) c.lev_base[l] = pl_deg_prefix(c, l);
  if (tid != 0) return;
  uint32_t rows = 0;
  if (in_lds) rows += pl_deg_prefix(c, sh->nlev + 1u);
  else {
    for (uint32_t l = 0; l <= sh->nlev; l++) {
      uint32_t n, nf;
      c.lev_base[l] = rows;
      rows += pl_group_span_of(c, l, &n, &nf); /* (lev_ops / lev_fin were written by pl_ops_layout_a) */
    }
    c.lev_base[sh->nlev + 1u] = rows;
  }
  sh->spare_base = rows; /* rows reserved for constraint rows added later */
  rows += PL_SPARE_ROWS + (NRQ_PIPE - 1u);
  sh->tmp0 = rows; /* rows of the stream */
  /* (in-stream form: generous bound for the combination group, nlow ones per reduced row at most) */
  const uint32_t bin_bound = pl_bin_in_stream(c.p.L) ? ((sh->nlow + PL_EXTRA_ROWS) * (sh->nlow + PL_EXTRA_ROWS) + NRQ_ROW - 1u) / NRQ_ROW + 1u : 0u;
  uint32_t total_rows = NRQ_STREAM_ROWS(rows + bin_bound + NRQ_PAD_ROWS);
  sh->off_ops = pl_r16(c.fixed_end);
  sh->arena_top = pl_r16(sh->off_ops + total_rows * NRQ_ROW * 4u);
  sh->opbase = total_rows;
  if (sh->arena_top > c.job.arena_cap) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
}
template <int Z> SB_HD void pl_ops_clear(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  uint32_t *ops = reinterpret_cast<uint32_t *>(c.arena + sh->off_ops);
  const uint32_t nw = sh->opbase * NRQ_ROW;
  for (uint32_t k = tid; k < nw; k += nt) ops[k] = NRQ_NOP_AT(NRQ_OP_LANE_OF_INDEX(k)); /* (the stream is quad-interleaved) */
}
/* the op word at place `pos` (row-major: 64 lanes per row) counted from stream row `row0` */
SB_HD uint32_t *pl_op_at(uint32_t *ops, uint32_t row0, uint32_t pos) { return ops + NRQ_OP_INDEX(row0 + pos / NRQ_ROW, pos % NRQ_ROW); }
/* position of the i-th op of a run starting at `pos` inside a group of n ops: a multiplicative shuffle
 * keeps the ops of one row apart so that the lanes of a wave rarely hit the same target slot */
SB_HD uint32_t pl_spread(uint32_t pos, uint32_t n) {
  static const uint32_t mult[6] = {61u, 67u, 71u, 73u, 79u, 83u};
  if (n < 128u) return pos;
  uint32_t m = 1u;
#pragma unroll
  for (int q = 5; q >= 0; q--)
    if (n % mult[q]) m = mult[q];
  return (uint32_t)(((uint64_t)pos * m) % n);
}
/* the ops of constraint row r (dst; on level `lev`), each into its group (pl_op_group): finishing ops fill the
 * group from the front, early ops follow them; the place inside either part is whatever the counter hands out,
 * which also keeps the ops of one row apart */
SB_HD void pl_emit_row(PlanCtx &c, uint32_t r, uint32_t own, uint32_t lev, uint32_t deg) {
  uint32_t *ops = reinterpret_cast<uint32_t *>(c.arena + c.sh->off_ops);
  uint32_t *fillF = pl_lev_ctr(c, 2);
  const uint16_t *cols;
  const uint32_t m = pl_row(c, r, &cols);
  /* (without the class counters only:) one run of `deg` places in the row's own group, shuffled (pl_spread) */
  uint32_t run = PL_ATOM_ADD(&fillF[lev], deg);
  for (uint32_t k = 0; k < m; k++) {
    const uint32_t col = cols[k], info = c.colinfo[col];
    if ((info >> 30) != PL_ST_PIVOT || col == own) continue;
    *pl_op_at(ops, c.lev_base[lev], pl_spread(run++, c.lev_ops[lev])) = NRQ_OP(r, c.pivslot[info & 0x3FFFFFFFu]);
  }
}
/* The permutation pass through LDS.  As stores, the ~46 k op words of a block of K=8192 are 4-byte writes scattered over a
 * quarter megabyte -- every one a partial line the memory has to read, merge and write -- and with all 256 planner workgroups in
 * this phase together (the chained peel keeps them in step) the pass took 0.72 M clocks against 0.13 M for the same pass
 * without its stores.  Here every record's place is computed once, as before, but the op goes to the list of the TILE of the
 * stream its place lies in (a tile: the quads that fit the aux region -- the rowstate image, dead since peeling, the column
 * levels in it since the entry pass; the lists: appended to at a counter in LDS, so their tails are lines being filled, not
 * lines being patched; a tile cannot receive more ops than it has places, so a list of that many entries never overflows); then
 * tile after tile is built in LDS -- padding everywhere, the list's ops at their places -- and written out as whole lines.
 * Device only (barriers inside the phase); the emulator and blocks whose class counters or tile would not be in LDS keep the
 * scattered form: same places, same stream. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool pl_ops_emit_tiled(PlanCtx &c, const uint32_t *cls, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (c.cls_glob || !c.aux_lds) return false;
  /* the tile: the aux region -- all of it when the class counters have a place of their own (the dense stage's region), what
   * lies in front of them when they sit at its end (blocks whose dense stage takes over the rowstate image: pl_cls_place) */
  const uint8_t *clsp = pl_cls_place(c);
  uint32_t tile_bytes;
  if (c.dense_lds != c.aux_lds && clsp == c.dense_lds) tile_bytes = c.aux_bytes;
  else if (clsp > c.aux_lds && clsp <= c.aux_lds + c.aux_bytes) tile_bytes = (uint32_t)(clsp - c.aux_lds);
  else return false;
  uint32_t tq = tile_bytes / 1024u; /* quads per tile: 4 rows x 64 lanes x 4 bytes each */
  if (tq > PL_EMIT_TILE_WORDS / 256u) tq = PL_EMIT_TILE_WORDS / 256u;
  const uint32_t nlev = sh->nlev, rows_emit = c.lev_base[nlev + 1u], nquad = (rows_emit + 3u) / 4u;
  if (tq < 16u || nquad == 0u || (nquad + tq - 1u) / tq > PL_EMIT_TILES_MAX) return false; /* (small blocks -- a tile of a few rows, a dozen
                                                                                               * tiles of three barriers each: K=2000 1.90 -> 1.95 ms per 1024 blocks -- keep the scattered form) */
  const uint32_t ntile = (nquad + tq - 1u) / tq, tile_words = tq * 4u * NRQ_ROW;
  if ((uint64_t)ntile * tile_words > 3ull * c.reccap) return false; /* (the lists' room in the workspace: pl_work_plan) */
  uint32_t *tile = reinterpret_cast<uint32_t *>(c.aux_lds); PL_ASSUME_LDS(tile);
  uint32_t *ops = reinterpret_cast<uint32_t *>(c.arena + sh->off_ops);
  uint2 *bkt = reinterpret_cast<uint2 *>(c.emit_bkt);
  const uint32_t nrec = sh->nrec;
  if (tid < PL_EMIT_TILES_MAX) sh->bkt_n[tid] = 0u;
  __syncthreads();
  for (uint32_t i0 = tid; i0 < nrec; i0 += PL_WU * nt) { /* PL_WU records in flight per thread */
    uint32_t m[PL_WU], g[PL_WU], w[PL_WU], base[PL_WU], next[PL_WU];
#pragma unroll
    for (uint32_t j = 0; j < PL_WU; j++) {
      const uint32_t i = i0 + j * nt < nrec ? i0 + j * nt : i0;
      m[j] = c.rec_idx[i]; g[j] = c.rec_g[i]; w[j] = c.rec_word[i];
    }
#pragma unroll
    for (uint32_t j = 0; j < PL_WU; j++) { base[j] = c.lev_base[g[j]]; next[j] = c.lev_base[g[j] + 1u]; }
#pragma unroll
    for (uint32_t j = 0; j < PL_WU; j++) {
      if (i0 + j * nt >= nrec) continue;
      uint32_t cf[NRQ_LANE_CLASSES], ct[NRQ_LANE_CLASSES], nf;
      pl_cls_counts(cls, g[j], cf, ct, &nf);
      const uint32_t d = nrq_op_class(w[j]);
      const uint32_t rank = (m[j] >> 31) ? cf[d] + (m[j] & 0x7FFFFFFFu) : m[j]; /* early ops rank behind the finishing ones */
      const uint32_t pos = nrq_lane_place(next[j] - base[j], ct, d, rank);
      const uint32_t at = (uint32_t)NRQ_OP_INDEX(base[j] + pos / NRQ_ROW, pos % NRQ_ROW), t = at / tile_words; /* word index in the stream, its tile */
      if (t >= ntile) { (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); continue; } /* (cannot happen: a place behind the groups' rows) */
      const uint32_t k = PL_ATOM_ADD(&sh->bkt_n[t], 1u);
      if (k < tile_words) bkt[(size_t)t * tile_words + k] = make_uint2(at - t * tile_words, w[j]);
      else (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); /* (cannot happen: more ops than places) */
    }
  }
  __syncthreads(); /* (the lists: global stores of this workgroup read back by it -- the barrier's release / acquire covers them) */
  for (uint32_t t = 0; t < ntile; t++) {
    const uint32_t q0 = t * tq, q1 = q0 + tq < nquad ? q0 + tq : nquad, nw = (q1 - q0) * 4u * NRQ_ROW;
    for (uint32_t k = tid; k < nw; k += nt) tile[k] = NRQ_NOP_AT(NRQ_OP_LANE_OF_INDEX(k));
    __syncthreads();
    const uint32_t n = sh->bkt_n[t] < tile_words ? sh->bkt_n[t] : tile_words;
    const uint2 *lst = bkt + (size_t)t * tile_words;
    pl_for_batched(tid, nt, n, [&](uint32_t k) { return lst[k]; }, [&](uint32_t, uint2 e) { if (e.x < nw) tile[e.x] = e.y; });
    __syncthreads();
    { /* the tile as whole lines */
      const uint4 *src = reinterpret_cast<const uint4 *>(tile);
      uint4 *dst = reinterpret_cast<uint4 *>(ops + (size_t)q0 * 4u * NRQ_ROW);
      for (uint32_t k = tid; k < nw / 4u; k += nt) dst[k] = src[k];
    }
    __syncthreads();
  }
  return true;
}
#endif
template <int Z> SB_HD void pl_ops_emit(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  if (const uint32_t *cls = pl_cls(c)) { /* the ops were recorded with group, class and rank: their lanes follow (plan.h "lane placement") */
#if defined(__HIP_DEVICE_COMPILE__)
    if (pl_ops_emit_tiled(c, cls, tid, nt)) return;
#endif
    uint32_t *ops = reinterpret_cast<uint32_t *>(c.arena + sh->off_ops);
    const uint32_t nrec = sh->nrec;
    for (uint32_t i0 = tid; i0 < nrec; i0 += PL_WU * nt) { /* PL_WU records in flight per thread */
      uint32_t m[PL_WU], g[PL_WU], w[PL_WU], base[PL_WU], next[PL_WU];
#pragma unroll
      for (uint32_t j = 0; j < PL_WU; j++) {
        const uint32_t i = i0 + j * nt < nrec ? i0 + j * nt : i0;
        m[j] = c.rec_idx[i]; g[j] = c.rec_g[i]; w[j] = c.rec_word[i];
      }
#pragma unroll
      for (uint32_t j = 0; j < PL_WU; j++) { base[j] = c.lev_base[g[j]]; next[j] = c.lev_base[g[j] + 1u]; }
#pragma unroll
      for (uint32_t j = 0; j < PL_WU; j++) {
        if (i0 + j * nt >= nrec) continue;
        uint32_t cf[NRQ_LANE_CLASSES], ct[NRQ_LANE_CLASSES], nf;
        pl_cls_counts(cls, g[j], cf, ct, &nf);
        const uint32_t d = nrq_op_class(w[j]);
        const uint32_t rank = (m[j] >> 31) ? cf[d] + (m[j] & 0x7FFFFFFFu) : m[j]; /* early ops rank behind the finishing ones */
        *pl_op_at(ops, base[j], nrq_lane_place(next[j] - base[j], ct, d, rank)) = w[j];
      }
    }
    return;
  }
  for (uint32_t k = tid; k < sh->npiv; k += nt) {
    const uint32_t r = c.pivslot[k], l = c.rowinfo[r] & PL_LEVEL_MASK;
    if (l) pl_emit_row(c, r, c.pivcol[k], l, c.pivdeg[k]); /* level 0 rows have nothing to gather */
  }
  for (uint32_t j = tid; j < sh->nlow; j += nt) pl_emit_row(c, c.lowslot[j], PL_NONE, sh->nlev, c.lowdeg[j]);
}

#ifdef NRQ_PLAN_SELFCHECK
/* debugging aid (-DNRQ_PLAN_SELFCHECK): every recorded op must have reached the stream (two ops on one place would lose
 * one); a mismatch is reported as capacity failure at "line" 70000 + the number of ops missing */
template <int Z> SB_HD void pl_ops_check_a(PlanCtx &c, uint32_t tid, uint32_t nt) { if (tid == 0) c.sh->tmp1 = 0; }
template <int Z> SB_HD void pl_ops_check_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status || !pl_cls(c)) return;
  const uint32_t *ops = reinterpret_cast<const uint32_t *>(c.arena + sh->off_ops);
  uint32_t n = 0;
  for (uint32_t k = tid; k < NRQ_STREAM_ROWS(sh->spare_base) * NRQ_ROW; k += nt) n += NRQ_OP_IS_NOP(ops[k]) ? 0u : 1u; /* (quad-interleaved: whole quads) */
  if (n) PL_ATOM_ADD(&sh->tmp1, n);
}
template <int Z> SB_HD void pl_ops_check_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid == 0 && !sh->status && pl_cls(c) && sh->tmp1 != sh->nrec) { sh->fail_site = 70000u + (sh->nrec - sh->tmp1); sh->status = PL_FAIL_CAPACITY; }
}
#endif
/* =============================== phase 5: HDPC rows over the inactive columns ================= */
/* MhT[x] = G_U[:,x] ^ SUM_k W[k][x] * G[:, pivcol k]  (16 bytes per inactive column x, in LDS).  The pivots
 * are streamed through LDS in tiles: their HDPC columns (16 bytes each, kconst GT) and their W rows. */
template <int Z> SB_HD void pl_mh_init(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const rq_params &p = c.p;
  const uint32_t u = p.L - sh->npiv, n_hd = p.Kp + p.S;
  uint4 *MhT = reinterpret_cast<uint4 *>(pl_mhm(c));
  for (uint32_t x = tid; x < u; x += nt) {
    const uint32_t col = c.ucol[x];
    uint4 v;
    if (col < n_hd) v = *reinterpret_cast<const uint4 *>(c.GT + (size_t)col * 16u);
    else {
      const uint32_t h = col - n_hd; /* identity part of the HDPC rows */
      v.x = h < 4 ? 1u << (8u * h) : 0u; v.y = (h >= 4 && h < 8) ? 1u << (8u * (h - 4)) : 0u;
      v.z = (h >= 8 && h < 12) ? 1u << (8u * (h - 8)) : 0u; v.w = (h >= 12 && h < 16) ? 1u << (8u * (h - 12)) : 0u;
    }
    MhT[x] = v;
  }
}
template <int Z> SB_HD void pl_mh_load(PlanCtx &c, uint32_t tile, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t k0 = tile * PL_MH_TILE, wpr = sh->wpr;
  const uint32_t cnt = (sh->npiv - k0) < PL_MH_TILE ? (sh->npiv - k0) : PL_MH_TILE;
  uint4 *Gt = reinterpret_cast<uint4 *>(pl_gtile(c));
  uint32_t *Wt = pl_wtile(c);
  for (uint32_t i = tid; i < cnt; i += nt) Gt[i] = *reinterpret_cast<const uint4 *>(c.GT + (size_t)c.pivcol[k0 + i] * 16u);
  for (uint32_t e = tid; e < cnt * wpr; e += nt) {
    const uint32_t i = e / wpr, wd = e - i * wpr;
    Wt[e] = c.wrows[(size_t)c.pivslot[k0 + i] * wpr + wd];
  }
}
template <int Z> SB_HD void pl_mh_acc(PlanCtx &c, uint32_t tile, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t k0 = tile * PL_MH_TILE, wpr = sh->wpr, u = c.p.L - sh->npiv;
  const uint32_t cnt = (sh->npiv - k0) < PL_MH_TILE ? (sh->npiv - k0) : PL_MH_TILE;
  uint4 *MhT = reinterpret_cast<uint4 *>(pl_mhm(c));
  const uint4 *Gt = reinterpret_cast<const uint4 *>(pl_gtile(c));
  const uint32_t *Wt = pl_wtile(c);
  /* u columns are far fewer than threads: `parts` threads share a column, each folds a slice of the tile's pivots in */
  const uint32_t upad = (u + 63u) & ~63u, parts = upad && nt / upad ? nt / upad : 1u;
  for (uint32_t t = tid; t < upad * parts; t += nt) {
    const uint32_t x = t % upad, part = t / upad;
    if (x >= u) continue;
    const uint32_t i0 = cnt * part / parts, i1 = cnt * (part + 1u) / parts;
    uint4 acc; acc.x = acc.y = acc.z = acc.w = 0u;
    const uint32_t wd = x >> 5, bt = x & 31u;
    for (uint32_t i = i0; i < i1; i++) {
      const uint32_t m = 0u - ((Wt[i * wpr + wd] >> bt) & 1u);
      const uint4 g = Gt[i];
      acc.x ^= g.x & m; acc.y ^= g.y & m; acc.z ^= g.z & m; acc.w ^= g.w & m;
    }
    uint32_t *dst = reinterpret_cast<uint32_t *>(&MhT[x]);
    if (acc.x) PL_ATOM_XOR(&dst[0], acc.x);
    if (acc.y) PL_ATOM_XOR(&dst[1], acc.y);
    if (acc.z) PL_ATOM_XOR(&dst[2], acc.z);
    if (acc.w) PL_ATOM_XOR(&dst[3], acc.w);
  }
}

/* ---- segmented runs (planner_seq.h): pl_shared through the workspace, MhT from nrq_mh_kernel ---- */
template <int Z> SB_HD void pl_sh_save(PlanCtx &c, uint32_t tid, uint32_t nt) {
  uint32_t *dst = reinterpret_cast<uint32_t *>(c.work + c.wl.sh_save);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(c.sh);
  for (uint32_t k = tid; k < c.sh_bytes / 4u; k += nt) dst[k] = src[k];
}
/* after nrq_wentry_kernel: its counters into this workgroup's LDS, its record count and verdict into pl_shared */
template <int Z> SB_HD void pl_cls_fetch(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t *g = reinterpret_cast<const uint32_t *>(c.work + c.wl.cls_g);
  if (uint32_t *l = pl_cls(c)) /* (cls_glob is off in this part: the LDS place) */
    for (uint32_t k = tid; k < pl_lev_words(c) * (PL_CLS_BYTES / 4u); k += nt) l[k] = g[k];
  if (tid == 0) {
    sh->nrec = c.wentry[0];
    sh->own_hits = c.wentry[3];
    if (c.wentry[1] && !sh->status) { sh->status = c.wentry[1]; sh->fail_site = c.wentry[2]; }
  }
}
/* a helper workgroup's capacity failure, for the planner workgroup to find */
template <int Z> SB_HD void pl_wentry_report(PlanCtx &c, uint32_t tid, uint32_t nt) {
  (void)nt;
  if (tid == 0 && c.sh->status) { c.wentry[2] = c.sh->fail_site; c.wentry[1] = c.sh->status; }
}
template <int Z> SB_HD void pl_sh_restore(PlanCtx &c, uint32_t tid, uint32_t nt) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(c.work + c.wl.sh_save);
  uint32_t *dst = reinterpret_cast<uint32_t *>(c.sh);
  for (uint32_t k = tid; k < c.sh_bytes / 4u; k += nt) dst[k] = src[k];
}
template <int Z> SB_HD void pl_mh_fetch(PlanCtx &c, uint32_t tid, uint32_t nt) {
  if (c.sh->status) return;
  const uint32_t u = c.p.L - c.sh->npiv;
  const uint4 *src = reinterpret_cast<const uint4 *>(c.work + c.wl.mh_ext);
  uint4 *MhT = reinterpret_cast<uint4 *>(pl_mhm(c));
  for (uint32_t x = tid; x < u; x += nt) MhT[x] = src[x];
}
/* nrq_mh_kernel: workgroup `part` of `nparts` folds its share of the pivot tiles into a private MhT (LDS) and XORs it
 * into the workspace copy (which the first part of the run zeroed: pl_mh_ext_clear).  Part 0 also contributes G_U. */
template <int Z> SB_HD void pl_mh_ext_clear(PlanCtx &c, uint32_t tid, uint32_t nt) {
  uint32_t *dst = reinterpret_cast<uint32_t *>(c.work + c.wl.mh_ext);
  for (uint32_t k = tid; k < c.ucap * PL_MAXH / 4u; k += nt) dst[k] = 0u;
}
template <int Z> SB_HD void pl_mh_part_zero(PlanCtx &c, uint32_t tid, uint32_t nt) {
  const uint32_t u = c.p.L - c.sh->npiv;
  uint4 *MhT = reinterpret_cast<uint4 *>(pl_mhm(c));
  uint4 z; z.x = z.y = z.z = z.w = 0u;
  for (uint32_t x = tid; x < u; x += nt) MhT[x] = z;
}
template <int Z> SB_HD void pl_mh_part_flush(PlanCtx &c, uint32_t tid, uint32_t nt) {
  const uint32_t u = c.p.L - c.sh->npiv;
  const uint32_t *MhT = reinterpret_cast<const uint32_t *>(pl_mhm(c));
  uint32_t *dst = reinterpret_cast<uint32_t *>(c.work + c.wl.mh_ext);
  for (uint32_t k = tid; k < u * 4u; k += nt)
    if (MhT[k]) PL_ATOM_XOR(&dst[k], MhT[k]);
}

/* =============================== phase 6: GF(2) Gauss-Jordan in LDS =========================== */
/* Column x: the unused row with the lowest index that has the column becomes its pivot row (atomic-min bids in
 * cand[x % 3]); the column is eliminated from every other row that has it.  gj_flag[j] bit (x & 1) = row j has
 * column x.  One barrier per column: while step B of column x rewrites the rows, the thread that owns the word with
 * bit x+1 of a row publishes that bit and bids for column x+1 (step A is only run for a column that has no
 * predecessor step).  Slot (x + 2) % 3 is cleared for the bids of the step after. */
template <int Z> SB_HD void pl_gj_a(PlanCtx &c, uint32_t x, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t *Mb = pl_mb(c);
  const uint32_t rowlen = sh->rowlen;
  for (uint32_t j = tid; j < sh->nlow; j += nt) {
    const uint32_t f = (Mb[(size_t)j * rowlen + (x >> 5)] >> (x & 31u)) & 1u;
    c.gj_flag()[j] = (uint8_t)(f << (x & 1u));
    if (f && !c.gj_used()[j]) PL_ATOM_MIN(&sh->cand[x % 3u], j);
  }
}
/* step B: eliminate the column from every other row that has it (or record a free column);
 * bit 31 of the argument: prepare column x + 1 */
template <int Z> SB_HD void pl_gj_b(PlanCtx &c, uint32_t xarg, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t x = xarg & 0x7FFFFFFFu, prep = xarg >> 31, xn = x + 1u;
  const uint32_t pr = sh->cand[x % 3u];
  uint32_t *Mb = pl_mb(c);
  const uint32_t rowlen = sh->rowlen, total = sh->nlow * rowlen;
  const uint32_t inv = 0xFFFFFFFFu / rowlen + 1u; /* e / rowlen == mulhi(e, inv) for e < 2^16 * rowlen */
  const uint32_t *src = Mb + (size_t)(pr == PL_NONE ? 0u : pr) * rowlen;
  const uint32_t cur = x & 1u, nxt = xn & 1u, wn = xn >> 5;
  if (pr != PL_NONE || prep) {
    for (uint32_t e = tid; e < total; e += nt) {
      const uint32_t j = (uint32_t)(((uint64_t)e * inv) >> 32), wd = e - j * rowlen;
      const uint32_t fl = c.gj_flag()[j];
      uint32_t v = Mb[e];
      if (pr != PL_NONE && j != pr && ((fl >> cur) & 1u)) { v ^= src[wd]; Mb[e] = v; }
      if (prep && wd == wn) {
        const uint32_t f = (v >> (xn & 31u)) & 1u;
        c.gj_flag()[j] = (uint8_t)((fl & (1u << cur)) | (f << nxt)); /* (the only writer of this byte in this phase) */
        if (f && j != pr && !c.gj_used()[j]) PL_ATOM_MIN(&sh->cand[xn % 3u], j);
      }
    }
  }
  if (tid == 0) {
    sh->cand[(x + 2u) % 3u] = PL_NONE;
    if (pr == PL_NONE) {
      if (sh->nfree < NRQ_MAX_FREE) sh->freex[sh->nfree] = x;
      sh->nfree++;
    } else {
      c.gj_used()[pr] = 1;
      c.red_row[sh->r2] = pr;
      c.red_x[sh->r2] = x;
      sh->r2++;
    }
  }
}

/* ---- the same elimination, a panel of 32 columns (one word of every row) at a time ----
 * A step of the loop above is a pass over the whole matrix and a barrier per column: 16 k clocks at K'=56403 (650 rows of
 * 41 words), 634 columns.  Per panel w instead: (1) the 32 columns are eliminated on the panel word of every row alone
 * (pl_gjp_bid / pl_gjp_step: same pivot rule, one word per row), while a mask per row records which pivots it absorbed;
 * (2) pivot row b at its pivot step is a GF(2) combination gj_A[b] of the pivot rows as they were when the panel began
 * (pl_gjp_comb), so every row ends as itself plus the panel-start pivot rows named by M_j = XOR of gj_A[s] over its mask;
 * (3) those 32 rows are copied aside (HBM, a dead array; pl_gjp_stage) and ONE pass applies them to every other word of
 * every row (pl_gjp_apply).  Same pivots, same result, one full pass per 32 columns.  Masks live in the frontier queues
 * (idle since peeling).  The single-column steps above stay for the column a late extra row may add (planner_seq.h). */
#define PL_NONE16 0xFFFFu
#ifndef PL_GJ_BLOCK_MIN
#define PL_GJ_BLOCK_MIN 4u /* words of the matrix per thread from which the panel form is used */
#endif
/* where the 32 panel-start pivot rows wait for pl_gjp_apply: the per-thread scratch words of the workgroup (LDS; idle between
 * the init scans and the final phases) when they fit, else a dead HBM array (rec_word: the op stream has been emitted) */
SB_HD uint32_t *pl_gj_rows_aside(const PlanCtx &c, uint32_t nt) {
  if (32u * c.sh->rowlen <= nt) return c.part();
  return c.rec_word;
}
SB_HD uint32_t *pl_gj_mask(const PlanCtx &c) { uint32_t *q = reinterpret_cast<uint32_t *>(c.qmem); PL_ASSUME_LDS(q); return q; } /* [2 * qcap] words */
/* (worth it when a pass over the matrix is more than a few words per thread: measured at K=8192 -- 216 rows of 15 words on
 * 1024 threads -- the column-at-a-time loop with its single barrier per column is faster: 0.8 M against 1.2 M clocks) */
SB_HD bool pl_gj_blocked(const PlanCtx &c, uint32_t nt) {
  return c.sh->nlow * c.sh->rowlen >= PL_GJ_BLOCK_MIN * nt && c.sh->nlow + PL_EXTRA_ROWS <= 2u * c.qcap && 32u * c.sh->rowlen <= c.reccap;
}
template <int Z> SB_HD void pl_gjp_init(PlanCtx &c, uint32_t w, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  uint32_t *m = pl_gj_mask(c);
  for (uint32_t j = tid; j < sh->nlow; j += nt) m[j] = 0;
  if (tid < 32u) { sh->gj_pr[tid] = PL_NONE16; sh->gj_A[tid] = 0; }
  if (tid == 0) sh->cand[0] = sh->cand[1] = PL_NONE;
}
/* column x = 32 w + b: the unused row with the lowest index that has it bids */
template <int Z> SB_HD void pl_gjp_bid(PlanCtx &c, uint32_t x, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t *Mb = pl_mb(c);
  const uint32_t rowlen = sh->rowlen, w = x >> 5, b = x & 31u;
  uint32_t best = PL_NONE; /* (one atomic per wave: hundreds of bids on one LDS word are served one after the other) */
  for (uint32_t j = tid; j < sh->nlow; j += nt)
    if (((Mb[(size_t)j * rowlen + w] >> b) & 1u) && !c.gj_used()[j] && j < best) best = j;
  best = PL_WAVE_MIN(best);
  if (best != PL_NONE && PL_WAVE_LEADER(tid)) PL_ATOM_MIN(&sh->cand[x & 1u], best);
}
/* ... and the column leaves the panel word of every other row that has it; one thread keeps the books (the bids of the
 * next column go to the other slot, cleared here) */
template <int Z> SB_HD void pl_gjp_step(PlanCtx &c, uint32_t x, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  uint32_t *Mb = pl_mb(c);
  uint32_t *m = pl_gj_mask(c);
  const uint32_t rowlen = sh->rowlen, w = x >> 5, b = x & 31u, pr = sh->cand[x & 1u];
  if (pr != PL_NONE) {
    const uint32_t pw = Mb[(size_t)pr * rowlen + w]; /* (the pivot row's own word does not change in this step) */
    for (uint32_t j = tid; j < sh->nlow; j += nt) {
      if (j == pr) continue;
      const uint32_t v = Mb[(size_t)j * rowlen + w];
      if ((v >> b) & 1u) { Mb[(size_t)j * rowlen + w] = v ^ pw; m[j] |= 1u << b; }
    }
  }
  if (tid != 0) return;
  sh->cand[(x & 1u) ^ 1u] = PL_NONE;
  if (pr == PL_NONE) {
    if (sh->nfree < NRQ_MAX_FREE) sh->freex[sh->nfree] = x;
    sh->nfree++;
  } else {
    c.gj_used()[pr] = 1;
    c.red_row[sh->r2] = pr;
    c.red_x[sh->r2] = x;
    sh->r2++;
    sh->gj_pr[b] = (uint16_t)pr;
  }
}
/* Steps (1) of all 32 columns of a panel by ONE wave, in one phase.  pl_gjp_bid / pl_gjp_step are two phases per column --
 * a shared atomic-min, two barriers, ~3 k clocks each, for a pass over ONE word per row: 64 phases per panel.  One wave holds
 * the panel word of every row in registers (row j with lane j % 64, PL_GJW_ROWS rows per lane), finds a column's pivot by a
 * minimum over the wave, takes the pivot row's word from its lane, and XORs: ~100 instructions per column, no barrier, no
 * shared word.  Same pivot rule (the unused row with the lowest index), same masks, same books -- the plan is the same.
 * The emulator runs the plain loop. */
#ifndef PL_GJ_WAVE
#define PL_GJ_WAVE 1
#endif
#ifndef PL_GJW_ROWS
#define PL_GJW_ROWS 12u
#endif
#ifndef PL_GJW_MIN_ROWS
#define PL_GJW_MIN_ROWS 64u /* leftover rows from which the panel form with the wave phase is used (small blocks keep the column loop) */
#endif
SB_HD bool pl_gjw_ok(const PlanCtx &c) {
  return c.sh->nlow >= PL_GJW_MIN_ROWS && c.sh->nlow <= 64u * PL_GJW_ROWS && c.sh->nlow + PL_EXTRA_ROWS <= 2u * c.qcap && 32u * c.sh->rowlen <= c.reccap;
}
template <uint32_t R> SB_HD void pl_gjp_wave_r(PlanCtx &c, uint32_t w, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  uint32_t *Mb = pl_mb(c);
  uint32_t *m = pl_gj_mask(c);
  const uint32_t rowlen = sh->rowlen, nlow = sh->nlow, u = c.p.L - sh->npiv;
  const uint32_t nb = u - 32u * w < 32u ? u - 32u * w : 32u; /* columns of this panel */
#if defined(__HIP_DEVICE_COMPILE__)
  (void)nt;
  if (tid >= 64u) return;
  uint32_t v[R], mk[R], used = 0u;
  /* mk[k]: which PANEL-START pivot rows have been XORed into the lane's k-th row so far -- kept in those terms from the start
   * (a row that absorbs pivot b absorbs what that pivot row had absorbed: its mask, and itself), so that no combination step
   * (pl_gjp_comb) is needed afterwards: gj_A is the identity */
  if (tid < 32u) { sh->gj_pr[tid] = PL_NONE16; sh->gj_A[tid] = 1u << tid; }
#pragma unroll
  for (uint32_t k = 0; k < R; k++) {
    const uint32_t j = tid + 64u * k;
    v[k] = j < nlow ? Mb[(size_t)j * rowlen + w] : 0u;
    mk[k] = 0u;
    if (j < nlow && c.gj_used()[j]) used |= 1u << k;
  }
  uint32_t r2 = sh->r2, nfree = sh->nfree;
  for (uint32_t b = 0; b < nb; b++) {
    uint32_t best = PL_NONE;
#pragma unroll
    for (uint32_t k = R; k-- > 0u;) /* (descending: the lowest k that qualifies stays) */
      if (((v[k] >> b) & 1u) && !((used >> k) & 1u)) best = tid + 64u * k;
    const uint32_t pr = PL_WAVE_MIN(best);
    if (pr == PL_NONE) { /* no unused row has the column: free */
      if (tid == 0u && nfree < NRQ_MAX_FREE) sh->freex[nfree] = 32u * w + b;
      nfree++;
      continue;
    }
    const uint32_t owner = pr & 63u, kk = pr >> 6;
    uint32_t mine = 0u, mmine = 0u;
#pragma unroll
    for (uint32_t k = 0; k < R; k++) { mine = k == kk ? v[k] : mine; mmine = k == kk ? mk[k] : mmine; }
    const uint32_t pw = (uint32_t)__shfl((int)mine, (int)owner);
    const uint32_t pa = (uint32_t)__shfl((int)mmine, (int)owner) ^ (1u << b); /* the pivot row in terms of panel-start rows */
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
      const bool hit = ((v[k] >> b) & 1u) && !(tid == owner && k == kk);
      v[k] ^= hit ? pw : 0u;
      mk[k] ^= hit ? pa : 0u;
    }
    if (tid == owner) used |= 1u << kk;
    if (tid == 0u) { c.red_row[r2] = pr; c.red_x[r2] = 32u * w + b; sh->gj_pr[b] = (uint16_t)pr; }
    r2++;
  }
#pragma unroll
  for (uint32_t k = 0; k < R; k++) {
    const uint32_t j = tid + 64u * k;
    if (j < nlow) { Mb[(size_t)j * rowlen + w] = v[k]; m[j] = mk[k]; c.gj_used()[j] = (uint8_t)((used >> k) & 1u); }
  }
  if (tid == 0u) { sh->r2 = r2; sh->nfree = nfree; }
#else
  if (tid != 0u) return;
  (void)nt;
  for (uint32_t b = 0; b < 32u; b++) { sh->gj_pr[b] = PL_NONE16; sh->gj_A[b] = 1u << b; }
  for (uint32_t j = 0; j < nlow; j++) m[j] = 0u;
  for (uint32_t b = 0; b < nb; b++) {
    const uint32_t x = 32u * w + b;
    uint32_t pr = PL_NONE;
    for (uint32_t j = 0; j < nlow; j++)
      if (((Mb[(size_t)j * rowlen + w] >> b) & 1u) && !c.gj_used()[j]) { pr = j; break; }
    if (pr == PL_NONE) {
      if (sh->nfree < NRQ_MAX_FREE) sh->freex[sh->nfree] = x;
      sh->nfree++;
      continue;
    }
    const uint32_t pw = Mb[(size_t)pr * rowlen + w], pa = m[pr] ^ (1u << b);
    for (uint32_t j = 0; j < nlow; j++) {
      if (j == pr) continue;
      const uint32_t val = Mb[(size_t)j * rowlen + w];
      if ((val >> b) & 1u) { Mb[(size_t)j * rowlen + w] = val ^ pw; m[j] ^= pa; }
    }
    c.gj_used()[pr] = 1;
    c.red_row[sh->r2] = pr;
    c.red_x[sh->r2] = x;
    sh->r2++;
    sh->gj_pr[b] = (uint16_t)pr;
  }
#endif
}
template <int Z> SB_HD void pl_gjp_wave(PlanCtx &c, uint32_t w, uint32_t tid, uint32_t nt) {
  /* (rows per lane as a compile-time bound: the column loop is ~12 instructions per row a lane may hold) */
  const uint32_t nlow = c.sh->nlow;
  if (nlow <= 64u * 4u) pl_gjp_wave_r<4u>(c, w, tid, nt);
  else if (nlow <= 64u * 8u) pl_gjp_wave_r<8u>(c, w, tid, nt);
  else pl_gjp_wave_r<PL_GJW_ROWS>(c, w, tid, nt);
}
/* pivot row b when it was used = the panel-start pivot rows named by gj_A[b]: itself and what it had absorbed before */
template <int Z> SB_HD void pl_gjp_comb(PlanCtx &c, uint32_t w, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid != 0) return;
  const uint32_t *m = pl_gj_mask(c);
  for (uint32_t b = 0; b < 32u; b++) {
    if (sh->gj_pr[b] == PL_NONE16) continue;
    uint32_t a = 1u << b, before = m[sh->gj_pr[b]] & ((1u << b) - 1u);
    while (before) { const uint32_t s = (uint32_t)__builtin_ctz(before); before &= before - 1u; a ^= sh->gj_A[s]; }
    sh->gj_A[b] = a;
  }
}
/* the 32 panel-start pivot rows aside (words other than the panel's: those are final), every row's mask in terms of them */
template <int Z> SB_HD void pl_gjp_stage(PlanCtx &c, uint32_t w, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t *Mb = pl_mb(c);
  uint32_t *m = pl_gj_mask(c);
  const uint32_t rowlen = sh->rowlen;
  uint32_t *P = pl_gj_rows_aside(c, nt);
  for (uint32_t e = tid; e < 32u * rowlen; e += nt) {
    const uint32_t b = e / rowlen, wd = e - b * rowlen, pr = sh->gj_pr[b];
    P[e] = pr == PL_NONE16 ? 0u : Mb[(size_t)pr * rowlen + wd];
  }
  for (uint32_t j = tid; j < sh->nlow; j += nt) {
    uint32_t mj = m[j], M = 0;
    while (mj) { const uint32_t s = (uint32_t)__builtin_ctz(mj); mj &= mj - 1u; M ^= sh->gj_A[s]; }
    m[j] = M;
  }
}
template <int Z> SB_HD void pl_gjp_apply(PlanCtx &c, uint32_t w, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  uint32_t *Mb = pl_mb(c);
  const uint32_t *m = pl_gj_mask(c);
  const uint32_t *P = pl_gj_rows_aside(c, nt);
  const uint32_t rowlen = sh->rowlen, total = sh->nlow * rowlen;
  const uint32_t inv = 0xFFFFFFFFu / rowlen + 1u; /* e / rowlen == mulhi(e, inv) for e < 2^16 * rowlen */
  for (uint32_t e = tid; e < total; e += nt) {
    const uint32_t j = (uint32_t)(((uint64_t)e * inv) >> 32), wd = e - j * rowlen;
    uint32_t M = m[j];
    if (wd == w || !M) continue;
    uint32_t v = Mb[e];
    while (M) { const uint32_t s = (uint32_t)__builtin_ctz(M); M &= M - 1u; v ^= P[s * rowlen + wd]; }
    Mb[e] = v;
  }
}

/* The GF(2) combinations E_q (slot M+q) = XOR of the leftover rows named by the augmented part of reduced row q, in one of
 * two forms (plan.h off_augt): small blocks (pl_bin_in_stream) keep them as one more accumulate-only group of the op stream
 * -- a few rows of the forward wave -- big blocks hand them over as a bit matrix that the solve kernel applies with XOR
 * tables by the whole workgroup (a fixed ~7 k clocks of trips and barriers per strip, against r2 * nlow / 2 ops). */
/* the GF(2) combinations E_q (slot M+q) as one more accumulate-only group of XOR ops */
SB_HD void pl_binops_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid == 0) {
    sh->tmp1 = 0;
  }
  const uint32_t *Mb = pl_mb(c);
  for (uint32_t q = tid; q < sh->r2; q += nt) {
    const uint32_t *aug = Mb + (size_t)c.red_row[q] * sh->rowlen + sh->wpr;
    uint32_t n = 0;
    for (uint32_t w = 0; w < sh->lpr; w++) n += (uint32_t)__builtin_popcount(aug[w]);
    c.pivdeg[q] = n; /* reuse */
  }
}
SB_HD void pl_binops_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (tid != 0) return;
  /* the ops of reduced row q (target: scratch row M + q) rank among those of the target's lane class (plan.h "lane placement") */
  uint32_t run = 0;
  for (uint32_t d = 0; d < NRQ_LANE_CLASSES; d++) sh->bin_ct[d] = 0;
  for (uint32_t q = 0; q < sh->r2; q++) {
    const uint32_t n = c.pivdeg[q], d = nrq_op_class(NRQ_OP(sh->M + q, 0u));
    c.pivdeg[q] = sh->bin_ct[d]; sh->bin_ct[d] += n; run += n;
  }
  const uint32_t g = sh->nlev + 1u;
  c.lev_ops[g] = run;
  sh->nrows = sh->tmp0 + pl_group_rows(run);
  if ((sh->nrows + NRQ_PAD_ROWS > sh->opbase || sh->M + sh->r2 + NRQ_SCRATCH > 65535u) && sh->status == 0)
    (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); /* op fields are 16 bits */
}
SB_HD void pl_binops_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  uint32_t *ops = reinterpret_cast<uint32_t *>(c.arena + sh->off_ops);
  const uint32_t row0 = sh->tmp0, span = sh->nrows - sh->tmp0;
  uint32_t ct[NRQ_LANE_CLASSES];
  for (uint32_t d = 0; d < NRQ_LANE_CLASSES; d++) ct[d] = sh->bin_ct[d];
  const uint32_t *Mb = pl_mb(c);
  for (uint32_t q = tid; q < sh->r2; q += nt) {
    const uint32_t *aug = Mb + (size_t)c.red_row[q] * sh->rowlen + sh->wpr;
    uint32_t i = c.pivdeg[q];
    for (uint32_t w = 0; w < sh->lpr; w++) {
      uint32_t bits = aug[w];
      while (bits) {
        const uint32_t j = w * 32u + (uint32_t)__builtin_ctz(bits);
        bits &= bits - 1u;
        const uint32_t word = NRQ_OP(sh->M + q, c.lowslot[j]);
        *pl_op_at(ops, row0, nrq_lane_place(span, ct, nrq_op_class(word), i)) = word;
        i++;
      }
    }
  }
}

/* the GF(2) combinations E_q (slot M+q) = XOR of the leftover rows named by the augmented part of reduced row q: handed to
 * the solve kernel as a bit matrix, word w of row q at [w * aug_stride + q] (plan.h off_augt), behind the op stream */
template <int Z> SB_HD void pl_bin_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (pl_bin_in_stream(c.p.L)) { if (tid == 0) { sh->off_augt = 0; sh->aug_stride = 0; } pl_binops_a(c, tid, nt); return; }
  if (tid == 0) {
    sh->tmp1 = 0;
    sh->aug_stride = (sh->r2 + 3u) & ~3u;
    if (sh->aug_stride < 4u) sh->aug_stride = 4u;
    sh->off_augt = sh->arena_top;
    sh->arena_top = pl_r16(sh->off_augt + sh->lpr * sh->aug_stride * 4u);
    sh->nrows = sh->tmp0; /* rows of the stream: the level groups and the spare rows */
    if ((sh->arena_top > c.job.arena_cap || sh->nrows + NRQ_PAD_ROWS > sh->opbase || sh->M + sh->r2 + NRQ_SCRATCH > 65535u) && sh->status == 0)
      (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); /* (op fields and slot numbers are 16 bits) */
  }
}
template <int Z> SB_HD void pl_bin_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (pl_bin_in_stream(c.p.L)) { pl_binops_b(c, tid, nt); return; }
  if (sh->status) return;
  const uint32_t *Mb = pl_mb(c);
  uint32_t *augt = reinterpret_cast<uint32_t *>(c.arena + sh->off_augt);
  const uint32_t lpr = sh->lpr, stride = sh->aug_stride;
  uint32_t n = 0;
  for (uint32_t e = tid; e < lpr * stride; e += nt) {
    const uint32_t w = e / stride, q = e - w * stride;
    const uint32_t v = q < sh->r2 ? Mb[(size_t)c.red_row[q] * sh->rowlen + sh->wpr + w] : 0u;
    augt[e] = v;
    n += (uint32_t)__builtin_popcount(v);
  }
  if (n) PL_ATOM_ADD(&sh->tmp1, n); /* terms, for the statistics */
}
template <int Z> SB_HD void pl_bin_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (pl_bin_in_stream(c.p.L)) { pl_binops_c(c, tid, nt); return; }
  if (tid == 0) c.lev_ops[sh->nlev + 1u] = sh->status ? 0u : sh->tmp1;
}

/* =============================== phase 7: the free columns over GF(256) ====================== */
/* coefficient columns mh[h][q], free-column masks fbits[q], and the H x (nfree+H) augmented system */
template <int Z> SB_HD void pl_dense_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  /* more free columns than HDPC rows: rank deficient for sure (decided identically by every thread) */
  const bool feasible = sh->nfree <= c.p.H && sh->nfree <= NRQ_MAX_FREE;
  if (tid == 0) sh->dense_ok = feasible ? 1u : 0u;
  if (!feasible) return;
  const uint32_t H = c.p.H, r2 = sh->r2, nfree = sh->nfree;
  const uint32_t *Mb = pl_mb(c);
  const uint8_t *Mh = pl_mhm(c);
  for (uint32_t q = tid; q < r2; q += nt) {
    const uint32_t *row = Mb + (size_t)c.red_row[q] * sh->rowlen;
    uint32_t fb = 0;
    for (uint32_t f = 0; f < nfree; f++) {
      const uint32_t x = sh->freex[f];
      if ((row[x >> 5] >> (x & 31u)) & 1u) fb |= 1u << f;
    }
    c.fbits[q] = fb;
    c.pivx[q] = (uint16_t)c.red_x[q];
    for (uint32_t h = 0; h < H; h++) c.mh[(size_t)h * r2 + q] = Mh[(size_t)c.red_x[q] * 16u + h];
  }
  for (uint32_t f = tid; f < nfree; f += nt) c.freex_out[f] = (uint16_t)sh->freex[f];
  for (uint32_t h = tid; h < PL_MAXH; h += nt) sh->taken[h] = 0;
}
template <int Z> SB_HD void pl_dense_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status || !sh->dense_ok) return;
  const uint32_t H = c.p.H, r2 = sh->r2, nfree = sh->nfree, aw = nfree + H;
  const uint8_t *Mh = pl_mhm(c);
  for (uint32_t e = tid; e < H * aw; e += nt) {
    const uint32_t h = e / aw, w = e - h * aw;
    uint8_t v;
    if (w < nfree) {
      v = Mh[(size_t)sh->freex[w] * 16u + h];
      for (uint32_t q = 0; q < r2; q++)
        if ((c.fbits[q] >> w) & 1u) v ^= c.mh[(size_t)h * r2 + q];
    } else {
      v = (uint8_t)((w - nfree) == h);
    }
    sh->aug[e] = v;
  }
}
/* one elimination step on the augmented system: column f.  Thread w owns augmented column w. */
template <int Z> SB_HD void pl_dense_step_a(PlanCtx &c, uint32_t f, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status || !sh->dense_ok || tid != 0) return;
  const uint32_t H = c.p.H, aw = sh->nfree + H;
  uint32_t pr = PL_NONE;
  for (uint32_t h = 0; h < H; h++)
    if (!sh->taken[h] && sh->aug[h * aw + f]) { pr = h; break; }
  if (pr == PL_NONE) { sh->dense_ok = 0; return; } /* the HDPC rows cannot resolve column f: rank deficient */
  sh->taken[pr] = 1;
  sh->solver[f] = (uint8_t)pr;
  sh->tmp1 = pr;
  for (uint32_t h = 0; h < H; h++) sh->colf[h] = sh->aug[h * aw + f]; /* snapshot of the pivot column */
}
template <int Z> SB_HD void pl_dense_step_b(PlanCtx &c, uint32_t f, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status || !sh->dense_ok) return;
  const uint32_t H = c.p.H, aw = sh->nfree + H, pr = sh->tmp1;
  if (tid >= aw) return;
  /* thread = augmented column; the pivot column is read from the snapshot step A took */
  const uint8_t *colf = sh->colf;
  (void)f;
  const uint8_t piv = colf[pr];
  const uint8_t inv = sh->gf_exp[255u - sh->gf_log[piv]];
  const uint8_t scaled = pl_gfmul(sh, sh->aug[pr * aw + tid], inv);
  for (uint32_t h = 0; h < H; h++) {
    if (h == pr) continue;
    if (colf[h]) sh->aug[h * aw + tid] ^= pl_gfmul(sh, colf[h], scaled);
  }
  sh->aug[pr * aw + tid] = scaled;
}
template <int Z> SB_HD void pl_dense_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status == 0 && !sh->dense_ok && tid == 0) sh->status = PL_FAIL_SINGULAR; /* and no symbol left to add */
  if (sh->status || !sh->dense_ok) return;
  const uint32_t H = c.p.H, nfree = sh->nfree, aw = nfree + H;
  for (uint32_t e = tid; e < nfree * H; e += nt) {
    const uint32_t f = e / H, h = e - f * H;
    c.hinv[e] = sh->aug[sh->solver[f] * aw + nfree + h];
  }
}

/* =============================== phase 7b: one more symbol for a rank-deficient block ======= */
/* The system is rank deficient with the symbols used so far.  If the caller holds further repair symbols
 * (job.nrep_avail), take the next one as an additional constraint row WITHOUT redoing the peeling: its
 * data ops go into the spare chunks of the op stream, its coefficient row over the inactive columns is
 * reduced against the Gauss-Jordan state, and if something is left it pivots one free column. */
template <int Z> SB_HD void pl_extra_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status || tid != 0) return;
  const rq_params &p = c.p;
  const uint32_t i = sh->npatch;
  if (i >= c.job.nrep_avail || sh->nextra >= PL_EXTRA_ROWS) { sh->status = PL_FAIL_SINGULAR; return; }
  const uint32_t esi = c.rep_esi[i];
  if (esi < p.K || esi >= (1u << 24)) { sh->status = PL_FAIL_SINGULAR; return; }
  const uint32_t row = sh->M, j = sh->nlow;
  if (row + 1u > c.Mcap || i + 1u > c.npcap || j + 1u > c.lowcap) { (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); return; }
  uint32_t cols[RQ_MAX_LT_COLS];
  const uint32_t n = rq_lt_columns(&p, esi + (p.Kp - p.K), cols);
  uint16_t *dst = c.patch_cols + (size_t)i * PL_PATCH_STRIDE;
  uint32_t *ops = reinterpret_cast<uint32_t *>(c.arena + sh->off_ops);
  for (uint32_t k = 0; k < n; k++) {
    dst[k] = (uint16_t)cols[k];
    const uint32_t info = c.colinfo[cols[k]];
    if ((info >> 30) == PL_ST_PIVOT) {
      if (sh->spare_fill >= PL_SPARE_ROWS * NRQ_ROW) { (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY); return; }
      *pl_op_at(ops, sh->spare_base, sh->spare_fill++) = NRQ_OP(row, c.pivslot[info & 0x3FFFFFFFu]);
    }
  }
  c.patch_len[i] = (uint8_t)n;
  c.patch_of[row] = (uint16_t)i;
  c.rowinfo[row] = PL_UNASSIGNED | PL_PATCHED;
  c.lowslot[j] = (uint16_t)row;
  c.gj_used()[j] = 0;
  sh->M = row + 1u; sh->npatch = i + 1u; sh->nlow = j + 1u; sh->nextra++;
  sh->cand[0] = sh->cand[1] = sh->cand[2] = PL_NONE;
}
/* its bit row over the inactive columns: own inactive entries plus the W rows of its pivot columns */
template <int Z> SB_HD void pl_extra_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t wpr = sh->wpr, rowlen = sh->rowlen, j = sh->nlow - 1u, row = c.lowslot[j];
  if (tid >= rowlen) return;
  uint32_t acc = 0;
  if (tid < wpr) {
    const uint16_t *cols;
    const uint32_t n = pl_row(c, row, &cols);
    for (uint32_t k = 0; k < n; k++) {
      const uint32_t info = c.colinfo[cols[k]], idx = info & 0x3FFFFFFFu;
      if ((info >> 30) == PL_ST_INACT) { if ((idx >> 5) == tid) acc ^= 1u << (idx & 31u); }
      else acc ^= c.wrows[(size_t)c.pivslot[idx] * wpr + tid];
    }
    c.wrows[(size_t)row * wpr + tid] = acc;
  } else {
    acc = ((tid - wpr) == (j >> 5)) ? (1u << (j & 31u)) : 0u;
  }
  sh->xrow[tid] = acc;
}
/* reduce it against the pivots found so far (pivot rows are fully reduced, so the order does not matter) */
template <int Z> SB_HD void pl_extra_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const uint32_t rowlen = sh->rowlen, j = sh->nlow - 1u;
  if (tid >= rowlen) return;
  uint32_t *Mb = pl_mb(c);
  uint32_t acc = sh->xrow[tid];
  for (uint32_t q = 0; q < sh->r2; q++) {
    const uint32_t x = c.red_x[q];
    if ((sh->xrow[x >> 5] >> (x & 31u)) & 1u) acc ^= Mb[(size_t)c.red_row[q] * rowlen + tid];
  }
  Mb[(size_t)j * rowlen + tid] = acc;
}
/* whatever is left sits in free columns: the first of them becomes this row's pivot column */
template <int Z> SB_HD void pl_extra_d(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status || tid != 0) return;
  const uint32_t *row = pl_mb(c) + (size_t)(sh->nlow - 1u) * sh->rowlen;
  sh->xcol = PL_NONE;
  const uint32_t nf = sh->nfree < NRQ_MAX_FREE ? sh->nfree : NRQ_MAX_FREE;
  for (uint32_t f = 0; f < nf; f++) {
    const uint32_t x = sh->freex[f];
    if ((row[x >> 5] >> (x & 31u)) & 1u) {
      sh->xcol = x;
      for (uint32_t g = f; g + 1u < nf; g++) sh->freex[g] = sh->freex[g + 1u];
      sh->nfree--; /* nfree may exceed the list capacity; entries beyond it were never recorded */
      break;
    }
  }
}

template <int Z> SB_HD void pl_mark_failed(PlanCtx &c, uint32_t tid, uint32_t nt) {
  if (tid == 0) (c.sh->fail_site = __LINE__, c.sh->status = PL_FAIL_CAPACITY);
}

/* =============================== phase 8: maps, W image, job ================================= */
template <int Z> SB_HD void pl_final_a(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const rq_params &p = c.p;
  const uint32_t n_hd = p.Kp + p.S, u = p.L - sh->npiv;
  for (uint32_t col = tid; col < ((n_hd + 7u) & ~7u); col += nt) c.pivof[col] = NRQ_NOSLOT;
  if (tid == 0) {
    /* W image and the per-block tail (rowsrc, output lists) go after the op stream */
    sh->tmp0 = (sh->npiv + 63u) & ~63u; /* npiv_pad */
  }
  /* homes of the inactive columns: the rows that did not become pivots, any one-to-one assignment */
  pl_for_batched(tid, nt, sh->M, [&](uint32_t r) { return c.rowinfo[r]; }, [&](uint32_t r, uint32_t info) {
    if (!(info & PL_UNASSIGNED)) return;
    uint32_t x = PL_ATOM_ADD(&sh->uslot_fill, 1u);
    if (x < u) { c.uslot[x] = (uint16_t)r; c.colslot[c.ucol[x]] = (uint16_t)r; }
  });
}
template <int Z> SB_HD void pl_final_b(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  struct CS { uint32_t col, slot; };
#if defined(__HIP_DEVICE_COMPILE__)
  /* the two maps by column through LDS when they fit the dense stage's region (done with by now): 2 x npiv two-byte stores
   * scattered over the maps are partial lines for the memory to patch, like the op words of pl_ops_emit_tiled; built in LDS the
   * maps go out as whole lines */
  const uint32_t n_hd8 = (c.p.Kp + c.p.S + 7u) & ~7u, Lp = (c.p.L + 7u) & ~7u;
  if (c.dense_lds && (Lp + n_hd8) * 2u <= c.dense_bytes) { /* (uniform: every thread takes the same way) */
    uint16_t *cs = reinterpret_cast<uint16_t *>(c.dense_lds), *po = cs + Lp;
    PL_ASSUME_LDS(cs); PL_ASSUME_LDS(po);
    const uint32_t u = c.p.L - sh->npiv;
    for (uint32_t col = tid; col < n_hd8; col += nt) po[col] = NRQ_NOSLOT;
    for (uint32_t col = tid; col < Lp; col += nt) cs[col] = 0;
    __syncthreads();
    for (uint32_t x = tid; x < u; x += nt) cs[c.ucol[x]] = c.uslot[x]; /* (pl_final_a's homes of the inactive columns) */
    pl_for_batched(tid, nt, sh->npiv, [&](uint32_t k) { return CS{c.pivcol[k], c.pivslot[k]}; },
                   [&](uint32_t, CS v) { cs[v.col] = (uint16_t)v.slot; if (v.col < n_hd8) po[v.col] = (uint16_t)v.slot; });
    __syncthreads();
    for (uint32_t col = tid; col < c.p.L; col += nt) c.colslot[col] = cs[col];
    for (uint32_t col = tid; col < n_hd8; col += nt) c.pivof[col] = po[col];
  } else
#endif
  pl_for_batched(tid, nt, sh->npiv, [&](uint32_t k) { return CS{c.pivcol[k], c.pivslot[k]}; },
                 [&](uint32_t, CS v) { c.colslot[v.col] = (uint16_t)v.slot; c.pivof[v.col] = (uint16_t)v.slot; });
  if (tid == 0) {
    const rq_params &p = c.p;
    const uint32_t nl = c.job.nlost;
    uint32_t o = sh->arena_top;
    c.part()[0] = o; o = pl_r16(o + sh->wpr * sh->tmp0 * 4u);          /* wt */
    c.part()[1] = o; o = pl_r16(o + sh->M * 4u);                       /* rowsrc */
    c.part()[2] = o; o = pl_r16(o + (nl + 1u) * 4u);                   /* out_cptr */
    c.part()[3] = o; o = pl_r16(o + nl * 4u + 4u);                     /* out_row */
    c.part()[4] = o; o = pl_r16(o + nl * PL_PATCH_STRIDE * 2u + NRQ_STORE_SLACK);  /* out_slots (+ what ph_store reads past a list) */
    sh->arena_top = o;
    if (o > c.job.arena_cap) (sh->fail_site = __LINE__, sh->status = PL_FAIL_CAPACITY);
    sh->tmp1 = 0; /* (sum of the level groups' op counts: pl_final_c) */
    (void)p;
  }
}
template <int Z> SB_HD void pl_final_c(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const rq_params &p = c.p;
  const uint32_t wpr = sh->wpr, stride = sh->tmp0, nl = c.job.nlost;
  uint32_t *wt = reinterpret_cast<uint32_t *>(c.arena + c.part()[0]);
  if (!sh->defer_wt) /* (a segmented run leaves the transposition to nrq_wt_kernel: many workgroups) */
    for (uint32_t e = tid; e < wpr * stride; e += nt) {
      const uint32_t w = e / stride, k = e - w * stride;
      wt[e] = k < sh->npiv ? c.wrows[(size_t)c.pivslot[k] * wpr + w] : 0u;
    }
  uint32_t *rowsrc = reinterpret_cast<uint32_t *>(c.arena + c.part()[1]);
  pl_for_batched(tid, nt, sh->M, [&](uint32_t r) { return (uint32_t)c.patch_of[r]; }, [&](uint32_t r, uint32_t pi) {
    uint32_t v = NRQ_ROW_ZERO;
    if (pi != 0xFFFFu) v = NRQ_ROW_REP | pi;
    else if (r >= p.S + p.H && r < p.S + p.H + p.K) v = r - p.S - p.H;
    rowsrc[r] = v;
  });
  /* the missing source symbols as LT combinations of slots (ISI of a source symbol = its ESI) */
  uint32_t *cptr = reinterpret_cast<uint32_t *>(c.arena + c.part()[2]);
  uint32_t *orow = reinterpret_cast<uint32_t *>(c.arena + c.part()[3]);
  uint16_t *osl = reinterpret_cast<uint16_t *>(c.arena + c.part()[4]);
  /* list lengths of the missing symbols: to LDS when they fit (the frontier queues are free by now), so that the
   * running sum in pl_final_d -- one thread -- does not walk an array in HBM */
  uint16_t *degq = c.queue(0u);
  const bool in_lds = nl <= 2u * c.qcap;
  struct ED { uint32_t e, dg; };
  pl_for_batched(tid, nt, nl, [&](uint32_t g) { const uint32_t e = c.lost[g]; return ED{e, c.b_rptr[p.S + p.H + e + 1] - c.b_rptr[p.S + p.H + e]}; },
                 [&](uint32_t g, ED v) { if (in_lds) degq[g] = (uint16_t)v.dg; else c.pivdeg[g] = v.dg; orow[g] = v.e; });
  (void)cptr; (void)osl;
  /* ops of the stream, summed by everybody (one thread walking nlev + 2 words of HBM: a trip each, 1.5 M clocks at K'=56403) */
  uint32_t ops = 0;
  pl_for_batched(tid, nt, sh->nlev + 2u, [&](uint32_t l) { return c.lev_ops[l]; }, [&](uint32_t, uint32_t v) { ops += v; });
  if (ops) PL_ATOM_ADD(&sh->tmp1, ops);
}
/* many missing symbols (more than the frontier queues hold lengths of): the list offsets by a three-step scan over pivdeg[]
 * instead of a running sum by one thread */
template <int Z> SB_HD void pl_final_c2(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const uint32_t nl = c.job.nlost;
  if (sh->status || nl <= 2u * c.qcap) return;
  const uint32_t per = (nl + nt - 1u) / nt, a = tid * per < nl ? tid * per : nl, b = a + per < nl ? a + per : nl;
  uint32_t sum = 0;
  for (uint32_t g0 = a; g0 < b; g0 += PL_BATCH) {
    uint32_t v[PL_BATCH];
#pragma unroll
    for (uint32_t j = 0; j < PL_BATCH; j++) v[j] = c.pivdeg[g0 + j < b ? g0 + j : g0];
#pragma unroll
    for (uint32_t j = 0; j < PL_BATCH; j++) if (g0 + j < b) sum += v[j];
  }
  reinterpret_cast<uint32_t *>(c.queue(0u))[tid] = sum; /* (the frontier queues are free by now; part()[0..4] hold the arena offsets) */
}
template <int Z> SB_HD void pl_final_c3(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status || c.job.nlost <= 2u * c.qcap || tid != 0) return;
  uint32_t *ps = reinterpret_cast<uint32_t *>(c.queue(0u));
  uint32_t run = 0;
  for (uint32_t t = 0; t < nt; t++) { const uint32_t v = ps[t]; ps[t] = run; run += v; }
  ps[nt] = run; /* the sum of all list lengths */
}
template <int Z> SB_HD void pl_final_d(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  const rq_params &p = c.p;
  if (!sh->status && c.job.nlost <= 2u * c.qcap) {
    /* the list offsets of the missing symbols: every thread sums the lengths before its own entry (LDS reads, no
     * stores in between) -- a running sum by one thread is a chain of LDS round trips, one per entry */
    uint32_t *cptr = reinterpret_cast<uint32_t *>(c.arena + c.part()[2]);
    for (uint32_t g = tid; g <= c.job.nlost; g += nt) cptr[g] = pl_deg_prefix(c, g);
  } else if (!sh->status) { /* (pl_final_c2 / c3 left every thread's starting offset in the queues) */
    const uint32_t nl = c.job.nlost, per = (nl + nt - 1u) / nt, a = tid * per < nl ? tid * per : nl, b = a + per < nl ? a + per : nl;
    const uint32_t *ps = reinterpret_cast<const uint32_t *>(c.queue(0u));
    uint32_t *cptr = reinterpret_cast<uint32_t *>(c.arena + c.part()[2]);
    uint32_t run = ps[tid];
    for (uint32_t g0 = a; g0 < b; g0 += PL_BATCH) {
      uint32_t v[PL_BATCH];
#pragma unroll
      for (uint32_t j = 0; j < PL_BATCH; j++) v[j] = c.pivdeg[g0 + j < b ? g0 + j : g0];
#pragma unroll
      for (uint32_t j = 0; j < PL_BATCH; j++) if (g0 + j < b) { cptr[g0 + j] = run; run += v[j]; }
    }
    if (tid == 0) cptr[nl] = ps[nt];
  }
  if (tid != 0) return;
  nrq_plan_hdr h;
  memset(&h, 0, sizeof(h));
  h.magic = NRQ_PLAN_MAGIC;
  h.status = sh->status ? 1u : 0u;
  h.reserved[0] = sh->status; /* PL_FAIL_* reason */
  h.reserved[1] = sh->nextra; /* repair symbols taken beyond job.nrep */
  h.fail_site = sh->fail_site;
  h.K = p.Kp; h.Kp = p.Kp; h.S = p.S; h.H = p.H; h.W = p.W; h.L = p.L; h.P = p.P; h.B = p.B;
  h.M = sh->M; h.npiv = sh->npiv; h.u = p.L - sh->npiv; h.nlow = sh->nlow; h.r2 = sh->r2; h.nfree = sh->nfree;
  h.nlev = sh->nlev; h.nrows = sh->nrows; h.pipe = NRQ_PIPE; h.wpr = sh->wpr;
  h.npiv_pad = sh->tmp0;
  h.off_ops = sh->off_ops; h.off_pivslot = c.off_pivslot; h.off_pivcol = c.off_pivcol; h.off_wt = c.part()[0];
  h.off_lowslot = c.off_lowslot; h.off_pivx = c.off_pivx; h.off_fbits = c.off_fbits; h.off_mh = c.off_mh;
  h.off_freex = c.off_freex; h.off_hinv = c.off_hinv; h.off_colslot = c.off_colslot; h.off_pivof = c.off_pivof;
  h.off_uslot = c.off_uslot; h.total_bytes = sh->arena_top;
  h.off_augt = sh->off_augt; h.lpr = pl_bin_in_stream(p.L) ? 0u : sh->lpr; h.aug_stride = sh->aug_stride;
  if (!sh->status) {
    const uint32_t nl = c.job.nlost;
    uint32_t *cptr = reinterpret_cast<uint32_t *>(c.arena + c.part()[2]);
    uint32_t run = 0;
    if (nl <= 2u * c.qcap) run = pl_deg_prefix(c, nl);
    else run = reinterpret_cast<const uint32_t *>(c.queue(0u))[nt];
    (void)cptr;
    h.n_xor_ops = sh->tmp1 + run + sh->spare_fill; /* (tmp1: the level groups' ops, summed in pl_final_c) */
  }
  *c.hdr = h;
  if (c.job.hdr_out) *PL_HBM(nrq_plan_hdr, c.job.hdr_out) = h; /* (the batch's headers side by side: one copy back to the host) */
  if (c.jobout) {
    nrq_job j;
    memset(&j, 0, sizeof(j));
    j.plan = (uint64_t)(uintptr_t)c.arena;
    if (!sh->status) {
      j.rowsrc = (uint64_t)(uintptr_t)(c.arena + c.part()[1]);
      j.src = c.job.src; j.rep = c.job.rep; j.inter = c.job.inter; j.out = c.job.src;
      j.out_cptr = (uint64_t)(uintptr_t)(c.arena + c.part()[2]);
      j.out_row = (uint64_t)(uintptr_t)(c.arena + c.part()[3]);
      j.out_slots = (uint64_t)(uintptr_t)(c.arena + c.part()[4]);
      j.nout = c.job.nlost;
    }
    *c.jobout = j;
  }
}

/* W transposed by word (plan.h off_wt), elements e0, e0 + step, ...: what pl_final_c does in an unsegmented run and
 * nrq_wt_kernel (many workgroups) after a segmented one -- from the finished header and the block's W rows */
SB_HD void pl_wt_fill(uint8_t *arena, const uint32_t *wrows, uint32_t e0, uint32_t step) {
  const nrq_plan_hdr *h = reinterpret_cast<const nrq_plan_hdr *>(arena);
  if (h->status) return;
  const uint32_t wpr = h->wpr, stride = h->npiv_pad, npiv = h->npiv;
  const uint16_t *pivslot = reinterpret_cast<const uint16_t *>(arena + h->off_pivslot);
  uint32_t *wt = reinterpret_cast<uint32_t *>(arena + h->off_wt);
  for (uint32_t e = e0; e < wpr * stride; e += step) {
    const uint32_t w = e / stride, k = e - w * stride;
    wt[e] = k < npiv ? wrows[(size_t)pivslot[k] * wpr + w] : 0u;
  }
}

/* the missing source symbols' LT neighbour lists, translated to slots (uses cptr from the step before) */
template <int Z> SB_HD void pl_final_e(PlanCtx &c, uint32_t tid, uint32_t nt) {
  pl_shared *sh = c.sh; PL_ASSUME_LDS(sh);
  if (sh->status) return;
  const rq_params &p = c.p;
  const uint32_t nl = c.job.nlost;
  const uint32_t *cptr = reinterpret_cast<const uint32_t *>(c.arena + c.part()[2]);
  uint16_t *osl = reinterpret_cast<uint16_t *>(c.arena + c.part()[4]);
  for (uint32_t g = tid; g < nl; g += nt) {
    const uint32_t e = c.lost[g], a = c.b_rptr[p.S + p.H + e], n = c.b_rptr[p.S + p.H + e + 1] - a, o = cptr[g];
    constexpr uint32_t CB = 8; /* entries whose two dependent loads are in flight together */
    for (uint32_t k0 = 0; k0 < n; k0 += CB) {
      uint32_t col[CB], sl[CB];
#pragma unroll
      for (uint32_t q = 0; q < CB; q++) col[q] = k0 + q < n ? c.b_cidx[a + k0 + q] : 0u;
#pragma unroll
      for (uint32_t q = 0; q < CB; q++) sl[q] = c.colslot[col[q]];
#pragma unroll
      for (uint32_t q = 0; q < CB; q++)
        if (k0 + q < n) osl[o + k0 + q] = (uint16_t)sl[q];
    }
  }
}

#endif /* NRQ_PLANNER_BODY_H */
