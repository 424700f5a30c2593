/*
 * nrq_device.hip -- gfx950 kernels and the C ABI of include/nanorq_hip.h.
 *
 * Kernels
 *   nrq_solve_kernel<WB,NT,WV>  the whole data stage of the precode solve: persistent workgroups, one WB-byte column
 *                            strip of one source block at a time, LDS-resident (phases in solve_body.h)
 *   nrq_plan_kernel          the symbolic stage of a decode block (phases in planner_body.h, order in planner_seq.h)
 *   nrq_gen_kernel           LT symbol generation from intermediate symbols in HBM
 *                            (reference decode_row, nanorq.c:184-204)
 * Host side: context, per-K' caches, plan staging, batching, launch geometry.
 * MI355X only: wave64, 160 KiB LDS per workgroup, XCD-aware placement of the work.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nanorq_hip.h"
#include "plan.h"
#include "planner_host.h"
#include "rq_math.h"
#include "solve_body.h"
#include "planner_body.h"
static_assert(RQ_LT_COLS_MAX_REAL <= NRQ_LT_LIST_MAX, "solve_body.h sizes the slack behind out_slots[] for the longest LT list");

#define NRQ_LDS_MAX 163840u /* 160 KiB per workgroup on gfx950 */
/* ... handed out in pieces of 320 dwords (LLVM getLdsDwGranularity for the 160 KiB parts): what a workgroup asks for is rounded
 * up to that, and how many workgroups share a CU follows from the rounded size.  (The HIP occupancy query divides the bytes:
 * it said nine 18 016-byte workgroups fit, eight were resident, and the ninth of every CU ran as a second round -- K=500.) */
#define NRQ_LDS_GRANULE 1280u
static inline uint32_t lds_alloc(uint32_t bytes) { return (bytes + NRQ_LDS_GRANULE - 1u) / NRQ_LDS_GRANULE * NRQ_LDS_GRANULE; }
#ifndef NRQ_WG
#define NRQ_WG 768 /* threads of the solve workgroup: 3 waves per SIMD.  One workgroup owns the CU (LDS), and its phases are
                    * bound by instruction issue and LDS latency: measured 256 -> 512 -> 768 -> 1024 threads = 673 / 796 / 835 /
                    * 807 Gbit/s on the headline workload */
#endif
#define NRQ_GEN_WG 256

/* ============================================================================================
 * Kernels
 * ========================================================================================== */

/* Work: "line groups" -- the 128/WB strips of one block that share a 128-byte line of every symbol row.  Slot q
 * of the work list -> (block, group); workgroup g takes the slots g, g + gridDim.x, ...  With many blocks in the
 * launch (by_block) all groups of a block go to workgroups of one XCD (workgroup g runs on XCD g % 8, observed;
 * speed only), so that the block's plan -- every strip walks the whole op stream -- is served by one L2. */
static inline bool nrq_map_by_block(uint32_t nblk) { return nblk >= 64u || (nblk >= 8u && (nblk & 7u) == 0u); }
__device__ __forceinline__ bool nrq_map_group(uint32_t q, uint32_t nblk, uint32_t gpb, bool by_block, uint32_t *blk, uint32_t *grp) {
  if (by_block) {
    const uint32_t m = q >> 3;
    *blk = (m / gpb) * 8u + (q & 7u);
    *grp = m % gpb;
  } else {
    *blk = q / gpb;
    *grp = q % gpb;
  }
  return *blk < nblk;
}

/* first slot >= q (stepping by gridDim.x) that holds a group of a solvable block; >= nslots if none */
/* first slot >= q (stepping by gridDim.x) that holds a group of a solvable block; >= nslots if none */
__device__ __forceinline__ uint32_t nrq_next_group(uint32_t q, uint32_t nslots, const nrq_job *__restrict__ jobs, uint32_t nblk,
                                                   uint32_t gpb, bool by_block) {
  for (; q < nslots; q += gridDim.x) {
    uint32_t blk, grp;
    if (!nrq_map_group(q, nblk, gpb, by_block, &blk, &grp)) continue;
    const nrq_plan_hdr *h = reinterpret_cast<const nrq_plan_hdr *>(jobs[blk].plan);
    if (h->status == 0) return q; /* rank deficient blocks: nothing is written for them */
  }
  return nslots;
}

/* The data stage: persistent workgroups, one line group at a time, its strips one after the other (solve_body.h:
 * load -> forward passes -> HDPC -> dense stage -> back-substitution -> store).  While wave 0 runs the forward
 * passes of a strip, a few of the other waves gather a portion of the NEXT line group into the input staging
 * buffers and scatter a portion of the PREVIOUS group's results from the output staging buffers to their rows. */
#ifndef NRQ_HDPC_NT
#define NRQ_HDPC_NT 512 /* threads of the HDPC phase; measured: 256 / 512 / 768 -> 34 k / 28 k / 29 k clocks (the closing fold is per thread) */
#endif
#define NRQ_HDPC_NT_ ((uint32_t)NRQ_HDPC_NT)
#ifndef NRQ_GATHER_WAVES_NARROW
#define NRQ_GATHER_WAVES_NARROW 2u /* gather waves of the 768-thread workgroup on strips of at most NRQ_GATHER_WAVES_NARROW_WB bytes */
#endif
#ifndef NRQ_GATHER_WAVES_NARROW_WB
#define NRQ_GATHER_WAVES_NARROW_WB 2
#endif
#ifndef NRQ_MOVER_WAVES
#define NRQ_MOVER_WAVES 2u
#endif
#ifndef NRQ_RING_5W
#define NRQ_RING_5W 36u /* five waves per SIMD: 102 registers per thread */
#endif
#ifndef NRQ_BIG_RING
#define NRQ_BIG_RING NRQ_RING_MAX /* op-word ring (rows) of the two forward waves of the 768-thread workgroup */
#endif
/* NT threads per workgroup: NRQ_WG when one strip image owns the CU's LDS (big blocks), 256 when several fit (small
 * blocks: more workgroups per CU beat more waves per workgroup, each has its own single-wave forward pass).
 * `lsub`: log2 of the strips per work slot -- a whole line (128/WB strips) unless that leaves CUs without work. */
/* WV: waves per SIMD the kernel is compiled for, i.e. its register budget (512 / WV).  The 768-thread variant has
 * the CU to itself (3 waves per SIMD, 168 registers).  The 256-thread variant puts one wave on every SIMD, so WV is
 * also the number of workgroups a CU holds: 4 (128 registers) or, when the LDS images are small enough for 5
 * workgroups, 5 (96 registers, a few more spills).  Left to itself the compiler took 207 registers: 2 workgroups. */
/* G > 1: WIDE strips -- an element is G adjacent 16-byte columns, one lane each (solve_body.h): thread t is lane t % G of
 * virtual thread t / G, and every phase runs on the NT / G virtual threads.  For small blocks. */
/* AL: every symbol row of the launch is aligned to the strip width and T is a multiple of it (the host checks): the movers are
 * then compiled WITHOUT their byte-wise forms -- with both forms at every call site the kernels were half as large again
 * (persistent workgroups in different phases share the instruction cache), and a byte-wise path inside a mover loop makes the
 * compiler wait for all loads in flight where the paths join. */
#ifndef NRQ_TINY_WV
#define NRQ_TINY_WV 3   /* the single-wave variant: waves per SIMD it is built for (170 registers), */
#endif
#ifndef NRQ_TINY_OCC
#define NRQ_TINY_OCC 12u /* and workgroups per compute unit it runs with.  Round 4, K=100 T=1024 x 8192 blocks: 5 / 18 (96 registers,
                          * 151 of them spilled: every phase reloads its pointers from scratch) 370 Gbit/s, 4 / 16 ~385, 3 / 12 419,
                          * 2 / 8 358; K=256 527 / 547 / 569 / 533 */
#endif
#ifndef NRQ_PROF_DONE
#define NRQ_PROF_DONE 9 /* NRQ_PROF samples this strip (counted from 0) of every 16th workgroup.  It must lie behind the workgroup's FIRST work
                         * slot to see a scatter at all (the first slot has no results of a slot before it to move): strip 9 is in the second
                         * slot when slots are 8 strips (16- and 12-byte strips), in the first when they are 16 or more (8 bytes and narrower) */
#endif
#ifndef NRQ_W12_FW
#define NRQ_W12_FW 3 /* forward waves of the 12-byte strip: 3 (a dword each) or 2 (two dwords, one dword) */
#endif
#ifndef NRQ_W12_GW
#define NRQ_W12_GW 2 /* gather / scatter waves beside the three forward waves of the 12-byte strip */
#endif
#ifndef NRQ_W12_SW
#define NRQ_W12_SW 2
#endif
#ifndef NRQ_SMALL_WV
#define NRQ_SMALL_WV 4   /* the 256-thread variant: workgroups per compute unit = waves per SIMD it is built for */
#endif
template <int WB, int NT, int WV, int G = 1, bool AL = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WV)))
void nrq_solve_kernel(const nrq_job *__restrict__ jobs, uint32_t nblk,
                                                       uint32_t T, uint32_t nstrips, uint32_t by_block, uint32_t nslots,
                                                       uint32_t lsub, const uint8_t *__restrict__ kc,
                                                       uint8_t *__restrict__ stage_all, uint32_t stage_stride,
                                                       uint32_t ostage_stride, unsigned long long *__restrict__ prof,
                                                       uint8_t *__restrict__ ybuf, size_t ybuf_stride) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t tid = threadIdx.x;
  /* NFW waves run the forward passes (two, half the strip width each, when the strip is wide enough and the workgroup
   * big enough to spare a second SIMD); the waves on the other SIMDs move data meanwhile: NGW gather, NSW scatter */
  static_assert(G == 1 || WB == 16, "wide strips are made of 16-byte lanes");
  static_assert(WB != 12 || NT >= 512, "12-byte strips: the 768-thread workgroup only (blocks whose 16-byte image does not fit the LDS)");
  constexpr uint32_t WBE = (uint32_t)WB * G; /* bytes of a strip */
  constexpr bool ALX = AL && G == 1 && WB >= 4;
  const uint32_t vt = tid / G, subl = tid % G; /* virtual thread, lane inside it */
  constexpr uint32_t VNT = NT / G;
  constexpr uint32_t SPL = nrq_group_strips(WBE), NFW = (G == 1 && WB == 12 && NT >= 512) ? (uint32_t)NRQ_W12_FW : (G == 1 && WB >= 8 && NT >= 512) ? 2u : 1u,
                     NMV = (NT / 64u) / 4u * (4u - NFW) + ((NT / 64u) % 4u > NFW ? (NT / 64u) % 4u - NFW : 0u),
                     /* two gather and two scatter waves in the big workgroup, not three: every mover wave that keeps loads in
                      * flight slows the forward waves' op words down (measured, headline encode: 3+3 -> row pipeline 89 k clocks,
                      * gather done at 70 k; 2+2 -> 79 k / 83 k; 1+1 -> 72 k / 125 k) */
                     /* (2-byte strips: the GATHER, not the forward wave, ends the window -- with work slots of 8 strips a row piece is
                      * 16 bytes of a 128-byte line and the two gather waves are bound by their requests in flight; round 6, NRQ_PROF
                      * marks at K'=56403: forward wave done 77 k clocks before the window's end, gather at its end) */
                     /* (12-byte strips: three forward waves leave one SIMD free of them, three waves -- two gathering and one scattering
                      * ended the forward window at 1.5 x (encode) and 2.7 x (decode) the forward waves' own time at K=10000; so the
                      * movers are NRQ_W12_GW + NRQ_W12_SW of ALL nine other waves, the free SIMD's first) */
                     NGW = NFW == 3u ? (uint32_t)NRQ_W12_GW : NMV >= 6u ? (WB <= NRQ_GATHER_WAVES_NARROW_WB ? NRQ_GATHER_WAVES_NARROW : NRQ_MOVER_WAVES) : NMV >= 3u ? 2u : 1u,
                     NSW = NFW == 3u ? (uint32_t)NRQ_W12_SW : NMV >= 6u ? NRQ_MOVER_WAVES : NMV - NGW;
  static_assert(NMV >= 2u || NT == 64, "workgroup too small for the data movers");
  /* op-word ring of the forward wave(s), in rows: what the variant's register budget holds without spilling */
#ifndef NRQ_PIPE_SMALL
#define NRQ_PIPE_SMALL 0 /* (round 3: with the movers' aligned-only form the software-pipelined loops of the 256-thread variant keep so many
                          * requests in flight that its forward wave waits for its op words: K=1000 9.7 / 8.6 ms with them, 8.2 / 7.7 without;
                          * K=500 14.6 / 13.4 vs 12.4 / 11.8; K=2000 8.5 / 7.3 vs 7.2 / 6.5) */
#endif
  /* software-pipelined data movers in the 768-thread workgroup (168 registers per thread); the 256-thread variant had them in
   * round 2 (K=1000 977 -> 1031 Gbit/s with the loops as they were then), see NRQ_PIPE_SMALL */
#ifndef NRQ_PIPE_BIG
#define NRQ_PIPE_BIG 1
#endif
  constexpr bool MPIPE = (NRQ_PIPE_BIG && NT >= 512) || (NRQ_PIPE_SMALL && NT == 256 && WV == 4);
#ifndef NRQ_RING_4W
#define NRQ_RING_4W NRQ_RING
#endif
#ifndef NRQ_RING_TINY
#define NRQ_RING_TINY 24u /* the single-wave variant: a dozen workgroups per CU hide each other's op-word latency, and a short ring is less code for
                           * them to share the instruction cache with (K=100 T=1024 / K=256, ring 60 / 36 / 24 / 12 rows: 460 / 468 / 471 / 473 and
                           * 815 / 833 / 835 / 836 Gbit/s; the 256-thread variant loses with a shorter ring: K=1000 1156 / 1093 / 1087 / 1057) */
#endif
  constexpr uint32_t RU = NT >= 512 ? NRQ_BIG_RING : NT == 64 ? NRQ_RING_TINY : WV >= 5 ? NRQ_RING_5W : NRQ_RING_4W;
  /* NT == 64: ONE wave solves the strip on its own (no mover waves: it gathers and scatters its portions itself after
   * the forward passes; barriers are free).  For images of a few KB -- K up to ~400 -- where a strip is a chain of
   * short phases with little parallel work: 19-20 such workgroups share a CU instead of five 256-thread ones, i.e.
   * four times as many strips are in flight to hide the phases' latencies. */
  const uint32_t sub = 1u << lsub;               /* strips per slot */
  const uint32_t gpb = (nstrips + sub - 1u) / sub; /* slots per block */
  /* per workgroup: two sets of SPL input staging buffers (the group being solved, the group being gathered) and two
   * sets of SPL output staging buffers (the group being solved, the group being scattered) */
  const size_t wg_bytes = 2u * SPL * ((size_t)stage_stride + ostage_stride);
  (void)SPL;
  NRQ_GAS uint8_t *stage0 = gptr_w<uint8_t>((uint64_t)(uintptr_t)(stage_all + (size_t)blockIdx.x * wg_bytes));
  NRQ_GAS uint8_t *ostage0 = stage0 + 2u * SPL * (size_t)stage_stride;
  auto group_src = [&](uint32_t q, GroupSrc<WB> &g, uint32_t *blk_out) {
    uint32_t blk, grp;
    nrq_map_group(q, nblk, gpb, by_block != 0u, &blk, &grp);
    const nrq_job *j = jobs + blk;
    g.rowsrc = nrq_uniform_ptr(gptr<uint32_t>(j->rowsrc)); g.src = nrq_uniform_ptr(gptr<uint8_t>(j->src)); g.rep = nrq_uniform_ptr(gptr<uint8_t>(j->rep));
    g.M = nrq_uniform(reinterpret_cast<const nrq_plan_hdr *>(j->plan)->M);
    g.T = nrq_uniform(T); g.strip0 = nrq_uniform(grp * sub); g.nstrips = nrq_uniform(nstrips); g.lsub = nrq_uniform(lsub);
    *blk_out = nrq_uniform(blk);
  };
  auto group_dst = [&](uint32_t q, GroupDst<WB> &g) -> uint32_t { /* returns the staged elements per strip */
    uint32_t blk, grp;
    nrq_map_group(q, nblk, gpb, by_block != 0u, &blk, &grp);
    const nrq_job *j = jobs + blk;
    g.inter = gptr_w<uint8_t>(j->inter); g.out = gptr_w<uint8_t>(j->out); g.orow = gptr<uint32_t>(j->out_row);
    g.ni = j->inter ? reinterpret_cast<const nrq_plan_hdr *>(j->plan)->L : 0u;
    g.nout = j->nout; g.T = nrq_uniform(T); g.strip0 = nrq_uniform(grp * sub); g.nstrips = nrq_uniform(nstrips); g.lsub = nrq_uniform(lsub);
    g.orow = nrq_uniform_ptr(g.orow);
    if (ybuf) { /* split solve: the slot image and C_u go to rows [0, M + u) of the block's work buffer */
      const nrq_plan_hdr *h = reinterpret_cast<const nrq_plan_hdr *>(j->plan);
      g.inter = gptr_w<uint8_t>((uint64_t)(uintptr_t)(ybuf + (size_t)blk * ybuf_stride));
      g.ni = h->M + h->u;
      g.nout = 0u;
    }
    g.inter = nrq_uniform_ptr(g.inter); g.out = nrq_uniform_ptr(g.out);
    g.ni = nrq_uniform(g.ni); g.nout = nrq_uniform(g.nout);
    return g.ni + g.nout;
  };
  uint32_t q = nrq_next_group(blockIdx.x, nslots, jobs, nblk, gpb, by_block != 0u);
  if (q >= nslots) return;
  uint32_t buf = 0, done = 0, qp = nslots; /* qp: the group whose results wait in the other output set */
  {
    GroupSrc<WB> g0;
    uint32_t b0;
    group_src(q, g0, &b0);
    pf_gather_impl<WB, G, MPIPE, ALX>(g0, stage0, stage_stride, 0u, g0.M << lsub, (tid) / G, (NT) / G, subl); /* the first group: nothing to overlap it with */
    __syncthreads();
  }
  while (q < nslots) {
    const uint32_t qn = nrq_next_group(q + gridDim.x, nslots, jobs, nblk, gpb, by_block != 0u);
    GroupSrc<WB> gn;
    GroupDst<WB> gp;
    uint32_t blk, blkn = 0, units_n = 0, units_p = 0;
    if (qn < nslots) { group_src(qn, gn, &blkn); units_n = gn.M << lsub; }
    if (qp < nslots) units_p = group_dst(qp, gp) << lsub;
    {
      GroupSrc<WB> gc;
      group_src(q, gc, &blk);
    }
    NRQ_GAS uint8_t *stage_cur = stage0 + (size_t)buf * SPL * stage_stride, *stage_nxt = stage0 + (size_t)(buf ^ 1u) * SPL * stage_stride;
    NRQ_GAS uint8_t *ostage_cur = ostage0 + (size_t)buf * SPL * ostage_stride, *ostage_prv = ostage0 + (size_t)(buf ^ 1u) * SPL * ostage_stride;
    const uint32_t strip0 = ((by_block ? (q >> 3) : q) % gpb) * sub;
    for (uint32_t sidx = 0; sidx < sub; sidx++) {
      const uint32_t u0 = (uint32_t)(((uint64_t)units_n * sidx) >> lsub), u1 = (uint32_t)(((uint64_t)units_n * (sidx + 1u)) >> lsub);
      const uint32_t s0 = (uint32_t)(((uint64_t)units_p * sidx) >> lsub), s1 = (uint32_t)(((uint64_t)units_p * (sidx + 1u)) >> lsub);
      /* part of the scatter portion is left to the waves that the HDPC phase does not use (when there are any) */
#ifndef NRQ_SCATTER_LATE_PCT
#define NRQ_SCATTER_LATE_PCT 20u
#endif
      /* ... and part of the gather portion: the gather of an encode strip is bound by its bytes in flight against the memory
       * latency and ends with the forward waves (more of it in flight there would delay their op words); in the HDPC window
       * nobody waits for op words.  K=8192 T=1280, encode / decode solve kernel in ms at gather % / scatter %: 0/35 7.09 / 6.28,
       * 40/20 6.61 / 6.30, 35/35 6.79 / 6.25, 50/25 6.82 / 6.37, 60/35 7.02 / 6.34 (beyond 40 % the HDPC window grows by more
       * than the forward window shrinks) */
#ifndef NRQ_GATHER_LATE_PCT
#define NRQ_GATHER_LATE_PCT 40u
#endif
#ifndef NRQ_SCATTER_LATE_PCT_NARROW
#define NRQ_SCATTER_LATE_PCT_NARROW 10u
#endif
#ifndef NRQ_GATHER_LATE_PCT_NARROW
#define NRQ_GATHER_LATE_PCT_NARROW 15u
#endif
#ifndef NRQ_W12_SLATE
#define NRQ_W12_SLATE NRQ_SCATTER_LATE_PCT
#endif
#ifndef NRQ_W12_GLATE
#define NRQ_W12_GLATE NRQ_GATHER_LATE_PCT
#endif
      constexpr uint32_t SLATE = WB == 12 ? NRQ_W12_SLATE : WB >= 8 ? NRQ_SCATTER_LATE_PCT : NRQ_SCATTER_LATE_PCT_NARROW,
                         GLATE = WB == 12 ? NRQ_W12_GLATE : WB >= 8 ? NRQ_GATHER_LATE_PCT : NRQ_GATHER_LATE_PCT_NARROW;
      const uint32_t sm = NT > NRQ_HDPC_NT_ ? s1 - (uint32_t)((uint64_t)(s1 - s0) * SLATE / 100u) : s1;
      const uint32_t um = NT > NRQ_HDPC_NT_ ? u1 - (uint32_t)((uint64_t)(u1 - u0) * GLATE / 100u) : u1;
      const uint32_t strip = strip0 + sidx;
      if (strip >= nstrips) { /* no such strip: everybody moves this portion */
        if (u1 > u0) pf_gather_impl<WB, G, MPIPE, ALX>(gn, stage_nxt, stage_stride, u0, u1, (tid) / G, (NT) / G, subl);
        if (s1 > s0) pf_scatter_impl<WB, G, MPIPE, ALX>(gp, ostage_prv, ostage_stride, s0, s1, (tid) / G, (NT) / G, subl);
        continue;
      }
      StripCtx<WB, G> c;
      c.job = jobs + blk;
      c.plan = reinterpret_cast<const uint8_t *>(c.job->plan);
      c.h = reinterpret_cast<const nrq_plan_hdr *>(c.plan);
      c.kc = kc;
      c.lds = smem + subl * WB;
      c.lay = nrq_lds_plan(c.h, WBE);
      c.T = T;
      c.strip = strip;
      const uint32_t at = strip * WBE + subl * WB, rem = at < T ? T - at : 0u;
      c.valid = rem < (uint32_t)WB ? rem : (uint32_t)WB;
      const uint32_t lpr_ = c.h->lpr; /* (fetched here, with the header fields the first phases need: not a trip of its own later) */
      /* NRQ_PROF=1 debugging aid: shader-clock stamps at phase boundaries, the 10th strip of every 16th workgroup */
      unsigned long long *stamp = nullptr;
      const bool sampled = prof && (blockIdx.x & 15u) == 0 && done == (uint32_t)NRQ_PROF_DONE;
      if (sampled && tid == 0) stamp = prof + (size_t)(blockIdx.x >> 4) * 16;
#define NRQ_STAMP(i) do { if (stamp) stamp[i] = (unsigned long long)clock64(); } while (0)
      /* -DNRQ_STOP_AFTER=p (tools/phase_counters.sh): a strip ends behind phase p -- 0 load, 1 forward, 2 HDPC, 3 GF(2) combinations,
       * 4 dense, 5 tables, 6 back-substitution (7 = the whole strip) -- so that the hardware counters of the variants p and p - 1
       * differ by what phase p costs (phases are separated by workgroup barriers).  Results are garbage; nothing reads them. */
#ifdef NRQ_STOP_AFTER
#define NRQ_STOP(p) if ((NRQ_STOP_AFTER) == (p)) { __syncthreads(); continue; }
#else
#define NRQ_STOP(p)
#endif
      c.dbg = sampled ? prof + (size_t)(blockIdx.x >> 4) * 16 + 9 : nullptr;
      c.dbg_t0 = tid == 0;
      done++;
      NRQ_STAMP(0);
      pf_commit<WB, G, (NT == 64 ? 4 : 8)>(c, stage_cur + (size_t)sidx * stage_stride + subl * WB, 0u, vt, VNT);
      ph_clear<WB, G>(c, vt, VNT);
      __syncthreads();
      NRQ_STAMP(1);
      NRQ_STOP(0)

      /* forward passes (plan.h): wave 0 walks the op stream alone (fwd_rows).  A few waves move data meanwhile -- few,
       * because the forward passes leave them plenty of time and a deep queue of their requests in the CU's memory
       * pipeline would delay the op words wave 0 is waiting for */
      const uint32_t wv = tid >> 6;
      if constexpr (NT == 64) {
        if constexpr (G > 1) fwd_rows_wide<G>(c.template arr<uint32_t>(c.h->off_ops), c.h->nrows, tid);
        else fwd_rows<WB, RU>(c.template arr<uint32_t>(c.h->off_ops), c.h->nrows, tid);
        if (u1 > u0) pf_gather_impl<WB, G, MPIPE, ALX>(gn, stage_nxt, stage_stride, u0, u1, (tid) / G, (64u) / G, subl);
        if (sm > s0) pf_scatter_impl<WB, G, MPIPE, ALX>(gp, ostage_prv, ostage_stride, s0, sm, (tid) / G, (64u) / G, subl);
      } else if (wv < NFW) {
        __builtin_amdgcn_s_setprio(3); /* the critical waves: ahead of the others at instruction issue */
        const NRQ_GAS uint32_t *ops_ = c.template arr<uint32_t>(c.h->off_ops);
        if constexpr (G > 1) {
          fwd_rows_wide<G>(ops_, c.h->nrows, tid);
        } else if constexpr (NFW == 3u) { /* 12-byte strips: a dword of the strip width each */
          if (wv == 0u) fwd_rows_third<0, RU>(ops_, c.h->nrows, tid);
          else if (wv == 1u) fwd_rows_third<4, RU>(ops_, c.h->nrows, tid & 63u);
          else fwd_rows_third<8, RU>(ops_, c.h->nrows, tid & 63u);
        } else if constexpr (NFW == 2u && WB == 12) { /* 12-byte strips on two waves: dwords 0 and 1, dword 2 */
          if (wv == 0u) fwd_rows_two_thirds<RU>(ops_, c.h->nrows, tid);
          else fwd_rows_third<8, RU>(ops_, c.h->nrows, tid & 63u);
        } else if constexpr (NFW == 2u) { /* one half of the strip width each */
          /* (the big workgroup has the registers for the deep ring: 168 per thread) */
          if (wv == 0u) fwd_rows_half<WB, 0, RU>(ops_, c.h->nrows, tid);
          else fwd_rows_half<WB, WB / 2, RU>(ops_, c.h->nrows, tid & 63u);
        } else {
          fwd_rows<WB, RU>(ops_, c.h->nrows, tid);
        }
        __builtin_amdgcn_s_setprio(0);
        NRQ_MARK(c, 1);
      } else if ((wv & 3u) >= NFW || NFW == 3u) { /* the waves that do not share a SIMD with the forward waves; index among them: */
        const uint32_t mv = NFW == 3u ? ((wv & 3u) == 3u ? (wv >> 2) : (wv >> 2) * 3u + (wv & 3u)) /* (SIMD 3's: 0-2; waves 4-6, 8-10: 3-8) */
                                      : (wv >> 2) * (4u - NFW) + (wv & 3u) - NFW;
        if (mv < NGW) {
#if !defined(NRQ_EXPERIMENT_NO_MOVERS) && !defined(NRQ_EXPERIMENT_NO_GATHER) /* (measurement only: what the forward window costs with no mover traffic beside it; results are garbage) */
          if (um > u0) pf_gather_impl<WB, G, MPIPE, ALX>(gn, stage_nxt, stage_stride, u0, um, (mv * 64u + (tid & 63u)) / G, (NGW * 64u) / G, subl);
#endif
          NRQ_MARK_MAX(c, 2);
        } else if (mv < NGW + NSW) {
#if !defined(NRQ_EXPERIMENT_NO_MOVERS) && !defined(NRQ_EXPERIMENT_NO_SCATTER)
          if (sm > s0) pf_scatter_impl<WB, G, MPIPE, ALX>(gp, ostage_prv, ostage_stride, s0, sm, ((mv - NGW) * 64u + (tid & 63u)) / G, (NSW * 64u) / G, subl);
#endif
          NRQ_MARK_MAX(c, 3);
        }
      }
      __syncthreads();
      NRQ_STAMP(2);
      NRQ_STOP(1)

      {
        /* (the 256-thread variant: NRQ_HDPC_NT_SMALL threads -- every thread of the phase ends in a closing fold of 8 x H masked XORs,
         * ~800 instructions whatever its chunk holds, and four such workgroups share a CU's issue slots) */
#ifndef NRQ_HDPC_NT_SMALL
#define NRQ_HDPC_NT_SMALL 256
#endif
        constexpr uint32_t HNT = NT == 256 ? (uint32_t)NRQ_HDPC_NT_SMALL : NT < NRQ_HDPC_NT_ ? NT : NRQ_HDPC_NT_;
        /* (narrow strips: the H sums in registers -- on 16-byte strips that form is slower, 24 k against 17.5 k clocks for the
         * recurrence: 4 x H operations per column instead of four LDS atomics) */
#ifndef NRQ_HDPC_REGS_MAX_WB
#define NRQ_HDPC_REGS_MAX_WB 4
#endif
#ifndef NRQ_HDPC_REGS_12
#define NRQ_HDPC_REGS_12 0
#endif
        if (tid < HNT) { ph_hdpc<WB, G, (NT >= 512 && G == 1 && (WB <= NRQ_HDPC_REGS_MAX_WB || (WB == 12 && NRQ_HDPC_REGS_12)))>(c, tid / G, HNT / G); }
        else { /* the waves HDPC leaves idle */
          if (u1 > um) pf_gather_impl<WB, G, MPIPE, ALX>(gn, stage_nxt, stage_stride, um, u1, (tid - HNT) / G, (NT - HNT) / G, subl);
          if (s1 > sm) pf_scatter_impl<WB, G, MPIPE, ALX>(gp, ostage_prv, ostage_stride, sm, s1, (tid - HNT) / G, (NT - HNT) / G, subl);
        }
      }
      __syncthreads();
      ph_hdpc_reduce<WB, G>(c, vt, VNT);
      __syncthreads();
      NRQ_STAMP(3);
      NRQ_STOP(2)
      /* the GF(2) combinations E_p of the dense stage: tables over the leftover rows in region X (zero again after the
       * reduce above), as many words of the bit rows at a time as it holds */
      for (uint32_t w0 = 0; w0 < lpr_; w0 += low_table_words<WB, G>(c)) {
        uint32_t cb_[NRQ_COMBINE_WU];
        ph_low_tables<WB, G>(c, w0, vt, VNT);
        ph_combine_fetch<WB, G>(c, w0, vt, VNT, cb_);
        __syncthreads();
        ph_combine<WB, G>(c, w0, vt, VNT, cb_);
        __syncthreads();
      }
      if (lpr_) { /* (0: the combinations were ops of the stream -- small blocks) */
        ph_clear_x<WB, G>(c, vt, VNT);
        __syncthreads();
      }
      NRQ_STAMP(4);
      NRQ_STOP(3)
#ifndef NRQ_FOLD_PRE256
#define NRQ_FOLD_PRE256 0 /* coefficients the 256-thread variant's dense fold asks for at once (0: one per term, a trip to L2 each) */
#endif
      ph_dense_fold<WB, G, (NT == 64 ? 8 : NT == 256 ? NRQ_FOLD_PRE256 : 0)>(c, vt, VNT);
      __syncthreads();
      NRQ_MARK(c, 4);
      if (dense_fold_shared(VNT) || G > 1) { /* (then the fold leaves its products in the accumulator copies) */
        ph_hdpc_reduce<WB, G>(c, vt, VNT);
        __syncthreads();
      }
      NRQ_MARK(c, 5);
      ph_dense_free<WB, G, (NT == 64)>(c, vt, VNT);
      __syncthreads();
      NRQ_MARK(c, 6);
      ph_dense_cu<WB, G, (NT == 64)>(c, vt, VNT);
      __syncthreads();
      NRQ_STAMP(5);
      NRQ_STOP(4)
      if (ybuf) { /* split solve (narrow strips): back-substitution and results are nrq_backsub_kernel / nrq_collect_kernel */
        NRQ_STAMP(6);
        NRQ_STAMP(7);
        ph_store_raw<WB, G>(c, ostage_cur + (size_t)sidx * ostage_stride + subl * WB, vt, VNT);
      } else {
        ph_tables<WB, G>(c, vt, VNT);
        __syncthreads();
        NRQ_STAMP(6);
        NRQ_STOP(5)
        ph_backsub<WB, G, (NT >= 512)>(c, vt, VNT);
        ph_park<WB, G>(c, vt, VNT);
        __syncthreads();
        NRQ_STAMP(7);
        NRQ_STOP(6)
        ph_store<WB, G, (NT >= 512)>(c, ostage_cur + (size_t)sidx * ostage_stride + subl * WB, vt, VNT);
      }
      __syncthreads();
      NRQ_STAMP(8);
#undef NRQ_STAMP
#undef NRQ_STOP
    }
    __syncthreads(); /* the gathered group is complete (and, for what a strip-less portion moved, visible) */
    qp = q;
    q = qn;
    buf ^= 1u;
  }
  /* the results of the last group */
  if (qp < nslots) {
    GroupDst<WB> gp;
    const uint32_t units_p = group_dst(qp, gp) << lsub;
    pf_scatter_impl<WB, G, MPIPE, ALX>(gp, ostage0 + (size_t)(buf ^ 1u) * SPL * ostage_stride, ostage_stride, 0u, units_p, (tid) / G, (NT) / G, subl);
  }
}

enum {
#ifdef NRQ_PROF_INIT /* diagnostic build: the init phases one by one, in the slots big blocks leave empty (Wrun, Wmove, mh) */
  pl_tag_pl_init_a = 0,
  pl_tag_pl_init_b = 2,
  pl_tag_pl_scan_a = 4,
  pl_tag_pl_scan_b = 4,
  pl_tag_pl_scan_c = 4,
  pl_tag_pl_pcsc_fill = 11,
#else
  pl_tag_pl_init_a = 0,
  pl_tag_pl_init_b = 0,
  pl_tag_pl_scan_a = 0,
  pl_tag_pl_scan_b = 0,
  pl_tag_pl_scan_c = 0,
  pl_tag_pl_pcsc_fill = 0,
#endif
  pl_tag_pl_round_claim = 1,
  pl_tag_pl_round_drop = 3,
  pl_tag_pl_round_chain_end = 3,
  pl_tag_pl_inact_find = 5,
  pl_tag_pl_inact_find_b = 5,
  pl_tag_pl_inact_find_c = 5,
  pl_tag_pl_inact_apply_a = 6,
  pl_tag_pl_inact_apply_b = 6,
  pl_tag_pl_inact_next = 6,
  pl_tag_pl_event_scan = 5,
  pl_tag_pl_event_pick = 6,
  pl_tag_pl_event_drop = 6,
  pl_tag_pl_lev_0 = 7,
  pl_tag_pl_pivot_sort = 7,
  pl_tag_pl_lev_a = 7,
  pl_tag_pl_check_a = 7,
  pl_tag_pl_check_b = 7,
  pl_tag_pl_check_c = 7,
  pl_tag_pl_lev_b = 7,
  pl_tag_pl_w_init = 8,
  pl_tag_pl_w_init_b = 8,
  pl_tag_pl_cls_fetch = 8,
  pl_tag_pl_w_group = 8,
  pl_tag_pl_w_stage = 8,
  pl_tag_pl_wfast_spill = 4,
  pl_tag_pl_wfast_load = 4,
  pl_tag_pl_wfast_store = 4,
  pl_tag_pl_wfast_restore = 4,
  pl_tag_pl_low_a = 9,
  pl_tag_pl_low_b = 9,
  pl_tag_pl_low_c = 9,
  pl_tag_pl_ops_layout_a = 10,
  pl_tag_pl_ops_layout = 10,
  pl_tag_pl_ops_clear = 10,
  pl_tag_pl_ops_emit = 10,
  pl_tag_pl_ops_check_a = 10,
  pl_tag_pl_ops_check_b = 10,
  pl_tag_pl_ops_check_c = 10,
  pl_tag_pl_sh_restore = 0,
  pl_tag_pl_sh_save = 15,
  pl_tag_pl_mh_ext_clear = 15,
  pl_tag_pl_mh_fetch = 11,
  pl_tag_pl_mh_init = 11,
  pl_tag_pl_mh_load = 11,
  pl_tag_pl_mh_acc = 11,
  pl_tag_pl_gj_a = 12,
  pl_tag_pl_gjp_init = 12,
  pl_tag_pl_gjp_bid = 12,
  pl_tag_pl_gjp_step = 12,
  pl_tag_pl_gjp_wave = 12,
  pl_tag_pl_mhrev_load = 11,
  pl_tag_pl_mhrev_load_b = 11,
  pl_tag_pl_mhrev_store = 11,
  pl_tag_pl_mhrev_scatter = 11,
  pl_tag_pl_gjp_comb = 12,
  pl_tag_pl_gjp_stage = 12,
  pl_tag_pl_gjp_apply = 12,
  pl_tag_pl_gj_b = 12,
  pl_tag_pl_bin_a = 13,
  pl_tag_pl_bin_b = 13,
  pl_tag_pl_bin_c = 13,
  pl_tag_pl_dense_a = 14,
  pl_tag_pl_dense_b = 14,
  pl_tag_pl_dense_step_a = 14,
  pl_tag_pl_dense_step_b = 14,
  pl_tag_pl_dense_c = 14,
  pl_tag_pl_extra_a = 14,
  pl_tag_pl_extra_b = 14,
  pl_tag_pl_extra_c = 14,
  pl_tag_pl_extra_d = 14,
  pl_tag_pl_final_a = 15,
  pl_tag_pl_final_b = 15,
  pl_tag_pl_final_c = 15,
  pl_tag_pl_final_c2 = 15,
  pl_tag_pl_final_c3 = 15,
  pl_tag_pl_final_d = 15,
  pl_tag_pl_final_e = 15,
  pl_tag_pl_mark_failed = 15,
};

/* The symbolic stage of one decode block per workgroup (phases in planner_body.h, order in
 * planner_seq.h): reception pattern -> device plan + the block's solve job. */
/* PK = 1: the instance for blocks whose peeling state does not fit the LDS -- it carries the compact form of that state
 * (planner_body.h "compact peeling state"); the others are compiled without it */
/* (the instances for small blocks are built for five waves per SIMD, 96 registers: at the ~100 the compiler takes by itself a
 * SIMD holds four, and what bounds their planner is the number of blocks in flight per CU) */
template <int NT, int PK = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT <= 256 ? 5 : 1)))
void nrq_plan_kernel(rq_params p, const uint8_t *__restrict__ kc,
                                                         const nrq_planjob *__restrict__ pjobs,
                                                         nrq_job *__restrict__ jobs_out, uint32_t nblk, uint32_t Mcap,
                                                         uint32_t npcap, uint32_t ucap, uint32_t lds_dyn_bytes,
                                                         unsigned long long *__restrict__ prof, uint32_t seg, uint32_t qcap,
                                                         uint32_t lowcap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
#define PL_SEG_ (PK == 0 ? 0u : seg) /* (blocks whose peeling state fits the LDS run in one part: launch_plan_kernel) */
  if (b >= nblk) return;
  /* the dynamic region comes first (LDS address 0: the W strip image is addressed like the solve kernel's), the
   * fixed workgroup state after it */
  uint8_t *dyn = smem;
  pl_shared *sh = reinterpret_cast<pl_shared *>(smem + lds_dyn_bytes);
  PlanCtx c;
  pl_ctx_setup(c, p, kc, pjobs[b], sh, dyn, lds_dyn_bytes, Mcap, npcap, ucap, jobs_out + b, qcap, lowcap, (uint32_t)NT);
  if (!PK) c.pk_cnt = c.pk_un = c.pk_pa = c.pk_vb = nullptr;
  if (PL_SEG_ == 3u) c.cls_glob = reinterpret_cast<uint32_t *>(c.work + c.wl.cls_g); /* (pl_lev_b zeroes the counters nrq_wentry_kernel counts into) */
  /* NRQ_PROF=1: thread 0 of block 0 accumulates shader clocks per phase family (index = PL_TAG) */
  unsigned long long t_prev = prof ? (unsigned long long)clock64() : 0ull;
#ifdef PL_STAMP
#define PL_ACC(tag) ((void)0) /* (the stamps alone: the per-phase clocks' stores to memory would be what the stamped loads wait for) */
#else
#define PL_ACC(tag) do { if (__builtin_expect(prof && b == 0 && tid == 0, 0)) { unsigned long long t_ = (unsigned long long)clock64(); \
                                                          prof[tag] += t_ - t_prev; prof[16 + tag] += 1; t_prev = t_; } } while (0)
#endif
#define PL_PHASE(fn) do { fn<PK>(c, tid, (uint32_t)NT); __syncthreads(); PL_ACC(pl_tag_##fn); } while (0)
#define PL_PHASE1(fn, a) do { fn<PK>(c, (a), tid, (uint32_t)NT); __syncthreads(); PL_ACC(pl_tag_##fn); } while (0)
#ifndef PL_CLAIM_FULL_BARRIER
#define PL_CLAIM_FULL_BARRIER 0
#endif
/* (the claim phase: with the peeling decisions in LDS its global stores -- pivot lists, the HBM copies of row / column
 * state -- are read after peeling only, or by nobody before the next full barrier: the barrier waits for the LDS alone, not
 * for the stores' way to memory and back, a trip per round) */
#ifndef PL_CLAIM_SKIP_IDLE
#define PL_CLAIM_SKIP_IDLE 1
#endif
#define PL_PHASE1_CLAIM(fn, a) do { if (!PL_CLAIM_SKIP_IDLE || (tid & ~63u) < nq_) fn<PK>(c, (a), tid, (uint32_t)NT); \
    if (!PL_CLAIM_FULL_BARRIER && (pl_peel_in_lds(c) || c.pk_cnt)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads(); \
    PL_ACC(pl_tag_##fn); } while (0)
#define PL_WFAST_RUN(wb) do { \
    if (tid < NRQ_ROW) { \
      const NRQ_GAS uint32_t *ops_ = gptr<uint32_t>(c.arena + c.sh->off_ops); \
      if ((wb) == 16u) fwd_rows<16>(ops_, pl_wfast_rows(c), tid); \
      else if ((wb) == 8u) fwd_rows<8>(ops_, pl_wfast_rows(c), tid); \
      else fwd_rows<4>(ops_, pl_wfast_rows(c), tid); \
    } \
    __syncthreads(); PL_ACC(2); } while (0)
#define PL_MHREV_RUN(wb) do { \
    if (tid < NRQ_ROW) { \
      const NRQ_GAS uint32_t *ops_ = gptr<uint32_t>(c.arena + c.sh->off_ops); \
      if ((wb) == 16u) rev_rows<16>(ops_, pl_wfast_rows(c), tid); \
      else if ((wb) == 8u) rev_rows<8>(ops_, pl_wfast_rows(c), tid); \
      else rev_rows<4>(ops_, pl_wfast_rows(c), tid); \
    } \
    __syncthreads(); PL_ACC(11); } while (0)
#define PL_SEG PL_SEG_ /* (blocks whose peeling state fits the LDS run in one part: launch_plan_kernel) */
#define PL_STEER_SYNC __syncthreads()
#define PL_NT_ ((uint32_t)NT)
#define PL_Z ((uint32_t)PK)
#ifdef PL_STAMP
  c.st_on = prof && b == 0 && tid == 0; c.st_prev = 0; c.st_blk = prof && b == 0;
  if (tid < 32) sh->st_acc[tid] = 0;
  __syncthreads();
#endif
  if (PL_SEG_ == 2u || PL_SEG_ == 4u) PL_PHASE(pl_sh_restore);
#include "planner_seq.h"
#ifdef PL_STAMP
  if (c.st_on) { for (int i_ = 0; i_ < 32; i_++) prof[32 + i_] = sh->st_acc[i_]; }
#endif
  if (PL_SEG_ == 1u || PL_SEG_ == 4u) PL_PHASE(pl_mh_ext_clear);
  if (PL_SEG_ == 1u || PL_SEG_ == 3u || PL_SEG_ == 4u) PL_PHASE(pl_sh_save);
#undef PL_SEG
#undef PL_SEG_
#undef PL_STEER_SYNC
#undef PL_NT_
#undef PL_PHASE
#undef PL_PHASE1
#undef PL_PHASE1_CLAIM
#undef PL_WFAST_RUN
#undef PL_MHREV_RUN
#undef PL_Z
#undef PL_ACC
}

/* Between parts 3 and 4 of a segmented planner run: the entry pass over the constraint matrix (pl_w_init: every entry either
 * toggles a bit of its row's W row or becomes a row op, counted in its level group and lane class and recorded) by `nparts`
 * workgroups per block.  One workgroup is bound by its CU's rate of scattered accesses there -- three per entry, 571 k entries at
 * K'=56403: 4.7 M clocks, a fifth of the plan.  Counters and the record counter are the workspace's for the duration. */
__global__ __launch_bounds__(1024) void nrq_wentry_kernel(rq_params p, const uint8_t *__restrict__ kc, const nrq_planjob *__restrict__ pjobs,
                                                          uint32_t Mcap, uint32_t npcap, uint32_t ucap, uint32_t lds_dyn_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t part = blockIdx.x, nparts = gridDim.x, b = blockIdx.y, tid = threadIdx.x;
  pl_shared *sh = reinterpret_cast<pl_shared *>(smem);
  PlanCtx c;
  /* (lds_dyn_bytes: what the planner workgroup has -- pl_cls_place / pl_col_level decide by it; nothing of it is touched here) */
  pl_ctx_setup(c, p, kc, pjobs[b], sh, nullptr, 0u, Mcap, npcap, ucap, nullptr, PL_QCAP, PL_LOWCAP, PL_NT);
  c.aux_lds = reinterpret_cast<uint8_t *>(smem); c.aux_bytes = lds_dyn_bytes; c.dense_lds = c.aux_lds; c.dense_bytes = lds_dyn_bytes; /* (as the planner sees them: only their sizes matter) */
  pl_sh_restore<0>(c, tid, 1024u);
  __syncthreads();
  if (sh->status != 0 || sh->nV != 0) return;
  c.cls_glob = reinterpret_cast<uint32_t *>(c.work + c.wl.cls_g);
  c.nrec_ptr = &c.wentry[0];
  c.own_ptr = &c.wentry[3];
  pl_w_init_part<0>(c, part, nparts, tid, 1024u);
  __syncthreads();
  pl_wentry_report<0>(c, tid, 1024u);
}

/* Between the two parts of a segmented planner run (planner_seq.h): the HDPC fold over the pivots,
 * MhT[x] = G_U[:,x] ^ SUM_k W[k][x] * G[:, pivcol k], by `nparts` workgroups per block -- each folds every nparts-th tile
 * of 256 pivots into a private MhT in LDS (the planner's own phases pl_mh_load / pl_mh_acc) and XORs it into the copy in
 * the block's workspace.  One workgroup did this alone in 12 M clocks at K'=56403 (18 % of the plan). */
__global__ __launch_bounds__(1024) void nrq_mh_kernel(rq_params p, const uint8_t *__restrict__ kc, const nrq_planjob *__restrict__ pjobs,
                                                      uint32_t Mcap, uint32_t npcap, uint32_t ucap, uint32_t lds_dyn_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t part = blockIdx.x, nparts = gridDim.x, b = blockIdx.y, tid = threadIdx.x;
  pl_shared *sh = reinterpret_cast<pl_shared *>(smem + lds_dyn_bytes);
  PlanCtx c;
  pl_ctx_setup(c, p, kc, pjobs[b], sh, smem, lds_dyn_bytes, Mcap, npcap, ucap, nullptr, PL_QCAP, PL_LOWCAP, PL_NT); /* (as the big planner workgroup) */
  pl_sh_restore<0>(c, tid, 1024u);
  __syncthreads();
  if (sh->status != 0 || sh->nV != 0) return;
  if (part == 0) pl_mh_init<0>(c, tid, 1024u); else pl_mh_part_zero<0>(c, tid, 1024u);
  __syncthreads();
  const uint32_t ntiles = (sh->npiv + PL_MH_TILE - 1u) / PL_MH_TILE;
  for (uint32_t tl = part; tl < ntiles; tl += nparts) {
    pl_mh_load<0>(c, tl, tid, 1024u);
    __syncthreads();
    pl_mh_acc<0>(c, tl, tid, 1024u);
    __syncthreads();
  }
  pl_mh_part_flush<0>(c, tid, 1024u);
}

/* Between the two parts of a segmented planner run: the W pass -- W = X^-1 * A_U and the leftover rows' reduced
 * coefficients are the op stream applied to the bit rows (planner_body.h "W pass, fast path").  Bit columns are
 * independent, so every workgroup takes a 2-byte strip of all W rows into LDS (slot image like the solve kernel's),
 * one wave runs the row pipeline over the stream (fwd_rows<2>), the strip goes back: wpr * 2 workgroups per block
 * instead of a level-by-level pass on the HBM rows by one (12 M clocks at K'=56403). */
__global__ __launch_bounds__(256) void nrq_wpass_kernel(rq_params p, const uint8_t *__restrict__ kc, const nrq_planjob *__restrict__ pjobs,
                                                        uint32_t Mcap, uint32_t npcap, uint32_t ucap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; /* the strip image, from LDS address 0 */
  const uint32_t strip = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const nrq_planjob &j = pjobs[b];
  const nrq_kconst_hdr *kh = reinterpret_cast<const nrq_kconst_hdr *>(kc);
  const pl_work_layout wl = pl_work_plan(p.L, Mcap, npcap, ucap, kh->nnz + npcap * PL_PATCH_STRIDE);
  uint8_t *work = PL_HBM(uint8_t, j.work);
  const pl_shared *sv = reinterpret_cast<const pl_shared *>(work + wl.sh_save); /* part 1's state */
  if (sv->status != 0 || sv->nV != 0 || strip >= sv->wpr * 2u) return;
  const uint32_t M = sv->M, wpr = sv->wpr, nrows = sv->spare_base;
  uint16_t *rows16 = reinterpret_cast<uint16_t *>(work + wl.wrows); /* W row r, halfword h at [r * wpr * 2 + h] */
  uint16_t *img = reinterpret_cast<uint16_t *>(smem);
  for (uint32_t e = tid; e < M + NRQ_SCRATCH; e += 256u)
    img[e] = e >= NRQ_SCRATCH ? rows16[(size_t)(e - NRQ_SCRATCH) * wpr * 2u + strip] : (uint16_t)0;
  __syncthreads();
  if (tid < NRQ_ROW) fwd_rows<2>(gptr<uint32_t>(PL_HBM(uint8_t, j.arena) + sv->off_ops), nrows, tid);
  __syncthreads();
  for (uint32_t r = tid; r < M; r += 256u) rows16[(size_t)r * wpr * 2u + strip] = img[r + NRQ_SCRATCH];
}

/* After a segmented planner run: W transposed by word into the plan (pl_wt_fill), by many workgroups. */
__global__ __launch_bounds__(256) void nrq_wt_kernel(const nrq_planjob *__restrict__ pjobs, uint32_t L, uint32_t Mcap, uint32_t npcap,
                                                     uint32_t ucap, uint32_t nnzcap) {
  const nrq_planjob &j = pjobs[blockIdx.y];
  const pl_work_layout wl = pl_work_plan(L, Mcap, npcap, ucap, nnzcap);
  pl_wt_fill(PL_HBM(uint8_t, j.arena), reinterpret_cast<const uint32_t *>(PL_HBM(uint8_t, j.work) + wl.wrows),
             blockIdx.x * 256u + threadIdx.x, gridDim.x * 256u);
}

/* ---- split solve of big blocks (strips of 2 or 4 bytes) ----
 * An LDS-resident strip pays every per-row cost per WB bytes; at WB = 2 the back-substitution C(pivot k) = Y_k ^ W_k * C_u
 * alone is 55 % of a strip (K'=56403: 160 table lookups of TWO bytes per pivot).  It needs no slot image, only Y_k and
 * the u values C_u -- so for narrow strips the solve kernel stops after the dense stage and hands Y and C_u over as
 * full-width rows of a per-block work buffer, and this kernel finishes on 32-byte strips: 16-entry XOR tables over
 * groups of 4 inactive columns in LDS (80 KB at u <= 640), one pivot per thread, 32 bytes per lookup.
 * Rows [0, M) of the work buffer: slot image (Y at the pivots' slots), rows [M, M + u): C_u.  On return the buffer
 * holds the FINAL slot image: the pivot rows hold their intermediate symbols, the rows uslot[x] the inactive columns'. */
template <int SB>
__global__ __launch_bounds__(256) void nrq_backsub_kernel(const nrq_job *__restrict__ jobs, uint32_t T, uint8_t *__restrict__ ybuf,
                                                          size_t ybuf_stride, uint32_t nchunks) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t tid = threadIdx.x, chunk = blockIdx.y, blk = blockIdx.z;
  /* Workgroup i runs on XCD i % 8 (round-robin dispatch).  The strips that share a 128-byte line of every row -- 128 / SB
   * neighbours -- are given to workgroups i, i + 8, i + 16 ... : same XCD, dispatched together, walking the pivots in step, so
   * a line fetched for one of them is in that XCD's L2 for the others (with strip = i the neighbours sat on different XCDs and
   * every one of them fetched the line from HBM: the kernel is bound by its scattered 32-byte row accesses). */
  uint32_t strip = blockIdx.x;
  {
    constexpr uint32_t NB_ = 128u / SB, PER = 8u * NB_;
    const uint32_t i = blockIdx.x, full = (gridDim.x / PER) * PER;
    if (i < full) {
      const uint32_t r = i % PER, g = (i / PER) * 8u + (r & 7u), j = r >> 3;
      strip = g * NB_ + j;
    }
  }
  const uint8_t *plan = reinterpret_cast<const uint8_t *>(jobs[blk].plan);
  const nrq_plan_hdr *h = reinterpret_cast<const nrq_plan_hdr *>(plan);
  if (h->status) return;
  const uint32_t M = h->M, u = h->u, wpr = h->wpr, npiv = h->npiv, stride = h->npiv_pad;
  const NRQ_GAS uint16_t *pivslot = gptr<uint16_t>(plan + h->off_pivslot);
  const NRQ_GAS uint16_t *uslot = gptr<uint16_t>(plan + h->off_uslot);
  const NRQ_GAS uint32_t *wt = gptr<uint32_t>(plan + h->off_wt);
  NRQ_GAS uint8_t *Y = gptr_w<uint8_t>((uint64_t)(uintptr_t)(ybuf + (size_t)blk * ybuf_stride));
  const NRQ_GAS uint8_t *Cu = Y + (size_t)M * T;
  const uint32_t col0 = strip * SB, rem = T - col0, valid = rem < (uint32_t)SB ? rem : (uint32_t)SB;
  constexpr int NQ = SB / 16;
  auto load = [&](const NRQ_GAS uint8_t *p, SV<16> (&v)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const uint32_t o = (uint32_t)q * 16u;
      v[q] = o < valid ? g_get<16>(p + o, valid - o < 16u ? valid - o : 16u) : sv_zero<16>();
    }
  };
  auto store = [&](NRQ_GAS uint8_t *p, const SV<16> (&v)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const uint32_t o = (uint32_t)q * 16u;
      if (o < valid) g_put<16>(p + o, valid - o < 16u ? valid - o : 16u, v[q]);
    }
  };
  /* tables: entry (grp, nib) = XOR of C_u[4 grp + b] over the bits b of nib */
  const uint32_t ngroups = wpr * 8u;
  for (uint32_t e = tid; e < ngroups * 16u; e += 256u) {
    const uint32_t grp = e >> 4, nib = e & 15u;
    SV<16> acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = sv_zero<16>();
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) {
      const uint32_t x = grp * 4u + b;
      if (((nib >> b) & 1u) && x < u) {
        SV<16> t[NQ];
        load(Cu + (size_t)x * T + col0, t);
#pragma unroll
        for (int q = 0; q < NQ; q++) sv_xor<16>(acc[q], t[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) lds_put<16>(smem, e * NQ + q, acc[q]);
  }
  __syncthreads();
  if (chunk == 0) /* park the inactive columns in the slots the plan reserved for them (rows that are no pivots) */
    for (uint32_t x = tid; x < u; x += 256u) {
      SV<16> t[NQ];
      load(Cu + (size_t)x * T + col0, t);
      store(Y + (size_t)uslot[x] * T + col0, t);
    }
  const uint32_t k0 = (uint32_t)(((uint64_t)npiv * chunk) / nchunks), k1 = (uint32_t)(((uint64_t)npiv * (chunk + 1u)) / nchunks);
  /* A pivot per thread and trip: Y(slot) ^= W_k * C_u through the tables.  Everything a pivot needs from memory -- its slot, its
   * row, its first WCH words of W -- is asked for while the pivot BEFORE it does its lookups (all loads unconditional: a word
   * beyond wpr re-reads the last one and is not used): before, a pivot was a chain of seven trips (slot, row, five groups of four
   * W words: 12 us per pivot at K'=56403, 1.8 ms per launch, 3.3 ms at K=27000 T=65504). */
  constexpr uint32_t WCH = 20u;
  auto words = [&](uint32_t k, uint32_t w0, uint32_t (&b)[WCH]) {
#pragma unroll
    for (uint32_t j = 0; j < WCH; j++) b[j] = wt[(size_t)(w0 + j < wpr ? w0 + j : wpr - 1u) * stride + k];
  };
  uint32_t k = k0 + tid;
  if (k >= k1) return;
  uint32_t bits_n[WCH];
  SV<16> acc_n[NQ];
  NRQ_GAS uint8_t *row_n = Y + (size_t)pivslot[k] * T + col0;
  load(row_n, acc_n);
  words(k, 0u, bits_n);
  for (; k < k1; k += 256u) {
    SV<16> acc[NQ];
    uint32_t bits[WCH];
    NRQ_GAS uint8_t *row = row_n;
#pragma unroll
    for (int z = 0; z < NQ; z++) acc[z] = acc_n[z];
#pragma unroll
    for (uint32_t j = 0; j < WCH; j++) bits[j] = bits_n[j];
    const uint32_t kn = k + 256u;
    if (kn < k1) {
      row_n = Y + (size_t)pivslot[kn] * T + col0;
      load(row_n, acc_n);
      words(kn, 0u, bits_n);
    }
    for (uint32_t w0 = 0;;) {
#pragma unroll
      for (uint32_t j = 0; j < WCH; j++) {
        if (w0 + j >= wpr) break;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
          const uint32_t e = ((w0 + j) * 8u + q) * 16u + ((bits[j] >> (4u * q)) & 15u);
#pragma unroll
          for (int z = 0; z < NQ; z++) sv_xor<16>(acc[z], lds_get<16>(smem, e * NQ + z));
        }
      }
      w0 += WCH;
      if (w0 >= wpr) break;
      words(k, w0, bits); /* (more than WCH words: u > 640) */
    }
    store(row, acc);
  }
}

/* Results of the split solve, from the final slot image in the work buffer: workgroup (e, blk) writes one row --
 * intermediate symbol e = row colslot[e] (if the job wants them), or generated symbol q = XOR of the rows its list
 * names (plan slots of its LT neighbours), to the row of `out` the job assigns. */
__global__ __launch_bounds__(256) void nrq_collect_kernel(const nrq_job *__restrict__ jobs, uint32_t T, const uint8_t *__restrict__ ybuf,
                                                          size_t ybuf_stride) {
  const uint32_t e = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x;
  const nrq_job *j = jobs + blk;
  const uint8_t *plan = reinterpret_cast<const uint8_t *>(j->plan);
  const nrq_plan_hdr *h = reinterpret_cast<const nrq_plan_hdr *>(plan);
  if (h->status) return;
  const uint32_t ni = j->inter ? h->L : 0u;
  if (e >= ni + j->nout) return;
  const NRQ_GAS uint8_t *F = gptr<uint8_t>((uint64_t)(uintptr_t)(ybuf + (size_t)blk * ybuf_stride));
  __shared__ uint32_t rows[RQ_MAX_LT_COLS + 1];
  __shared__ uint32_t nrows;
  NRQ_GAS uint8_t *dst;
  if (e < ni) {
    if (tid == 0) { rows[0] = gptr<uint16_t>(plan + h->off_colslot)[e]; nrows = 1u; }
    dst = gptr_w<uint8_t>(j->inter) + (size_t)e * T;
  } else {
    const uint32_t q = e - ni;
    const NRQ_GAS uint32_t *cptr = gptr<uint32_t>(j->out_cptr);
    const NRQ_GAS uint16_t *osl = gptr<uint16_t>(j->out_slots);
    const uint32_t a = cptr[q], n = cptr[q + 1] - a;
    if (tid < n && tid <= RQ_MAX_LT_COLS) rows[tid] = osl[a + tid];
    if (tid == 0) nrows = n <= RQ_MAX_LT_COLS ? n : RQ_MAX_LT_COLS;
    dst = gptr_w<uint8_t>(j->out) + (size_t)gptr<uint32_t>(j->out_row)[q] * T;
  }
  __syncthreads();
  const uint32_t n = nrows;
  const bool vec = (T & 15u) == 0 && ((reinterpret_cast<uintptr_t>(F) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  if (vec) {
    for (uint32_t off = tid * 16u; off < T; off += 256u * 16u) {
      SV<16> acc = sv_zero<16>();
      for (uint32_t k = 0; k < n; k++) sv_xor<16>(acc, g_get_stream<16>(F + (size_t)rows[k] * T + off, 16u));
      g_put<16>(dst + off, 16u, acc);
    }
  } else {
    for (uint32_t off = tid; off < T; off += 256u) {
      uint8_t acc = 0;
      for (uint32_t k = 0; k < n; k++) acc ^= F[(size_t)rows[k] * T + off];
      dst[off] = acc;
    }
  }
}

/* Symbol ingestion on the device (nrq_scatter_symbols): symbol k of a contiguous packet buffer goes to the row its
 * tag names -- dst[k] is the row's device address (0 = drop).  One workgroup per symbol. */
/* Control data (job records, lists of lost / received ESIs: KBs) from page-locked host memory into a device buffer by a
 * KERNEL that reads the host memory directly: a hipMemcpyAsync of it is served by the host-to-device copy engine in the order
 * of submission, i.e. behind every bulk upload queued before it on ANY stream -- the planner of nanorq_repair_all then
 * started when the last packet of a deferred ingestion had landed (24 ms for 128 blocks of K=8192) instead of at once. */
__global__ __launch_bounds__(256) void nrq_ctl_copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint32_t n16) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}

__global__ __launch_bounds__(128) void nrq_scatter_kernel(const uint8_t *__restrict__ blob, uint32_t T, const uint64_t *__restrict__ dst,
                                                          uint32_t n) {
  const uint32_t k = blockIdx.x;
  if (k >= n) return;
  uint8_t *d = reinterpret_cast<uint8_t *>(dst[k]);
  if (!d) return;
  const uint8_t *s = blob + (size_t)k * T;
  if ((T & 15u) == 0 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
    for (uint32_t off = threadIdx.x * 16u; off < T; off += 128u * 16u)
      *reinterpret_cast<uint4 *>(d + off) = *reinterpret_cast<const uint4 *>(s + off);
  } else {
    for (uint32_t off = threadIdx.x; off < T; off += 128u) d[off] = s[off];
  }
}

/* Rows by address pair: row k (T bytes) from src[k] to dst[k]; either may be page-locked HOST memory (the receiver's repaired
 * symbols leave for their places in the caller's output buffer: no staging buffer, no copy per row).  pairs = src0, dst0, src1, ... */
__global__ __launch_bounds__(128) void nrq_move_rows_kernel(const uint64_t *__restrict__ pairs, uint32_t T, uint32_t n) {
  const uint32_t k = blockIdx.x;
  if (k >= n) return;
  const uint8_t *s = reinterpret_cast<const uint8_t *>(pairs[2u * k]);
  uint8_t *d = reinterpret_cast<uint8_t *>(pairs[2u * k + 1u]);
  if (!s || !d) return;
  if ((T & 15u) == 0 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
    for (uint32_t off = threadIdx.x * 16u; off < T; off += 128u * 16u)
      *reinterpret_cast<uint4 *>(d + off) = *reinterpret_cast<const uint4 *>(s + off);
  } else {
    for (uint32_t off = threadIdx.x; off < T; off += 128u) d[off] = s[off];
  }
}

/* One workgroup per (symbol, block): out = XOR of the LT neighbours of `isi` among the L
 * intermediate symbols in HBM (coalesced 16-byte lanes along the symbol). */
__global__ __launch_bounds__(NRQ_GEN_WG) void nrq_gen_kernel(rq_params p, uint32_t T, const uint8_t *__restrict__ inter,
                                                         size_t inter_stride, const uint32_t *__restrict__ isis,
                                                         uint8_t *__restrict__ out, size_t out_stride) {
  __shared__ uint32_t cols[RQ_MAX_LT_COLS];
  __shared__ uint32_t ncols;
  const uint32_t q = blockIdx.x, b = blockIdx.y;
  if (threadIdx.x == 0) {
    uint32_t tmp[RQ_MAX_LT_COLS];
    uint32_t n = rq_lt_columns(&p, isis[q], tmp);
    for (uint32_t k = 0; k < n; k++) cols[k] = tmp[k];
    ncols = n;
  }
  __syncthreads();
  const uint8_t *C = inter + (size_t)b * inter_stride;
  uint8_t *dst = out + (size_t)b * out_stride + (size_t)q * T;
  const uint32_t n = ncols;
  const bool vec = ((T & 15u) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0) &&
                   ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
  if (vec) {
    for (uint32_t off = threadIdx.x * 16u; off < T; off += NRQ_GEN_WG * 16u) {
      uint4 acc = make_uint4(0, 0, 0, 0);
      for (uint32_t k = 0; k < n; k++) {
        uint4 v = *reinterpret_cast<const uint4 *>(C + (size_t)cols[k] * T + off);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
      *reinterpret_cast<uint4 *>(dst + off) = acc;
    }
  } else {
    for (uint32_t off = threadIdx.x; off < T; off += NRQ_GEN_WG) {
      uint8_t acc = 0;
      for (uint32_t k = 0; k < n; k++) acc ^= C[(size_t)cols[k] * T + off];
      dst[off] = acc;
    }
  }
}

/* ============================================================================================
 * Host side
 * ========================================================================================== */
namespace {

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct DevBuf {
  uint8_t *p = nullptr;
  size_t cap = 0;
};
struct PinBuf {
  uint8_t *p = nullptr;
  size_t cap = 0;
};

struct KConst {
  uint8_t *host = nullptr;
  uint8_t *dev = nullptr;
  uint32_t bytes = 0;
};

struct EncPlan {
  bool valid = false;     /* false after nrq_plan_cache_clear: rebuilt on next use, buffers are kept */
  size_t dev_cap = 0;
  uint8_t *pin = nullptr; /* host planner: pinned image for the asynchronous upload; device planner: what comes back */
  size_t pin_cap = 0;
  uint8_t *dev = nullptr; /* the plan in use: plan arena and rowsrc (one of devbuf[]) */
  uint32_t plan_bytes = 0;
  uint32_t rowsrc_off = 0;
  nrq_plan_hdr hdr;
  std::vector<uint16_t> colslot; /* host copy, to translate LT neighbour lists into slots */
  double build_ms = 0;
  /* Two device buffers: a plan that is rebuilt ON THE DEVICE (big K', encplan_device_launch) goes to the buffer that is
   * not in use, on a stream of its own, while solves may still read the other one.  The host planner keeps to buffer 0
   * (its upload is ordered on the caller's stream). */
  uint8_t *devbuf[2] = {nullptr, nullptr};
  size_t devcap[2] = {0, 0};
  hipEvent_t used[2] = {nullptr, nullptr}; /* last solve launch that reads devbuf[i] */
  bool used_set[2] = {false, false};
  int cur = 0;
  bool pending = false;   /* a device build is in flight into devbuf[pend]; encplan_finish() completes it */
  int pend = 0;
  hipEvent_t ready = nullptr;
  uint32_t rb_bytes = 0;  /* device build: bytes of the arena's front (header .. colslot) read back */
  double t_launch = 0;
};

inline size_t r16(size_t x) { return (x + 15) & ~(size_t)15; }

} // namespace

/* Tuning / debugging knobs from the environment, read ONCE when the context is created (the launch path does not
 * call getenv). */
struct Tuning {
  bool map_spread = false;   /* NRQ_MAP_SPREAD: deal line groups round-robin instead of block octets per XCD */
  bool big_wg = false;       /* NRQ_BIG_WG: never use the 256-thread solve variants */
  bool small_waves4 = true;  /* NRQ_SMALL_WAVES5 clears it: the 256-thread variant compiled for 5 workgroups per CU (96 registers per
                              * thread) is used where five images fit; since the row pipeline forms its addresses ahead of the LDS wait the
                              * 4-workgroup one (128 registers) is faster there: K=1000 930 -> 980 Gbit/s */
  bool prof = false;         /* NRQ_PROF: per-phase shader-clock marks, printed to stderr */
  bool diag = false;         /* NRQ_DIAG: why a block was reported not decodable, to stderr */
  bool plan_lds_max = false; /* NRQ_PLAN_LDS_MAX: planner always takes the whole LDS */
  bool plan_big_wg = false;  /* NRQ_PLAN_BIG_WG: planner always 1024 threads */
  uint32_t small_div = 2;    /* NRQ_SMALL_DIV: LDS images per CU from which the 256-thread variant is used (measured: 2 beats 3) */
  uint64_t solve_grid = 0;   /* NRQ_SOLVE_GRID: persistent workgroups of the solve launch (0 = fill the device) */
  uint32_t max_wb = 16;      /* NRQ_MAX_WB: widest strip considered */
  bool no_wb12 = false;      /* NRQ_NO_WB12: strip widths 16, 8, 4, 2 only (round 5's set) */
  bool tiny_any = false;      /* "tiny_any" / NRQ_TINY_ANY: the solve's throughput forms whatever the launch size (tests): single-wave workgroups also for a
                               * few hundred strips, 256-thread ones also for a lone mid-size block */
  bool plan_pack = false;     /* NRQ_PLAN_PACK: small blocks' planner workgroups share a CU whatever the block count */
  bool host_plan_auto = true; /* NRQ_HOST_PLAN_AUTO=0: a call of one or two small blocks is planned by the planner kernel like any other */
  int prof_base = 2;         /* NRQ_PROF_BASE: stamp the free-form marks are measured from */
  uint32_t encplan_dev_min_l = 12000; /* NRQ_ENCPLAN_DEV_MIN_L: from this many intermediate symbols on, encode plans are built by
                              * the device planner, asynchronously (the host planner takes 25 ms at K=27000, 95 ms at K'=56403) */
  uint32_t wide_g = 0;       /* NRQ_WIDE_G: wide strips of G = 2, 4, 8 lanes per element where two such images fit a CU */
  bool no_wentry = false;    /* NRQ_NO_WENTRY: big blocks' entry pass by the planner workgroup itself, not by nrq_wentry_kernel */
  bool no_tiny = false;      /* NRQ_NO_TINY: no single-wave workgroups for tiny strip images */
  uint32_t tiny_div = 7;     /* NRQ_TINY_DIV: LDS images per CU from which the single-wave variant is used (launches with ONE plan: encode).
                              * (Twelve until the workgroups per CU were counted by allocated LDS, lds_alloc(): between eight and
                              * eleven images the ninth.. workgroup of a CU had run as a second round and the variant looked slow;
                              * with the count right it wins from seven on -- K=450 +22 %, K=500 +14 %, K=600 +12 %, K=700 +2 %.) */
  uint32_t tiny_div_dec = 7; /* NRQ_TINY_DIV_DEC: the same for launches with a plan per block (decode): every strip walks a plan of its own
                              * through L2 / HBM, and more independent strips in flight hide more of those trips.  (Seven: at six
                              * images per CU -- K=1000 once its plans have a few inactive columns fewer -- the single waves lose to
                              * the 256-thread workgroups, decode 11.1 against 7.4 ms per 2048 blocks; at seven, K=700, they win 5.2 : 6.2.) */
  bool no_split = false;     /* NRQ_NO_SPLIT: narrow strips also do their back-substitution in the solve kernel */
  bool no_balance = false;   /* NRQ_NO_BALANCE: keep whole-line work slots even when the rounds come out uneven */
  uint32_t lds_max = NRQ_LDS_MAX; /* "lds_max" (tests): LDS bytes a strip image may take when the batch's block lists are formed (pick_and_launch) */
  bool no_lists = false;     /* NRQ_NO_LISTS: one solve launch per batch at the width EVERY block fits (round 5), no second list */
  int reserve_cus = -1;      /* NRQ_RESERVE_CUS: compute units a big-block solve launch leaves to the planner (-1 = automatic) */
  bool plan_small_state = true;  /* NRQ_PLAN_BIG_STATE clears it: small blocks' planner workgroups keep the full-size queues */
  bool plan_split_force = false; /* "plan_split_force": every block planned in two parts + helper kernels (tests) */
  bool no_plan_split = false;  /* NRQ_NO_PLAN_SPLIT: big blocks planned by one kernel (no helper kernels for the HDPC fold / W transposition) */
  bool no_plan_stream = false; /* NRQ_NO_PLAN_STREAM: planner kernel on the caller's stream (no overlap with the solve before it) */
  bool plan_no_wg128 = false;  /* NRQ_PLAN_NO_WG128: the smallest blocks' planner workgroups stay at 256 threads */
  bool plan_wrong_instance = false; /* "plan_wrong_instance" (tests): blocks whose peeling state fits the LDS are given to the planner instance for
                                     * the others -- pl_init_a must notice (PL_PEEL_FORM_OK) and the blocks go to the host planner */
  uint32_t plan_ucap = 0;      /* "plan_ucap": inactive-column capacity of the device planner (0 = P + 768, at most 1280); tests lower it
                                * to send blocks through the capacity fallback (host re-plan) */
  void read() {
    auto flag = [](const char *n) { const char *e = getenv(n); return e != nullptr; };
    auto num = [](const char *n, long long d) { const char *e = getenv(n); return (e && *e) ? atoll(e) : d; };
    map_spread = flag("NRQ_MAP_SPREAD"); big_wg = flag("NRQ_BIG_WG"); small_waves4 = !flag("NRQ_SMALL_WAVES5");
    prof = flag("NRQ_PROF"); diag = flag("NRQ_DIAG"); plan_lds_max = flag("NRQ_PLAN_LDS_MAX"); plan_big_wg = flag("NRQ_PLAN_BIG_WG");
    small_div = (uint32_t)num("NRQ_SMALL_DIV", 2); solve_grid = (uint64_t)num("NRQ_SOLVE_GRID", 0);
    max_wb = (uint32_t)num("NRQ_MAX_WB", 16); prof_base = (int)num("NRQ_PROF_BASE", 2);
    encplan_dev_min_l = (uint32_t)num("NRQ_ENCPLAN_DEV_MIN_L", 12000);
    wide_g = (uint32_t)num("NRQ_WIDE_G", 0);
    if (wide_g != 2u && wide_g != 4u && wide_g != 8u) wide_g = 0;
    no_wentry = flag("NRQ_NO_WENTRY"); no_tiny = flag("NRQ_NO_TINY"); tiny_div = (uint32_t)num("NRQ_TINY_DIV", 7); tiny_div_dec = (uint32_t)num("NRQ_TINY_DIV_DEC", 7);
    no_split = flag("NRQ_NO_SPLIT"); no_balance = flag("NRQ_NO_BALANCE"); no_lists = flag("NRQ_NO_LISTS"); no_wb12 = flag("NRQ_NO_WB12"); host_plan_auto = num("NRQ_HOST_PLAN_AUTO", 1) != 0; plan_pack = flag("NRQ_PLAN_PACK"); tiny_any = flag("NRQ_TINY_ANY"); reserve_cus = (int)num("NRQ_RESERVE_CUS", -1); no_plan_stream = flag("NRQ_NO_PLAN_STREAM");
    no_plan_split = flag("NRQ_NO_PLAN_SPLIT");
    plan_small_state = !flag("NRQ_PLAN_BIG_STATE"); plan_no_wg128 = flag("NRQ_PLAN_NO_WG128");
  }
};

#define NRQ_PLAN_AHEAD_MAX 2u
struct PlanRun;
static void plan_ahead_drop(struct nrq_ctx *ctx);
struct nrq_ctx {
  int device = 0;
  long long fail_after = 0;      /* fault injection: checked runtime calls until the injected failure (0 = off) */
  bool fault_inject_armed = false; /* NANORQ_HIP_FAULT_INJECT=1 was in the environment when the context was created */
  long long faults_injected = 0;
  Tuning tune;
  int ncu = 256; /* compute units of the device */
  hipStream_t stream = nullptr;
  std::string err;
  std::map<uint32_t, KConst> kconst;    /* by K' */
  std::map<uint64_t, EncPlan> encplans; /* by (K', K) */
  DevBuf scratch[2];                    /* per-call device arrays (double-buffered across calls) */
  PinBuf staging[2];
  hipEvent_t staged[2] = {nullptr, nullptr};
  int flip = 0;
  int threads = 0;
  nrq_call_stats stats;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  hipEvent_t encplan_uploaded = nullptr;
  bool attr_set[5] = {false, false, false, false, false};
  /* optional per-launch timing of the solve kernel (HIP events on the launch stream) */
  bool ktime_on = false;
  bool ktime_outer = false; /* the solve launches in flight are bracketed by their caller's pair of events */
  hipEvent_t ktime_base = nullptr; /* recorded by nrq_ktime_enable: origin of the launch intervals */
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ktime_pool;
  size_t ktime_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ptime_pool; /* same for the planner launches (all kernels of a planner run) */
  size_t ptime_used = 0;
  unsigned long long *prof = nullptr; /* NRQ_PROF=1 */
  /* device planner */
  int planner = 1; /* 1 = device planner for decode (default), 0 = host planner */
  /* Planner runs may be issued AHEAD of their decode call (nrq_decode_plan_ahead), up to NRQ_PLAN_AHEAD_MAX of them: plan
   * arenas / job records come in three sets in turn (one being read by the solve in flight, two being written), the planner
   * workspace and the planner stream in two (two runs side by side: a planner workgroup is latency bound on ONE compute unit,
   * so two batches' planners on twice the compute units take the time of one), the input / header staging in four. */
  DevBuf plan_work[2], plan_arena[3], plan_jobs[3];
  uint32_t prun = 0; /* planner runs launched so far (chooses stream and workspace) */
  uint32_t ahead_hint = 0; /* most planner runs that were waiting at once since the last discard: sizes the CU reserve of big-block launches */
  DevBuf stage; /* solve kernel: staging buffers of the persistent workgroups */
  DevBuf ybuf;  /* split solve of narrow strips: per block, (M + u) full-width rows (slot image + inactive columns) */
  DevBuf jobs_b; /* pick_and_launch: the job records of a batch's second (narrower) block list, side by side */
  /* nrq_dev_alloc / nrq_dev_free: a caching pool (the object API allocates per call; hipMalloc / hipFree are
   * device-wide synchronisation points).  Freed blocks are reused for requests of up to 1.25x less; reuse is safe
   * because all work on a block is ordered on the context's streams and the object layer waits for its copy
   * streams before it frees. */
  const uint64_t *vec_inter = nullptr; /* nrq_encode_blocks_v: per-block addresses of the intermediate symbols */
  const uint64_t *vec_src = nullptr, *vec_rep = nullptr; /* nrq_decode_blocks_v: per-block buffer addresses instead of base + stride */
  bool io_aligned = false;              /* set by the entry points: every symbol row of the call is 16-byte aligned (base addresses and strides) */
  uint32_t chunk_blocks = 0;            /* nrq_decode_blocks_vc: blocks per solve launch (0: one launch for all) */
  void *const *chunk_done = nullptr, *const *chunk_up = nullptr;
  std::multimap<size_t, void *> pool_free;
  std::map<void *, size_t> pool_size;
  size_t pool_cached = 0;
  hipStream_t aux[3] = {nullptr, nullptr, nullptr}; /* streams of the object layer: [0] host -> device copies, [1] device -> host copies,
                                                     * [2] kernels that sort uploaded packets into rows beside both */
  DevBuf scat_dev[2];
  PinBuf scat_pin[2];
  hipEvent_t scat_ev[2] = {nullptr, nullptr};
  int scat_flip = 0;
  bool plan_attr = false;
  /* The planner kernel runs on a stream of its own: it depends on the reception pattern only, not on the symbols, so
   * it may run beside whatever the caller's stream is still doing (typically the encode solve launched just before).
   * `planned`: planner + header download done (the host waits for it, the solve launch on the caller's stream is
   * ordered behind it); `arena_free`: the solve that reads the plan arenas / job records has finished -- the next
   * planner launch overwrites them and waits for it. */
  hipStream_t plan_stream = nullptr;
  hipStream_t plan_stream_b = nullptr; /* the second planner stream (runs issued ahead alternate) */
  hipEvent_t planned[3] = {nullptr, nullptr, nullptr}, arena_free[3] = {nullptr, nullptr, nullptr};
  bool arena_busy[3] = {false, false, false};
  int aflip = 0;
  std::deque<struct PlanRun *> ahead; /* planner runs issued by nrq_decode_plan_ahead and not yet consumed, oldest first */
  hipStream_t plan_stream2 = nullptr; /* encode plans built on the device (encplan_device_launch): beside both of the above */
  DevBuf encplan_work;                /* planner workspace of that build */
  DevBuf pscratch[4]; /* planner inputs: buffers of their own (the per-call arrays above belong to the caller's stream) */
  PinBuf pstaging[4];
  hipEvent_t pstaged[4] = {nullptr, nullptr, nullptr, nullptr};
  int pflip = 0;
};

namespace {

#ifdef NRQ_NO_FAULT_INJECT
static inline bool nrq_inject(nrq_ctx *) { return false; } /* (a build without the hook: -DNRQ_NO_FAULT_INJECT) */
#else
static inline bool nrq_inject(nrq_ctx *ctx) {
  if (!ctx || ctx->fail_after <= 0) return false; /* (fail_after can only be set on a context created with NANORQ_HIP_FAULT_INJECT=1) */
  if (--ctx->fail_after > 0) return false;
  ctx->faults_injected++;
  return true;
}
#endif

int fail(nrq_ctx *ctx, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

/* Fault injection (nrq_ctx_set_option "fail_after" n): the n-th checked runtime call of the context from now on -- an
 * allocation, a copy, an event or stream operation, the error check behind a launch -- is not made and reports an error
 * instead, once.  The tests drive the error paths of the object layer with it (rollback of a packet batch, a failed chunk of
 * nanorq_repair_all, ...); nothing else sets it.  It is a TEST facility: the option exists only on a context created while
 * the environment holds NANORQ_HIP_FAULT_INJECT=1 (tests/conftest.py, tools/sanitize.sh set it) -- in any other process
 * "fail_after" is an unknown option, so no caller of the public nanorq_hip_option can make a runtime call fail -- and
 * -DNRQ_NO_FAULT_INJECT builds the library without the hook altogether. */
static inline bool nrq_inject(nrq_ctx *ctx);
#define HIPCHK(ctx, call)                                                                                   \
  do {                                                                                                      \
    hipError_t e_ = nrq_inject(ctx) ? hipErrorUnknown : (call);                                             \
    if (e_ != hipSuccess) return fail(ctx, -10, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                                      __FILE__, __LINE__);                                                  \
  } while (0)

int ensure_dev(nrq_ctx *ctx, DevBuf &b, size_t bytes) {
  if (b.cap >= bytes) return 0;
  if (b.p) HIPCHK(ctx, hipFree(b.p));
  b.p = nullptr; b.cap = 0;
  size_t want = bytes + bytes / 4 + 4096;
  HIPCHK(ctx, hipMalloc((void **)&b.p, want));
  b.cap = want;
  return 0;
}
int ensure_pin(nrq_ctx *ctx, PinBuf &b, size_t bytes) {
  if (b.cap >= bytes) return 0;
  if (b.p) HIPCHK(ctx, hipHostFree(b.p));
  b.p = nullptr; b.cap = 0;
  size_t want = bytes + bytes / 4 + 4096;
  HIPCHK(ctx, hipHostMalloc((void **)&b.p, want, hipHostMallocDefault));
  b.cap = want;
  return 0;
}

/* parameters for a block of K symbols coded with the table row K' = Kp (0: the row RFC 6330 assigns to
 * K).  nanorq codes every block of an object with block 0's row (lib/nanorq.c:289, :372), so a short
 * last block can carry a K' larger than its own. */
int block_params(nrq_ctx *ctx, uint32_t K, uint32_t Kp, rq_params *p) {
  if (!rq_params_init(Kp ? Kp : K, p)) return fail(ctx, -1, "K=%u K'=%u out of range", K, Kp);
  if (K == 0 || K > p->Kp || (Kp && p->Kp != Kp)) return fail(ctx, -1, "K=%u does not fit table row K'=%u", K, Kp);
  p->K = K;
  return 0;
}

/* the planner kernels may own the whole LDS of a CU (also the first touch of the code object: the runtime loads it here) */
static int plan_attr_once(nrq_ctx *ctx) {
  if (ctx->plan_attr) return 0;
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_plan_kernel<(int)PL_NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_plan_kernel<(int)PL_NT, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_plan_kernel<(int)PL_NT_MIN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_plan_kernel<(int)PL_NT_TINY>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_mh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)NRQ_LDS_MAX));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_wpass_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)NRQ_LDS_MAX));
  ctx->plan_attr = true;
  return 0;
}

int get_kconst(nrq_ctx *ctx, uint32_t K, KConst **out) {
  rq_params p;
  if (!rq_params_init(K, &p)) return fail(ctx, -1, "K=%u out of range", K);
  auto it = ctx->kconst.find(p.Kp);
  if (it == ctx->kconst.end()) {
    KConst kc;
    if (nrq_host_kconst_build(K, &kc.host, &kc.bytes) != 0) return fail(ctx, -2, "kconst build failed");
    HIPCHK(ctx, hipMalloc((void **)&kc.dev, kc.bytes));
    HIPCHK(ctx, hipMemcpy(kc.dev, kc.host, kc.bytes, hipMemcpyHostToDevice));
    it = ctx->kconst.emplace(p.Kp, kc).first;
  }
  *out = &it->second;
  return 0;
}

/* launch geometry of the planner kernel: LDS sizing, workgroup shape (shared by the decode planner and the device
 * build of encode plans) */
/* Big blocks, whose peeling state does not fit the LDS next to the dense-stage reserve (pl_ctx_setup's rule), run the
 * planner in two parts with helper kernels between and after them (planner_seq.h).  The jobs carry the choice in
 * bit 8 of nrq_planjob::mode (pl_final_c then leaves the W transposition to nrq_wt_kernel). */
bool plan_is_segmented(const nrq_ctx *ctx, const rq_params &p, uint32_t Mcap) {
  const uint32_t sh_bytes = pl_shared_bytes(PL_QCAP, PL_LOWCAP, PL_NT), dyn = NRQ_LDS_MAX - sh_bytes;
  if (ctx->tune.plan_split_force) return true; /* (tests: the segmented path at sizes the oracle checks quickly) */
  return pl_state_in_lds(p.L, Mcap, dyn) == 0u && !ctx->tune.no_plan_split;
}

int launch_plan_kernel(nrq_ctx *ctx, hipStream_t ps, const rq_params &p, const uint8_t *d_kc, const nrq_planjob *d_pj,
                       nrq_job *d_jobs, uint32_t nblk, uint32_t Mcap, uint32_t npcap, uint32_t ucap, unsigned long long *pprof,
                       uint32_t nnzcap) {
  /* The workgroup state (pl_shared and the arrays behind it: frontier queues, claim lists, per-thread scratch, Gauss-Jordan
   * flags) is sized by the launch: 25 KB for big blocks; a small block's frontier and dense stage need a fraction, and
   * with 8 KB of it four 256-thread planner workgroups share a CU instead of two (the planner is latency bound: twice
   * the workgroups, half the time).  Overflowing a capacity is reported as such and re-planned on the host. */
  uint32_t qcap = PL_QCAP, lowcap = PL_LOWCAP;
  uint32_t sh_bytes = pl_shared_bytes(qcap, lowcap, PL_NT);
  /* dynamic LDS: everything a CU has for a big block; for small blocks what the planner can use (peeling state plus the
   * dense-stage reserve, or a 16-byte strip image of the W rows), so that several workgroups share a CU */
  uint32_t dyn_bytes = NRQ_LDS_MAX - sh_bytes;
  bool small_wg = false; /* 256-thread workgroups: a small block has no use for 1024 threads, a CU has for 4 blocks */
  bool tiny_wg = false;  /* 128-thread workgroups (with small_wg) */
  {
    const uint32_t peel = 2u * pl_r16(Mcap * 4u) + pl_r16(p.L * 4u) + pl_dense_reserve(p.L);
    const uint32_t wimg = (Mcap + 320u + NRQ_SCRATCH) * 16u;
    const uint32_t fit = pl_r16((peel > wimg ? peel : wimg) + 2048u);
    /* (queues / claim lists / Gauss-Jordan flags: a frontier, a round's claims and the leftover rows are at most the block's rows) */
    const uint32_t q_s = Mcap <= 248u ? 256u : p.L <= 1500u ? 512u : 1024u, low_s = Mcap <= 248u ? 256u : p.L <= 1500u ? 384u : 768u;
    const uint32_t sh_s = ctx->tune.plan_small_state ? pl_shared_bytes(q_s, low_s, PL_NT_MIN) : sh_bytes;
    /* ... when there are more blocks than compute units.  A batch of at most one block per CU gains nothing from sharing: every block
     * gets the 1024-thread workgroup and the whole LDS (round 6, planner per batch, 256-thread / 1024-thread workgroups: K=500 x 256
     * blocks 0.55 / 0.39 ms, K=1000 x 256 0.53 / 0.44, K=2500 x 256 0.95 / 0.69, K=2500 x 64 1.15 / 0.79; one block of K=2500 through
     * the reference's benchmark.c: decode column 24.1 -> 30.8 Gbit/s).  "plan_pack" = 1 packs regardless (tests of the small forms). */
    const bool pack = nblk > (uint32_t)ctx->ncu || ctx->tune.plan_pack;
    if (lds_alloc(fit + sh_s) <= NRQ_LDS_MAX / 2u && !ctx->tune.plan_lds_max) {
      dyn_bytes = fit; /* (also for the 1024-thread workgroup of a small batch: it then leaves LDS and wave slots to a solve running beside it) */
      small_wg = !ctx->tune.plan_big_wg && pack;
      if (small_wg && ctx->tune.plan_small_state) { qcap = q_s; lowcap = low_s; sh_bytes = sh_s; }
      /* The smallest blocks: 128 threads.  A planner phase is one wave's chain of instructions and trips,
       * the other waves of the workgroup mostly wait; the registers of the kernel (~100) let a CU hold 20 waves -- five
       * 256-thread workgroups, or as many 128-thread ones as the LDS takes (six or more from here on): more blocks in flight
       * for the same waves. */
      const uint32_t sh_t = pl_shared_bytes(q_s, low_s, PL_NT_TINY);
      if (small_wg && ctx->tune.plan_small_state && !ctx->tune.plan_no_wg128 && lds_alloc(fit + sh_t) * 6u <= NRQ_LDS_MAX) { tiny_wg = true; sh_bytes = sh_t; }
    }
  }
  const bool seg = plan_is_segmented(ctx, p, Mcap);
  if (seg && ctx->tune.plan_split_force) {
    /* a segmented run keeps nothing in LDS between its parts: its peeling state must live in the workspace, which
     * pl_ctx_setup chooses when the dynamic region is too small for it -- so make it too small (dense stage only) */
    const uint32_t need = 2u * pl_r16(Mcap * 4u) + pl_r16(p.L * 4u);
    uint32_t only_dense = (pl_dense_reserve(p.L) + need - 16u) & ~15u; /* 16 bytes short of holding the peeling state */
    if (only_dense < dyn_bytes) dyn_bytes = only_dense;
    if (pl_state_in_lds(p.L, Mcap, dyn_bytes) != 0u) dyn_bytes = (need - 16u) & ~15u; /* (... also for the form that shares the rowstate image) */
    small_wg = false; tiny_wg = false;
    qcap = PL_QCAP; lowcap = PL_LOWCAP; sh_bytes = pl_shared_bytes(qcap, lowcap, PL_NT);
  }
  const uint32_t mh_dyn = 72u * 1024u; /* nrq_mh_kernel: MhT (16 B x u <= 20 KB) + the tiles (4 KB + 256 x wpr words <= 40 KB) */
  { int rc_ = plan_attr_once(ctx); if (rc_) return rc_; }
  /* parts of the run: everything | 3, (entry pass), 4, (W pass, HDPC fold), 2 */
  const bool wentry = seg && !ctx->tune.no_wentry;
  const uint32_t parts_seg[3] = {wentry ? 3u : 1u, wentry ? 4u : 2u, 2u};
  const uint32_t nparts_run = !seg ? 1u : wentry ? 3u : 2u;
  for (uint32_t pi = 0; pi < nparts_run; pi++) {
    const uint32_t part = seg ? parts_seg[pi] : 0u;
    const bool hbm_state = pl_state_in_lds(p.L, Mcap, dyn_bytes) == 0u; /* (pl_ctx_setup's rule) */
    if (part && (tiny_wg || small_wg || !hbm_state)) return fail(ctx, -2, "planner: a segmented run needs the instance for big blocks");
    if (tiny_wg)
      hipLaunchKernelGGL(nrq_plan_kernel<(int)PL_NT_TINY>, dim3(nblk), dim3(PL_NT_TINY), dyn_bytes + sh_bytes, ps, p, d_kc, d_pj, d_jobs,
                         nblk, Mcap, npcap, ucap, dyn_bytes, pprof, part, qcap, lowcap);
    else if (small_wg)
      hipLaunchKernelGGL(nrq_plan_kernel<(int)PL_NT_MIN>, dim3(nblk), dim3(PL_NT_MIN), dyn_bytes + sh_bytes, ps, p, d_kc, d_pj, d_jobs,
                         nblk, Mcap, npcap, ucap, dyn_bytes, pprof, part, qcap, lowcap);
    else if (hbm_state || ctx->tune.plan_wrong_instance) /* (the state stays in HBM) */
      hipLaunchKernelGGL((nrq_plan_kernel<(int)PL_NT, 1>), dim3(nblk), dim3(PL_NT), dyn_bytes + sh_bytes, ps, p, d_kc, d_pj, d_jobs, nblk,
                         Mcap, npcap, ucap, dyn_bytes, pprof, part, qcap, lowcap);
    else
      hipLaunchKernelGGL(nrq_plan_kernel<(int)PL_NT>, dim3(nblk), dim3(PL_NT), dyn_bytes + sh_bytes, ps, p, d_kc, d_pj, d_jobs, nblk,
                         Mcap, npcap, ucap, dyn_bytes, pprof, part, qcap, lowcap);
    HIPCHK(ctx, hipGetLastError());
    if (part == 3u) {
      uint32_t nw = 64u / (nblk ? nblk : 1u); /* workgroups per block: what a batch of few big blocks finds free beside the solves */
      if (nw < 2u) nw = 2u;
      if (nw > 8u) nw = 8u;
      hipLaunchKernelGGL(nrq_wentry_kernel, dim3(nw, nblk), dim3(1024), sh_bytes, ps, p, d_kc, d_pj, Mcap, npcap, ucap, dyn_bytes);
      HIPCHK(ctx, hipGetLastError());
    }
    if (part == 1u || part == 4u) {
      const uint32_t wp_lds = (Mcap + NRQ_SCRATCH) * 2u + 64u;
      hipLaunchKernelGGL(nrq_wpass_kernel, dim3(((ucap + 31u) / 32u) * 2u, nblk), dim3(256), wp_lds, ps, p, d_kc, d_pj, Mcap, npcap, ucap);
      HIPCHK(ctx, hipGetLastError());
      /* the HDPC fold: as many workgroups per block as leave the whole batch ~256 */
      uint32_t nparts = 256u / (nblk ? nblk : 1u);
      if (nparts < 1u) nparts = 1u;
      if (nparts > 64u) nparts = 64u;
      hipLaunchKernelGGL(nrq_mh_kernel, dim3(nparts, nblk), dim3(1024), mh_dyn + sh_bytes, ps, p, d_kc, d_pj, Mcap, npcap, ucap, mh_dyn);
      HIPCHK(ctx, hipGetLastError());
    }
  }
  if (seg) {
    hipLaunchKernelGGL(nrq_wt_kernel, dim3(64, nblk), dim3(256), 0, ps, d_pj, p.L, Mcap, npcap, ucap, nnzcap);
    HIPCHK(ctx, hipGetLastError());
  }
  return 0;
}

int ensure_encbuf(nrq_ctx *ctx, EncPlan &ep, int i, size_t total) {
  if (ep.devcap[i] >= total) return 0;
  if (ep.devbuf[i]) HIPCHK(ctx, hipFree(ep.devbuf[i]));
  ep.devbuf[i] = nullptr; ep.devcap[i] = 0;
  HIPCHK(ctx, hipMalloc((void **)&ep.devbuf[i], total + total / 8));
  ep.devcap[i] = total + total / 8;
  return 0;
}

/* host planner: build, stage in pinned memory, upload on the caller's stream (buffer 0) */
int encplan_host_build(nrq_ctx *ctx, const rq_params &p, uint32_t K, KConst *kc, EncPlan &ep) {
  double t0 = now_ms();
  std::vector<uint32_t> isis(p.Kp);
  for (uint32_t j = 0; j < p.Kp; j++) isis[j] = j;
  uint8_t *arena = nullptr;
  uint32_t bytes = 0;
  if (nrq_host_plan_build(p.Kp, p.Kp, isis.data(), kc->host, &arena, &bytes) != 0)
    return fail(ctx, -2, "encode plan build failed for K=%u", K);
  memcpy(&ep.hdr, arena, sizeof(ep.hdr));
  if (ep.hdr.status) { nrq_host_free(arena); return fail(ctx, -3, "encode matrix singular for K=%u (cannot happen)", K); }
  ep.plan_bytes = bytes;
  ep.colslot.assign(reinterpret_cast<const uint16_t *>(arena + ep.hdr.off_colslot),
                    reinterpret_cast<const uint16_t *>(arena + ep.hdr.off_colslot) + p.L);
  ep.rowsrc_off = (uint32_t)r16(bytes);
  const size_t total = ep.rowsrc_off + (size_t)p.L * 4;
  /* device and pinned buffers survive a cache clear (same K => same size class); the upload is
   * asynchronous on the context's stream, so rebuilding a plan never waits for the GPU */
  int rc = ensure_encbuf(ctx, ep, 0, total);
  if (rc) { nrq_host_free(arena); return rc; }
  if (ep.pin_cap < total) {
    if (ep.pin) HIPCHK(ctx, hipHostFree(ep.pin));
    ep.pin = nullptr;
    HIPCHK(ctx, hipHostMalloc((void **)&ep.pin, total + total / 8, hipHostMallocDefault));
    ep.pin_cap = total + total / 8;
  } else {
    /* the previous upload from this pinned image must have been consumed */
    HIPCHK(ctx, hipEventSynchronize(ctx->encplan_uploaded));
  }
  memcpy(ep.pin, arena, bytes);
  uint32_t *rowsrc = reinterpret_cast<uint32_t *>(ep.pin + ep.rowsrc_off);
  for (uint32_t r = 0; r < p.L; r++) rowsrc[r] = NRQ_ROW_ZERO;
  for (uint32_t j = 0; j < K; j++) rowsrc[p.S + p.H + j] = j;
  nrq_host_free(arena);
  HIPCHK(ctx, hipMemcpyAsync(ep.devbuf[0], ep.pin, total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->encplan_uploaded, ctx->stream));
  ep.cur = 0;
  ep.dev = ep.devbuf[0];
  ep.valid = true;
  ep.pending = false;
  ep.build_ms = now_ms() - t0;
  return 0;
}

/* Device planner for an encode plan: the constraint matrix of an encode is that of a decode in which nothing was
 * replaced -- the planner kernel runs it as a job without missing symbols (nrq_planjob::mode).  Asynchronous, on
 * plan_stream2, into the buffer that is not in use; encplan_finish() waits for it and reads the header, the job
 * record and colslot[] back.  (reference: nanorq_precalculate, lib/nanorq.c:393-401) */
int encplan_device_launch(nrq_ctx *ctx, const rq_params &p, uint32_t K, KConst *kc, EncPlan &ep) {
  const nrq_kconst_hdr *kh = reinterpret_cast<const nrq_kconst_hdr *>(kc->host);
  uint32_t ucap = p.P + 768u;
  if (ucap > 1280u) ucap = 1280u;
  if (ucap < p.P + 32u) return 1; /* not for the device planner */
  const uint32_t Mcap = p.L + PL_EXTRA_ROWS + 8u, npcap = PL_EXTRA_ROWS + 8u;
  const pl_work_layout wl = pl_work_plan(p.L, Mcap, npcap, ucap, kh->nnz + npcap * PL_PATCH_STRIDE);
  const uint32_t arena_cap = pl_arena_bound(p.L, Mcap, ucap, kh->nnz + npcap * PL_PATCH_STRIDE, 8u);
  /* front of the arena that comes back: header, pivslot, pivcol, colslot (pl_ctx_setup's layout) */
  const uint32_t rb = (uint32_t)(r16(r16(r16(r16(sizeof(nrq_plan_hdr)) + p.L * 2u) + p.L * 2u) + p.L * 2u));
  const size_t off_pj = arena_cap, off_job = off_pj + r16(sizeof(nrq_planjob)), dev_total = off_job + r16(sizeof(nrq_job));
  const size_t pin_pj = 0, pin_rb = r16(sizeof(nrq_planjob)), pin_job = pin_rb + rb, pin_total = pin_job + r16(sizeof(nrq_job));
  const int buf = (ep.devbuf[ep.cur] && (ep.valid || ep.used_set[ep.cur])) ? ep.cur ^ 1 : ep.cur;
  int rc;
  if ((rc = ensure_encbuf(ctx, ep, buf, dev_total))) return rc;
  if ((rc = ensure_dev(ctx, ctx->encplan_work, wl.total))) return rc;
  /* (the pinned image may still be the source of a host-built plan's asynchronous upload: encplan_host_build after a failed
   * device build -- wait for it before the job record is written over its first bytes) */
  HIPCHK(ctx, hipEventSynchronize(ctx->encplan_uploaded));
  if (ep.pin_cap < pin_total) {
    if (ep.pin) HIPCHK(ctx, hipHostFree(ep.pin));
    ep.pin = nullptr;
    HIPCHK(ctx, hipHostMalloc((void **)&ep.pin, pin_total + pin_total / 8, hipHostMallocDefault));
    ep.pin_cap = pin_total + pin_total / 8;
  }
  if (!ep.ready) HIPCHK(ctx, hipEventCreateWithFlags(&ep.ready, hipEventDisableTiming));
  for (int i = 0; i < 2; i++)
    if (!ep.used[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ep.used[i], hipEventDisableTiming));
  hipStream_t ps = ctx->plan_stream2;
  if (ep.used_set[buf]) HIPCHK(ctx, hipStreamWaitEvent(ps, ep.used[buf], 0));
  nrq_planjob *pj = reinterpret_cast<nrq_planjob *>(ep.pin + pin_pj);
  memset(pj, 0, sizeof(*pj));
  pj->work = (uint64_t)(uintptr_t)ctx->encplan_work.p;
  pj->arena = (uint64_t)(uintptr_t)ep.devbuf[buf];
  pj->arena_cap = arena_cap;
  pj->mode = 1u | (plan_is_segmented(ctx, p, Mcap) ? (ctx->tune.no_wentry ? 0x100u : 0x300u) : 0u);
  HIPCHK(ctx, hipMemcpyAsync(ep.devbuf[buf] + off_pj, pj, sizeof(*pj), hipMemcpyHostToDevice, ps));
  rq_params pk = p;
  pk.K = K;
  if ((rc = launch_plan_kernel(ctx, ps, pk, kc->dev, reinterpret_cast<const nrq_planjob *>(ep.devbuf[buf] + off_pj),
                               reinterpret_cast<nrq_job *>(ep.devbuf[buf] + off_job), 1u, Mcap, npcap, ucap, nullptr,
                               kh->nnz + npcap * PL_PATCH_STRIDE)))
    return rc;
  HIPCHK(ctx, hipMemcpyAsync(ep.pin + pin_rb, ep.devbuf[buf], rb, hipMemcpyDeviceToHost, ps));
  HIPCHK(ctx, hipMemcpyAsync(ep.pin + pin_job, ep.devbuf[buf] + off_job, sizeof(nrq_job), hipMemcpyDeviceToHost, ps));
  HIPCHK(ctx, hipEventRecord(ep.ready, ps));
  ep.pending = true;
  ep.pend = buf;
  ep.rb_bytes = rb;
  ep.t_launch = now_ms();
  return 0;
}

int encplan_finish(nrq_ctx *ctx, const rq_params &p, uint32_t K, KConst *kc, EncPlan &ep) {
  if (!ep.pending) return 0;
  HIPCHK(ctx, hipEventSynchronize(ep.ready));
  ep.pending = false;
  const uint8_t *rbp = ep.pin + r16(sizeof(nrq_planjob));
  nrq_plan_hdr hd;
  memcpy(&hd, rbp, sizeof(hd));
  nrq_job jb;
  memcpy(&jb, rbp + ep.rb_bytes, sizeof(jb));
  if (hd.magic != NRQ_PLAN_MAGIC || hd.status != 0 || hd.off_colslot + p.L * 2u > ep.rb_bytes || jb.plan != (uint64_t)(uintptr_t)ep.devbuf[ep.pend]) {
    /* a planner capacity was exceeded (or worse): the host planner takes over */
    if (ctx->tune.prof) fprintf(stderr, "[NRQ_PROF] encode plan K'=%u: device build failed (status %u, planner_body.h:%u), host planner\n",
                                p.Kp, hd.reserved[0], hd.fail_site);
    return encplan_host_build(ctx, p, K, kc, ep);
  }
  ep.hdr = hd;
  ep.colslot.assign(reinterpret_cast<const uint16_t *>(rbp + hd.off_colslot), reinterpret_cast<const uint16_t *>(rbp + hd.off_colslot) + p.L);
  ep.plan_bytes = hd.total_bytes;
  ep.rowsrc_off = (uint32_t)(jb.rowsrc - jb.plan);
  ep.cur = ep.pend;
  ep.dev = ep.devbuf[ep.cur];
  ep.valid = true;
  ep.build_ms = now_ms() - ep.t_launch;
  return 0;
}

/* The encode plan of (K', K): cached; built by the host planner (synchronously, uploaded on the caller's stream) or,
 * for big K', by the device planner -- asynchronously: with finish == false (nrq_precalculate) the call returns once
 * the build is enqueued, and the encode that needs the plan completes it. */
int get_encplan(nrq_ctx *ctx, uint32_t K, uint32_t Kp, EncPlan **out, bool finish = true) {
  rq_params p;
  int rc = block_params(ctx, K, Kp, &p);
  if (rc) return rc;
  const uint64_t key = ((uint64_t)p.Kp << 32) | K;
  auto it = ctx->encplans.find(key);
  if (it != ctx->encplans.end() && it->second.valid) { *out = &it->second; return 0; }
  KConst *kc;
  rc = get_kconst(ctx, p.Kp, &kc);
  if (rc) return rc;
  if (it == ctx->encplans.end()) it = ctx->encplans.emplace(key, EncPlan()).first;
  EncPlan &ep = it->second;
  *out = &ep;
  if (!ep.pending) {
    rc = 1;
    if (ctx->planner && p.L >= ctx->tune.encplan_dev_min_l) rc = encplan_device_launch(ctx, p, K, kc, ep);
    if (rc < 0) return rc;
    if (rc > 0) return encplan_host_build(ctx, p, K, kc, ep);
  }
  return finish ? encplan_finish(ctx, p, K, kc, ep) : 0;
}

/* LT neighbour lists of the symbols to generate, already translated to LDS slots through the plan's
 * colslot[], in the layout ph_store() reads */
void build_out_lists(const rq_params &p, const uint16_t *colslot, uint32_t n, const uint32_t *isis,
                     std::vector<uint32_t> &cptr, std::vector<uint16_t> &cols) {
  cptr.resize(n + 1);
  cols.clear();
  cols.reserve((size_t)n * 9);
  uint32_t tmp[RQ_MAX_LT_COLS];
  for (uint32_t q = 0; q < n; q++) {
    cptr[q] = (uint32_t)cols.size();
    uint32_t m = rq_lt_columns(&p, isis[q], tmp);
    for (uint32_t k = 0; k < m; k++) cols.push_back(colslot[tmp[k]]); /* slot that holds C[col] */
  }
  cptr[n] = (uint32_t)cols.size();
}

template <int WB> int launch_wb(nrq_ctx *ctx, int slot, const nrq_job *d_jobs, uint32_t nblk, uint32_t T,
                                const uint8_t *d_kc, uint32_t lds_bytes, uint32_t max_slots, uint32_t max_out, uint32_t max_u,
                                uint32_t max_wpr, const std::vector<const nrq_plan_hdr *> &hdrs) {
  /* WIDE strips (G lanes per element, 16 * G bytes per strip; solve_body.h) -- an experiment for small blocks, whose levels
   * hold a dozen ops and whose HDPC / dense phases a dozen rows, so that most lanes of a wave idle through them on a
   * 16-byte strip.  Measured (K=100 / 500 / 1000, G = 8 / 4 / 2, two or three 256-thread workgroups per CU): 270-313 /
   * 600-619 / 727-731 Gbit/s against 326 / 613 / 881 with 16-byte strips: the LDS holds the same number of symbol bytes
   * either way, a strip's chain of phases is no shorter for being wider, and G x fewer virtual threads make its
   * per-thread loops longer.  Not selected automatically; NRQ_WIDE_G / "wide_g" forces it (tests keep it correct). */
  uint32_t G = 1;
  if (WB == 16 && ctx->tune.wide_g > 1u && T >= 16u * ctx->tune.wide_g) {
    uint32_t need = 0;
    for (const nrq_plan_hdr *h : hdrs)
      if (!h->status) { const uint32_t t = nrq_lds_plan(h, 16u * ctx->tune.wide_g).total; if (t > need) need = t; }
    if (lds_alloc(need) * 2u <= NRQ_LDS_MAX) { G = ctx->tune.wide_g; lds_bytes = need; }
  }
  const uint32_t WBE = WB * G;
  /* narrow strips: the solve kernel stops after the dense stage, nrq_backsub_kernel / nrq_collect_kernel finish on
   * full-width rows of a per-block work buffer (see there) */
  const bool split = WB <= 4 && !ctx->tune.no_split;
  const uint32_t res_elems = max_out; /* rows nrq_collect_kernel writes per block at most */
  if (split) max_out = max_slots + max_u;
  const uint32_t nstrips = (T + WBE - 1) / WBE, spl = nrq_group_strips(WBE);
  const bool by_block = nrq_map_by_block(nblk) && !ctx->tune.map_spread;
  /* workgroup shape: the full-size workgroup when a strip image needs more than half of the CU's LDS, 256-thread ones
   * when two or more fit */
  /* (a launch whose strips each get a CU of their own -- a lone block of the reference's harness -- takes the full-size workgroup from
   * K ~ 1200 on: encode column of benchmark.c K=1500 208 -> 225 Gbit/s, K=3000 294 -> 315, K=4000 313 -> 333; below, the same) */
  const bool lone = (uint64_t)nblk * ((T + WB - 1) / WB) <= (uint64_t)ctx->ncu && max_slots >= 1200u && !ctx->tune.tiny_any;
  const bool small = WB != 12 && lds_alloc(lds_bytes) * ctx->tune.small_div <= NRQ_LDS_MAX && !ctx->tune.big_wg && !lone; /* (12-byte strips: big blocks) */
  /* single-wave workgroups when 12 or more images fit a CU (see the kernel; K=256: +26 % over five 256-thread workgroups) */
  const uint32_t tdiv = hdrs.size() > 1u ? ctx->tune.tiny_div_dec : ctx->tune.tiny_div;
  /* ... and when the launch has the strips to fill them: a lone block's 80 strips each get a 256-thread workgroup and a CU of their own
   * (the reference's benchmark.c, one block per call, K=500: encode column 81 -> 106 Gbit/s without the single-wave form; from ~1000
   * strips on -- 16 blocks of K=500 -- the single-wave form is the faster one again: 0.07 against 0.09 ms) */
  const bool tiny = G == 1 && small && (uint64_t)lds_alloc(lds_bytes) * tdiv <= NRQ_LDS_MAX && !ctx->tune.no_tiny &&
                    ((uint64_t)nblk * nstrips > 2u * (uint64_t)ctx->ncu || ctx->tune.tiny_any);
  const uint32_t nt = tiny ? 64u : small ? 256u : (uint32_t)NRQ_WG;
  uint32_t occ = NRQ_LDS_MAX / lds_alloc(lds_bytes ? lds_bytes : 1u);
  if (occ > 2048u / nt) occ = 2048u / nt;
  /* registers: the 256-thread variant (one wave per SIMD) is compiled for NRQ_SMALL_WAVES waves per SIMD.  More
   * workgroups than are resident at once would run as a second, thinner round of a statically partitioned job. */
  const bool five = G == 1 && small && !tiny && occ >= 5u && !ctx->tune.small_waves4;
  if (tiny) { if (occ > NRQ_TINY_OCC) occ = NRQ_TINY_OCC; } /* one wave per workgroup, compiled for 5 waves per SIMD; 20 per CU by the LDS sum, but
                                              * measured: with 20 x 256 workgroups not all are resident and the rest runs as a second
                                              * round (10.4 ms against 8.7 ms with 18 x 256 at K=100, T=1024, 8192 blocks) */
  else if (small && occ > (five ? 5u : (uint32_t)NRQ_SMALL_WV)) occ = five ? 5u : (uint32_t)NRQ_SMALL_WV;
  if (tiny && ctx->tune.diag) fprintf(stderr, "[NRQ_DIAG] single-wave workgroups: %u bytes of LDS each, %u per CU\n", lds_bytes, occ);
  if (occ < 1u) occ = 1u;
  /* persistent workgroups fill the device; a multiple of 8 keeps a workgroup's slots on its XCD */
  uint64_t grid = (uint64_t)(ctx->ncu / 8) * 8 * occ;
  /* A batch of few big blocks: leave a compute unit per block (one per XCD at least) to the planner workgroups of the
   * decode that follows or runs beside this launch on the context's planner stream (decode_device) -- a planner
   * workgroup needs a whole CU's LDS, and the persistent workgroups of this launch would otherwise hold every CU until
   * they are all done.  ~3 % of the solve's throughput for 8 blocks; the planner (one workgroup per block, latency
   * bound: 12 ms at K=27000, 36 ms at K'=56403) then hides behind the encode solve. */
  if (!small && occ == 1u) {
    /* (with planner runs issued ahead -- nrq_decode_plan_ahead -- up to `ahead_hint` batches' planner workgroups are resident at
     * once, and the encode plan of a big block is built by one more workgroup on a stream of its own: without a compute unit
     * for each of them one waits until this launch's persistent workgroups are through, and its 20-30 ms start from there) */
    const uint32_t runs = ctx->ahead_hint > 1u ? ctx->ahead_hint : 1u;
    uint32_t reserve = ctx->tune.reserve_cus >= 0 ? (uint32_t)ctx->tune.reserve_cus : (nblk <= 16u ? (nblk * runs + 1u + 7u) / 8u * 8u : 0u);
    if (reserve + 64u <= grid) grid -= reserve / 8u * 8u;
  }
  if (ctx->tune.solve_grid) grid = ctx->tune.solve_grid / 8 * 8;
  if (grid < 8) grid = 8;
  /* work slots (nrq_map_group): `sub` strips of a block each -- the strips of a whole line unless that would leave
   * workgroups idle -- incl. the empty slots of a partial block octet */
  uint32_t lsub = 0;
  while ((1u << lsub) < spl) lsub++;
  auto slots_for = [&](uint32_t ls) -> uint64_t {
    const uint32_t sub = 1u << ls, spb = (nstrips + sub - 1) / sub;
    return by_block ? (uint64_t)((nblk + 7u) / 8u) * 8u * spb : (uint64_t)nblk * spb;
  };
  while (lsub > 0 && slots_for(lsub) < grid) lsub--;
  {
    /* Work is dealt statically (workgroup g takes slots g, g + grid, ...): with few big blocks the rounds do not come
     * out even -- K'=56403, 8 blocks: 320 whole-line slots on 256 workgroups is two rounds for a quarter of them, 32
     * strips against 20 on average.  Smaller slots even that out; what they cost is gather efficiency (pieces shorter
     * than a 128-byte line), which matters for wide strips only: a 2- or 4-byte strip is solved at the same cost per
     * strip as a 16-byte one, so its data movement is an eighth or a quarter of the time share. */
    auto max_strips = [&](uint32_t ls) -> uint64_t {
      const uint64_t ns = slots_for(ls), g = grid < ns ? grid : ns;
      return ((ns + g - 1) / g) << ls;
    };
    const uint32_t min_ls = WB >= 8 ? (lsub < 2u ? lsub : 2u) : 0u;
    uint32_t best = lsub;
    for (uint32_t ls = lsub; ls-- > min_ls;)
      if (max_strips(ls) * 100u < max_strips(best) * 93u) best = ls;
    if (!ctx->tune.no_balance) lsub = best;
  }
  const uint64_t nslots = slots_for(lsub);
  if (nslots > 0x7FFFFFFFull) return fail(ctx, -4, "grid too large");
  if (grid > nslots) grid = by_block ? (nslots + 7) / 8 * 8 : nslots;
  /* per workgroup: two sets of `spl` input staging buffers (the line group being solved, the one being gathered)
   * and two sets of `spl` output staging buffers (the group being solved, the one being scattered) */
  const uint32_t stage_stride = (max_slots * WBE + 255u) & ~255u, ostage_stride = (max_out * WBE + 255u) & ~255u;
  {
    int rc_ = ensure_dev(ctx, ctx->stage, (size_t)grid * 2u * spl * ((size_t)stage_stride + ostage_stride));
    if (rc_) return rc_;
  }
  size_t ybuf_stride = 0;
  uint8_t *ybuf = nullptr;
  if (split) {
    ybuf_stride = ((size_t)(max_slots + max_u) * T + 255u) & ~(size_t)255u;
    int rc_ = ensure_dev(ctx, ctx->ybuf, (size_t)nblk * ybuf_stride);
    if (rc_) return rc_;
    ybuf = ctx->ybuf.p;
  }
  if (!ctx->attr_set[slot]) {
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_backsub_kernel<32>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_backsub_kernel<16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
#define NRQ_SET_LDS_ATTR(...)                                                                                                              \
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_solve_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)NRQ_LDS_MAX))
    NRQ_SET_LDS_ATTR(WB, NRQ_WG, 1, 1, false); NRQ_SET_LDS_ATTR(WB, NRQ_WG, 1, 1, true);
    if constexpr (WB != 12) {
    NRQ_SET_LDS_ATTR(WB, 256, NRQ_SMALL_WV, 1, false);    NRQ_SET_LDS_ATTR(WB, 256, NRQ_SMALL_WV, 1, true);
    NRQ_SET_LDS_ATTR(WB, 256, 5, 1, false);    NRQ_SET_LDS_ATTR(WB, 256, 5, 1, true);
    NRQ_SET_LDS_ATTR(WB, 64, NRQ_TINY_WV, 1, false);     NRQ_SET_LDS_ATTR(WB, 64, NRQ_TINY_WV, 1, true);
    }
#undef NRQ_SET_LDS_ATTR
    if constexpr (WB == 16) {
      HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_solve_kernel<16, 256, 4, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
      HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_solve_kernel<16, 256, 4, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
      HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nrq_solve_kernel<16, 256, 4, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)NRQ_LDS_MAX));
    }
    ctx->attr_set[slot] = true;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (ctx->ktime_on && !ctx->ktime_outer) { /* (ktime_outer: pick_and_launch brackets the launches of both block lists itself) */
    if (ctx->ktime_used == ctx->ktime_pool.size()) {
      hipEvent_t a, b;
      HIPCHK(ctx, hipEventCreate(&a));
      HIPCHK(ctx, hipEventCreate(&b));
      ctx->ktime_pool.emplace_back(a, b);
    }
    ev0 = ctx->ktime_pool[ctx->ktime_used].first;
    ev1 = ctx->ktime_pool[ctx->ktime_used].second;
    ctx->ktime_used++;
    HIPCHK(ctx, hipEventRecord(ev0, ctx->stream));
  }
  const uint32_t nprof = (uint32_t)((grid + 15) / 16);
  if (ctx->tune.prof) {
    if (ctx->prof) { (void)hipFree(ctx->prof); ctx->prof = nullptr; }
    HIPCHK(ctx, hipMalloc((void **)&ctx->prof, (size_t)nprof * 16 * 8));
    HIPCHK(ctx, hipMemsetAsync(ctx->prof, 0, (size_t)nprof * 16 * 8, ctx->stream));
  }
#define NRQ_LAUNCH_WIDE(GG)                                                                                                          \
  hipLaunchKernelGGL((nrq_solve_kernel<16, 256, 4, GG>), dim3((uint32_t)grid), dim3(256), lds_bytes, ctx->stream, d_jobs, nblk, T, nstrips, \
                     by_block ? 1u : 0u, (uint32_t)nslots, lsub, d_kc, (uint8_t *)ctx->stage.p, stage_stride, ostage_stride, ctx->prof,     \
                     ybuf, ybuf_stride)
  if (WB == 16 && G == 8) { if constexpr (WB == 16) NRQ_LAUNCH_WIDE(8); }
  else if (WB == 16 && G == 4) { if constexpr (WB == 16) NRQ_LAUNCH_WIDE(4); }
  else if (WB == 16 && G == 2) { if constexpr (WB == 16) NRQ_LAUNCH_WIDE(2); }
  else {
    /* (the movers' aligned-only form; 12-byte strips: whole dwords, solve_body.h g_get_al12) */
    const bool al = ctx->io_aligned && G == 1 && WB >= 4 && T % (uint32_t)(WB == 12 ? 4 : WB) == 0u;
#define NRQ_LAUNCH(NTT, WVV, ALL)                                                                                                        \
  hipLaunchKernelGGL((nrq_solve_kernel<WB, NTT, WVV, 1, ALL>), dim3((uint32_t)grid), dim3(NTT), lds_bytes, ctx->stream, d_jobs, nblk, T, nstrips, \
                     by_block ? 1u : 0u, (uint32_t)nslots, lsub, d_kc, (uint8_t *)ctx->stage.p, stage_stride, ostage_stride, ctx->prof, ybuf,   \
                     ybuf_stride)
    if constexpr (WB == 12) { if (al) NRQ_LAUNCH(NRQ_WG, 1, true); else NRQ_LAUNCH(NRQ_WG, 1, false); }
    else if (tiny) { if (al) NRQ_LAUNCH(64, NRQ_TINY_WV, true); else NRQ_LAUNCH(64, NRQ_TINY_WV, false); }
    else if (five) { if (al) NRQ_LAUNCH(256, 5, true); else NRQ_LAUNCH(256, 5, false); }
    else if (small) { if (al) NRQ_LAUNCH(256, NRQ_SMALL_WV, true); else NRQ_LAUNCH(256, NRQ_SMALL_WV, false); }
    else { if (al) NRQ_LAUNCH(NRQ_WG, 1, true); else NRQ_LAUNCH(NRQ_WG, 1, false); }
#undef NRQ_LAUNCH
    ctx->stats.movers_aligned = al ? 1u : 0u;
  }
  HIPCHK(ctx, hipGetLastError());
  if (split) {
    /* 32-byte strips while the tables (4 KiB per W word) leave room for two workgroups per CU, 16-byte strips beyond */
    const bool wide = max_wpr <= 20u;
    const uint32_t sb = wide ? 32u : 16u, nsb = (T + sb - 1u) / sb, tbl = max_wpr * 8u * 16u * sb;
    if (tbl > NRQ_LDS_MAX) return fail(ctx, -5, "back-substitution tables do not fit the LDS (wpr=%u)", max_wpr);
    uint32_t nchunks = (2048u + nsb * nblk - 1u) / (nsb * nblk);
    if (nchunks < 1u) nchunks = 1u;
    if (nchunks > 16u) nchunks = 16u;
    if (wide)
      hipLaunchKernelGGL(nrq_backsub_kernel<32>, dim3(nsb, nchunks, nblk), dim3(256), tbl, ctx->stream, d_jobs, T, ybuf, ybuf_stride, nchunks);
    else
      hipLaunchKernelGGL(nrq_backsub_kernel<16>, dim3(nsb, nchunks, nblk), dim3(256), tbl, ctx->stream, d_jobs, T, ybuf, ybuf_stride, nchunks);
    HIPCHK(ctx, hipGetLastError());
    if (res_elems) {
      hipLaunchKernelGGL(nrq_collect_kernel, dim3(res_elems, nblk), dim3(256), 0, ctx->stream, d_jobs, T, (const uint8_t *)ybuf, ybuf_stride);
      HIPCHK(ctx, hipGetLastError());
    }
  }
  if (ev1) HIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
  if (ctx->prof) {
    std::vector<unsigned long long> hp((size_t)nprof * 16);
    HIPCHK(ctx, hipMemcpyAsync(hp.data(), ctx->prof, hp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    static const char *names[8] = {"load", "fwd", "hdpc", "bin", "dense", "tables", "backsub", "store"};
    double sum[8] = {0}, tot = 0;
    uint32_t cnt = 0;
    for (uint32_t w = 0; w < nprof; w++) {
      const unsigned long long *q = &hp[(size_t)w * 16];
      if (!q[8]) continue;
      for (int k = 0; k < 8; k++) sum[k] += (double)(q[k + 1] - q[k]);
      tot += (double)(q[8] - q[0]);
      cnt++;
    }
    fprintf(stderr, "[NRQ_PROF] WB=%d grid=%llu sampled=%u total=%.0f clk:", WB, (unsigned long long)grid, cnt,
            cnt ? tot / cnt : 0.0);
    for (int k = 0; k < 8; k++) fprintf(stderr, " %s=%.0f", names[k], cnt ? sum[k] / cnt : 0.0);
    /* slots 9..15: free-form marks a phase may leave through StripCtx::dbg (differences to the phase start) */
    double ext[7] = {0};
    for (uint32_t w = 0; w < nprof; w++) {
      const unsigned long long *q = &hp[(size_t)w * 16];
      if (!q[8]) continue;
      for (int k = 0; k < 7; k++) if (q[9 + k]) ext[k] += (double)(q[9 + k] - q[ctx->tune.prof_base]);
    }
    fprintf(stderr, " | marks since fwd end:");
    for (int k = 0; k < 7; k++) fprintf(stderr, " %.0f", cnt ? ext[k] / cnt : 0.0);
    fprintf(stderr, "\n");
    (void)hipFree(ctx->prof);
    ctx->prof = nullptr;
  }
  ctx->stats.strip_bytes = WBE;
  ctx->stats.lds_bytes = lds_bytes;
  ctx->stats.grid = (uint32_t)grid;
  ctx->stats.wg_threads = nt;
  ctx->stats.strips_per_slot = 1u << lsub;
  ctx->stats.wg_waves_per_simd = tiny ? (uint32_t)NRQ_TINY_WV : five ? 5u : small ? (uint32_t)NRQ_SMALL_WV : 1u;
  return 0;
}

/* every row of a base + b * stride array of T-byte rows starts 16-byte aligned */
inline bool rows_aligned(const void *base, size_t stride, uint32_t T) {
  return ((reinterpret_cast<uintptr_t>(base) | (uintptr_t)stride | (uintptr_t)T) & 15u) == 0;
}
inline bool vec_aligned(const uint64_t *v, uint32_t n, uint32_t T) {
  if (T & 15u) return false;
  for (uint32_t b = 0; b < n; b++)
    if (v[b] & 15u) return false;
  return true;
}

/* widest strip whose LDS image fits for every plan header in hdrs */
static int launch_list(nrq_ctx *ctx, const std::vector<const nrq_plan_hdr *> &hdrs, const nrq_job *d_jobs, uint32_t nblk,
                       uint32_t T, const uint8_t *d_kc, uint32_t max_out, uint32_t wb, uint32_t need) {
  uint32_t max_slots = 0, max_u = 0, max_wpr = 0;
  for (const nrq_plan_hdr *h : hdrs) {
    if (h->status) continue;
    if (h->M > max_slots) max_slots = h->M;
    if (h->u > max_u) max_u = h->u;
    if (h->wpr > max_wpr) max_wpr = h->wpr;
  }
  switch (wb) {
    case 16: return launch_wb<16>(ctx, 0, d_jobs, nblk, T, d_kc, need, max_slots, max_out, max_u, max_wpr, hdrs);
    case 12: return launch_wb<12>(ctx, 4, d_jobs, nblk, T, d_kc, need, max_slots, max_out, max_u, max_wpr, hdrs);
    case 8: return launch_wb<8>(ctx, 1, d_jobs, nblk, T, d_kc, need, max_slots, max_out, max_u, max_wpr, hdrs);
    case 4: return launch_wb<4>(ctx, 2, d_jobs, nblk, T, d_kc, need, max_slots, max_out, max_u, max_wpr, hdrs);
    default: return launch_wb<2>(ctx, 3, d_jobs, nblk, T, d_kc, need, max_slots, max_out, max_u, max_wpr, hdrs);
  }
}
/* widest width at which the image of h fits the LDS (0: none) and its size there */
static uint32_t widest_fit(const nrq_ctx *ctx, const nrq_plan_hdr *h, uint32_t *need) {
  /* (12 bytes: between the 16-byte image's limit, K ~ 8500, and ~12000; below, a block that does not fit 16 bytes is the odd one
   * of its batch -- a decode plan with many inactive columns -- and goes on the second list at 12 as well) */
  static const uint32_t widths[5] = {16, 12, 8, 4, 2};
  for (int s = 0; s < 5; s++) {
    if (widths[s] > ctx->tune.max_wb) continue;
    if (widths[s] == 12u && ctx->tune.no_wb12) continue;
    const uint32_t t = nrq_lds_plan(h, widths[s]).total;
    if (t <= ctx->tune.lds_max) { *need = t; return widths[s]; }
  }
  *need = 0;
  return 0;
}

/* The solve launch(es) of a batch.  A launch runs at one strip width, and the widest width a block can have is set by ITS plan
 * (a decode plan's LDS image grows with its inactive columns): at K=8192 one block in a few thousand -- one launch in 40 at 10 %
 * loss, 4 in 40 at 30 % -- does not fit the 16-byte image.  Round 5 sent the whole launch to the width every block fits (8 bytes:
 * ~1.7 x the time for 256 blocks because of one); now the batch is split into at most TWO LISTS -- the blocks that fit the
 * widest width any block has, and the others, launched at the widest width THEY all fit; each list's job records are copied
 * side by side first (a handful of small device-to-device copies, only when a batch splits).
 * blk_of_hdr: index in d_jobs of every header (nullptr: all blocks share one plan, nothing to split). */
int pick_and_launch(nrq_ctx *ctx, const std::vector<const nrq_plan_hdr *> &hdrs, const nrq_job *d_jobs, uint32_t nblk,
                    uint32_t T, const uint8_t *d_kc, uint32_t max_out, const std::vector<uint32_t> *blk_of_hdr = nullptr) {
  uint32_t wa = 0, need_a = 0, wb_ = 16, need_b = 0, na = 0, nsolv = 0;
  std::vector<uint32_t> wd(hdrs.size(), 0), nd(hdrs.size(), 0);
  for (size_t i = 0; i < hdrs.size(); i++) {
    if (hdrs[i]->status) continue;
    nsolv++;
    wd[i] = widest_fit(ctx, hdrs[i], &nd[i]);
    if (!wd[i]) return fail(ctx, -5, "block too large for the LDS-resident solver");
    if (wd[i] > wa) wa = wd[i];
  }
  if (!nsolv) return 0; /* nothing solvable in this batch */
  for (size_t i = 0; i < hdrs.size(); i++) {
    if (hdrs[i]->status) continue;
    if (wd[i] == wa) { na++; if (nd[i] > need_a) need_a = nd[i]; }
    else if (wd[i] < wb_) wb_ = wd[i];
  }
  const uint32_t nb = nsolv - na;
  ctx->stats.strip_bytes_b = 0; ctx->stats.blocks_b = 0;
  /* one list: everybody fits the widest width -- or lists are off / impossible (no block indices) / not worth it (the wide list
   * would be the minority: then everybody runs at the narrow width, as before) */
  if (nb == 0 || !blk_of_hdr || ctx->tune.no_lists || na < nb) {
    const uint32_t w = nb == 0 ? wa : wb_;
    uint32_t need = 0;
    for (const nrq_plan_hdr *h : hdrs)
      if (!h->status) { const uint32_t t = nrq_lds_plan(h, w).total; if (t > need) need = t; }
    return launch_list(ctx, hdrs, d_jobs, nblk, T, d_kc, max_out, w, need);
  }
  /* two lists */
  std::vector<const nrq_plan_hdr *> ha, hb;
  std::vector<uint32_t> ib;
  for (size_t i = 0; i < hdrs.size(); i++) {
    if (hdrs[i]->status) continue;
    if (wd[i] == wa) ha.push_back(hdrs[i]);
    else { hb.push_back(hdrs[i]); ib.push_back((*blk_of_hdr)[i]); const uint32_t t = nrq_lds_plan(hdrs[i], wb_).total; if (t > need_b) need_b = t; }
  }
  /* both lists' job records side by side in scratch arrays: the second list's few records one by one (runs of neighbours in one
   * copy), the first list's as the stretches between them -- |B| + 1 copies of 80-byte records at most.  (A kernel-side rule --
   * "leave out the block whose image exceeds this launch's LDS" -- was tried first: the extra argument and header reads cost
   * the 256-thread variants, which live on 128 registers, 2.6 % at K=1000.) */
  int rc = ensure_dev(ctx, ctx->jobs_b, (ib.size() + (size_t)nblk) * sizeof(nrq_job));
  if (rc) return rc;
  nrq_job *jb = reinterpret_cast<nrq_job *>(ctx->jobs_b.p), *ja = jb + ib.size();
  for (size_t k = 0; k < ib.size(); k++) {
    size_t run = 1;
    while (k + run < ib.size() && ib[k + run] == ib[k] + run) run++;
    HIPCHK(ctx, hipMemcpyAsync(jb + k, d_jobs + ib[k], run * sizeof(nrq_job), hipMemcpyDeviceToDevice, ctx->stream));
    k += run - 1;
  }
  uint32_t na_blocks = 0; /* (every block of the batch that is not on the second list: the unsolvable ones stay, the kernel skips them) */
  for (size_t k = 0, from = 0; k <= ib.size(); k++) {
    const uint32_t to = k < ib.size() ? ib[k] : nblk;
    if (to > from) {
      HIPCHK(ctx, hipMemcpyAsync(ja + na_blocks, d_jobs + from, (to - from) * sizeof(nrq_job), hipMemcpyDeviceToDevice, ctx->stream));
      na_blocks += to - (uint32_t)from;
    }
    from = (size_t)to + 1u;
  }
  /* one pair of timing events around both launches (bench.py reads one interval per call) */
  hipEvent_t ev1 = nullptr;
  if (ctx->ktime_on) {
    if (ctx->ktime_used == ctx->ktime_pool.size()) {
      hipEvent_t a, b;
      HIPCHK(ctx, hipEventCreate(&a));
      HIPCHK(ctx, hipEventCreate(&b));
      ctx->ktime_pool.emplace_back(a, b);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ktime_pool[ctx->ktime_used].first, ctx->stream));
    ev1 = ctx->ktime_pool[ctx->ktime_used].second;
    ctx->ktime_used++;
    ctx->ktime_outer = true;
  }
  rc = launch_list(ctx, hb, jb, (uint32_t)ib.size(), T, d_kc, max_out, wb_, need_b);
  const uint32_t sb_b = ctx->stats.strip_bytes;
  if (!rc) rc = launch_list(ctx, ha, ja, na_blocks, T, d_kc, max_out, wa, need_a); /* (last: the call's stats describe the wide list) */
  ctx->ktime_outer = false;
  if (rc) return rc;
  if (ev1) HIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
  ctx->stats.strip_bytes_b = sb_b;
  ctx->stats.blocks_b = (uint32_t)ib.size();
  return 0;
}

} // namespace

/* ============================================================================================
 * C ABI
 * ========================================================================================== */
extern "C" {

int nrq_params(uint32_t K, uint32_t out[10]) {
  rq_params p;
  if (!rq_params_init(K, &p)) return -1;
  out[0] = p.Kp; out[1] = p.J; out[2] = p.S; out[3] = p.H; out[4] = p.W;
  out[5] = p.L; out[6] = p.P; out[7] = p.P1; out[8] = p.U; out[9] = p.B;
  return 0;
}

/* The runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues, 4 by default; a context runs up to six
 * streams beside the caller's, and two streams on one queue execute one after the other (seen as the encode-plan build
 * serialised into the solve stream).  The variable must be there before the runtime initialises, and it belongs to the HOST
 * process: the library does not touch the environment unless asked to -- NANORQ_HIP_SET_ENV=1 makes this load-time
 * constructor set GPU_MAX_HW_QUEUES=8 if it is unset (a constructor runs before the library's first HIP call and, for a
 * library linked at program start, before the process has other threads that could be reading the environment; a library
 * dlopen'ed late gets neither guarantee, which is why it is opt-in).  bench.py, the tools and the tests export the variable
 * themselves; include/nanorq.h and INTEGRATION.md tell a host process to. */
__attribute__((constructor)) static void nrq_env_at_load(void) {
  const char *e = getenv("NANORQ_HIP_SET_ENV");
  if (e && *e == '1') setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

int nrq_ctx_create(int device, void *stream, nrq_ctx **out) {
  if (!out) return -1;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -20; /* no HIP device: fail loudly, no fallback */
  if (device < 0 || device >= ndev) return -21;
  if (hipSetDevice(device) != hipSuccess) return -22;
  nrq_ctx *ctx = new nrq_ctx();
  ctx->device = device;
  ctx->stream = (hipStream_t)stream;
  {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) ctx->ncu = n;
  }
  ctx->tune.read();
  if (getenv("NRQ_HOST_PLANNER")) ctx->planner = 0;
  {
    const char *fi = getenv("NANORQ_HIP_FAULT_INJECT");
    ctx->fault_inject_armed = fi && *fi == '1';
  }
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  if (hipStreamCreateWithFlags(&ctx->plan_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->plan_stream_b, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->plan_stream2, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->aux[0], hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->aux[1], hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->aux[2], hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->scat_ev[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->scat_ev[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->planned[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->planned[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->planned[2], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->arena_free[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->arena_free[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->arena_free[2], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->pstaged[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->pstaged[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->pstaged[2], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->pstaged[3], hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&ctx->t0) != hipSuccess || hipEventCreate(&ctx->t1) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->encplan_uploaded, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->staged[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->staged[1], hipEventDisableTiming) != hipSuccess) {
    delete ctx;
    return -23;
  }
  *out = ctx;
  return 0;
}

void nrq_plan_cache_clear(nrq_ctx *ctx) {
  if (!ctx) return;
  for (auto &kv : ctx->encplans) kv.second.valid = false; /* buffers are reused by the rebuild */
}

static void encplans_release(nrq_ctx *ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->plan_stream2) (void)hipStreamSynchronize(ctx->plan_stream2);
  for (auto &kv : ctx->encplans) {
    for (int i = 0; i < 2; i++) {
      if (kv.second.devbuf[i]) (void)hipFree(kv.second.devbuf[i]);
      if (kv.second.used[i]) (void)hipEventDestroy(kv.second.used[i]);
    }
    if (kv.second.ready) (void)hipEventDestroy(kv.second.ready);
    if (kv.second.pin) (void)hipHostFree(kv.second.pin);
  }
  ctx->encplans.clear();
}

void nrq_ctx_destroy(nrq_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  encplans_release(ctx);
  if (ctx->encplan_uploaded) (void)hipEventDestroy(ctx->encplan_uploaded);
  for (auto &kv : ctx->kconst) {
    if (kv.second.dev) (void)hipFree(kv.second.dev);
    nrq_host_free(kv.second.host);
  }
  plan_ahead_drop(ctx);
  if (ctx->plan_stream) { (void)hipStreamSynchronize(ctx->plan_stream); (void)hipStreamDestroy(ctx->plan_stream); }
  if (ctx->plan_stream_b) { (void)hipStreamSynchronize(ctx->plan_stream_b); (void)hipStreamDestroy(ctx->plan_stream_b); }
  if (ctx->plan_stream2) { (void)hipStreamSynchronize(ctx->plan_stream2); (void)hipStreamDestroy(ctx->plan_stream2); }
  if (ctx->encplan_work.p) (void)hipFree(ctx->encplan_work.p);
  for (int i = 0; i < 2; i++) {
    if (ctx->aux[i]) { (void)hipStreamSynchronize(ctx->aux[i]); (void)hipStreamDestroy(ctx->aux[i]); }
    if (i == 0 && ctx->aux[2]) { (void)hipStreamSynchronize(ctx->aux[2]); (void)hipStreamDestroy(ctx->aux[2]); }
    if (ctx->scat_dev[i].p) (void)hipFree(ctx->scat_dev[i].p);
    if (ctx->scat_pin[i].p) (void)hipHostFree(ctx->scat_pin[i].p);
    if (ctx->scat_ev[i]) (void)hipEventDestroy(ctx->scat_ev[i]);
  }
  for (auto &kv : ctx->pool_size) (void)hipFree(kv.first);
  ctx->pool_size.clear(); ctx->pool_free.clear();
  for (int i = 0; i < 3; i++) {
    if (ctx->planned[i]) (void)hipEventDestroy(ctx->planned[i]);
    if (ctx->arena_free[i]) (void)hipEventDestroy(ctx->arena_free[i]);
    if (ctx->plan_arena[i].p) (void)hipFree(ctx->plan_arena[i].p);
    if (ctx->plan_jobs[i].p) (void)hipFree(ctx->plan_jobs[i].p);
  }
  for (int i = 0; i < 2; i++)
    if (ctx->plan_work[i].p) (void)hipFree(ctx->plan_work[i].p);
  for (int i = 2; i < 4; i++) {
    if (ctx->pscratch[i].p) (void)hipFree(ctx->pscratch[i].p);
    if (ctx->pstaging[i].p) (void)hipHostFree(ctx->pstaging[i].p);
    if (ctx->pstaged[i]) (void)hipEventDestroy(ctx->pstaged[i]);
  }
  if (ctx->stage.p) (void)hipFree(ctx->stage.p);
  if (ctx->ybuf.p) (void)hipFree(ctx->ybuf.p);
  for (int i = 0; i < 2; i++) {
    if (ctx->pscratch[i].p) (void)hipFree(ctx->pscratch[i].p);
    if (ctx->pstaging[i].p) (void)hipHostFree(ctx->pstaging[i].p);
    if (ctx->pstaged[i]) (void)hipEventDestroy(ctx->pstaged[i]);
    if (ctx->scratch[i].p) (void)hipFree(ctx->scratch[i].p);
    if (ctx->staging[i].p) (void)hipHostFree(ctx->staging[i].p);
    if (ctx->staged[i]) (void)hipEventDestroy(ctx->staged[i]);
  }
  for (auto &pr : ctx->ktime_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto &pr : ctx->ptime_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (ctx->ktime_base) (void)hipEventDestroy(ctx->ktime_base);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  delete ctx;
}

int nrq_ctx_set_stream(nrq_ctx *ctx, void *stream) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = (hipStream_t)stream;
  return 0;
}

const char *nrq_ctx_error(nrq_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int nrq_ctx_sync(nrq_ctx *ctx) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

void nrq_ctx_last_stats(nrq_ctx *ctx, nrq_call_stats *out) {
  if (ctx && out) *out = ctx->stats;
}

int nrq_ctx_set_planner(nrq_ctx *ctx, int device_planner) {
  if (!ctx) return -1;
  ctx->planner = device_planner ? 1 : 0;
  return 0;
}

int nrq_ctx_set_option(nrq_ctx *ctx, const char *name, long long value) {
  if (!ctx || !name) return -1;
  Tuning &t = ctx->tune;
  const std::string n(name);
  if (n == "max_wb") t.max_wb = (uint32_t)value;
  else if (n == "no_split") t.no_split = value != 0;
  else if (n == "no_tiny") t.no_tiny = value != 0;
  else if (n == "no_wentry") t.no_wentry = value != 0;
  else if (n == "wide_g") t.wide_g = (value == 2 || value == 4 || value == 8) ? (uint32_t)value : 0u;
  else if (n == "tiny_div") t.tiny_div = (uint32_t)value;
  else if (n == "tiny_div_dec") t.tiny_div_dec = (uint32_t)value;
  else if (n == "no_balance") t.no_balance = value != 0;
  else if (n == "no_lists") t.no_lists = value != 0;
  else if (n == "no_wb12") t.no_wb12 = value != 0;
  else if (n == "host_plan_auto") t.host_plan_auto = value != 0;
  else if (n == "plan_pack") t.plan_pack = value != 0;
  else if (n == "tiny_any") t.tiny_any = value != 0;
  else if (n == "lds_max") t.lds_max = value > 0 && value <= (long long)NRQ_LDS_MAX ? (uint32_t)value : NRQ_LDS_MAX;
  else if (n == "no_plan_stream") t.no_plan_stream = value != 0;
  else if (n == "no_plan_split") t.no_plan_split = value != 0;
  else if (n == "plan_split_force") t.plan_split_force = value != 0;
  else if (n == "plan_small_state") t.plan_small_state = value != 0;
  else if (n == "plan_big_wg") t.plan_big_wg = value != 0;
  else if (n == "reserve_cus") t.reserve_cus = (int)value;
  else if (n == "solve_grid") t.solve_grid = (uint64_t)value;
  else if (n == "big_wg") t.big_wg = value != 0;
  else if (n == "small_waves4") t.small_waves4 = value != 0;
  else if (n == "map_spread") t.map_spread = value != 0;
  else if (n == "encplan_dev_min_l") t.encplan_dev_min_l = (uint32_t)value;
  else if (n == "plan_ucap") t.plan_ucap = (uint32_t)value;
  else if (n == "plan_wrong_instance") t.plan_wrong_instance = value != 0;
  else if (n == "plan_no_wg128") t.plan_no_wg128 = value != 0;
  else if (n == "fail_after" && ctx->fault_inject_armed) ctx->fail_after = value > 0 ? value : 0;
  else if (n == "faults_injected") return (int)ctx->faults_injected; /* (read: injected failures so far) */
  else return fail(ctx, -1, "unknown option %s", name);
  return 0;
}

int nrq_ctx_set_threads(nrq_ctx *ctx, int n) {
  if (!ctx || n < 0) return -1;
  ctx->threads = n;
  return 0;
}

int nrq_precalculate(nrq_ctx *ctx, uint32_t K, uint32_t Kp) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  EncPlan *ep;
  return get_encplan(ctx, K, Kp, &ep, /*finish=*/false); /* a device build is only enqueued; the encode that needs it waits */
}

int nrq_warm(nrq_ctx *ctx, uint32_t K, uint32_t Kp, int encode_plan) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = plan_attr_once(ctx);
  if (rc) return rc;
  rq_params p;
  if ((rc = block_params(ctx, K, Kp, &p))) return rc;
  KConst *kc;
  if ((rc = get_kconst(ctx, p.Kp, &kc))) return rc;
  if (encode_plan) {
    EncPlan *ep;
    rc = get_encplan(ctx, K, Kp, &ep, /*finish=*/encode_plan > 1); /* (2: also wait for a device build of the plan) */
  }
  return rc;
}

int nrq_encode_blocks(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_src, size_t src_stride,
                      void *d_inter, size_t inter_stride, uint32_t nrep, const uint32_t *h_esis, void *d_rep,
                      size_t rep_stride) {
  if (!ctx) return -1;
  if (!d_src || T == 0 || nblk == 0 || (nrep && (!h_esis || !d_rep))) return fail(ctx, -1, "bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const double t_begin = now_ms();
  rq_params p;
  int rc = block_params(ctx, K, Kp, &p);
  if (rc) return rc;
  for (uint32_t q = 0; q < nrep; q++)
    if (h_esis[q] < K || h_esis[q] >= (1u << 24)) return fail(ctx, -1, "repair ESI %u out of range", h_esis[q]);
  ctx->io_aligned = rows_aligned(d_src, src_stride, T) && (!nrep || rows_aligned(d_rep, rep_stride, T)) &&
                    (ctx->vec_inter ? vec_aligned(ctx->vec_inter, nblk, T) : (!d_inter || rows_aligned(d_inter, inter_stride, T)));
  KConst *kc;
  rc = get_kconst(ctx, p.Kp, &kc);
  if (rc) return rc;
  const auto cached_it = ctx->encplans.find(((uint64_t)p.Kp << 32) | K);
  const bool cached = cached_it != ctx->encplans.end() && cached_it->second.valid;
  EncPlan *ep;
  rc = get_encplan(ctx, K, p.Kp, &ep);
  if (rc) return rc;
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats.plan_ms = cached ? 0.0 : ep->build_ms;
  ctx->stats.plan_bytes = ep->plan_bytes;
  ctx->stats.xor_ops = (uint64_t)ep->hdr.n_xor_ops * nblk;
  ctx->stats.npiv = ep->hdr.npiv; ctx->stats.u = ep->hdr.u; ctx->stats.nlev = ep->hdr.nlev;
  ctx->stats.nfree = ep->hdr.nfree;

  /* per-call arrays: [jobs][isi->(cptr, cols, row)] */
  std::vector<uint32_t> isis(nrep), cptr;
  std::vector<uint16_t> cols;
  for (uint32_t q = 0; q < nrep; q++) isis[q] = h_esis[q] + (p.Kp - K);
  build_out_lists(p, ep->colslot.data(), nrep, isis.data(), cptr, cols);
  const size_t off_jobs = 0;
  const size_t off_cptr = r16(off_jobs + (size_t)nblk * sizeof(nrq_job));
  const size_t off_row = r16(off_cptr + (size_t)(nrep + 1) * 4);
  const size_t off_cols = r16(off_row + (size_t)(nrep ? nrep : 1) * 4);
  const size_t total = r16(off_cols + cols.size() * 2 + NRQ_STORE_SLACK); /* (ph_store reads a whole trip from a list's start) */
  const int f = ctx->flip;
  ctx->flip ^= 1;
  HIPCHK(ctx, hipEventSynchronize(ctx->staged[f]));
  if ((rc = ensure_pin(ctx, ctx->staging[f], total))) return rc;
  if ((rc = ensure_dev(ctx, ctx->scratch[f], total))) return rc;
  uint8_t *hs = ctx->staging[f].p, *ds = ctx->scratch[f].p;
  memcpy(hs + off_cptr, cptr.data(), (size_t)(nrep + 1) * 4);
  for (uint32_t q = 0; q < nrep; q++) reinterpret_cast<uint32_t *>(hs + off_row)[q] = q;
  if (!cols.empty()) memcpy(hs + off_cols, cols.data(), cols.size() * 2);
  nrq_job *jobs = reinterpret_cast<nrq_job *>(hs + off_jobs);
  for (uint32_t b = 0; b < nblk; b++) {
    nrq_job &j = jobs[b];
    memset(&j, 0, sizeof(j));
    j.plan = (uint64_t)(uintptr_t)ep->dev;
    j.rowsrc = (uint64_t)(uintptr_t)(ep->dev + ep->rowsrc_off);
    j.src = (uint64_t)(uintptr_t)((const uint8_t *)d_src + (size_t)b * src_stride);
    j.rep = 0;
    j.inter = ctx->vec_inter ? ctx->vec_inter[b] : d_inter ? (uint64_t)(uintptr_t)((uint8_t *)d_inter + (size_t)b * inter_stride) : 0;
    j.out = nrep ? (uint64_t)(uintptr_t)((uint8_t *)d_rep + (size_t)b * rep_stride) : 0;
    j.out_cptr = (uint64_t)(uintptr_t)(ds + off_cptr);
    j.out_slots = (uint64_t)(uintptr_t)(ds + off_cols);
    j.out_row = (uint64_t)(uintptr_t)(ds + off_row);
    j.nout = nrep;
  }
  HIPCHK(ctx, hipMemcpyAsync(ds, hs, total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->staged[f], ctx->stream));
  std::vector<const nrq_plan_hdr *> hdrs(1, &ep->hdr);
  rc = pick_and_launch(ctx, hdrs, reinterpret_cast<const nrq_job *>(ds + off_jobs), nblk, T, kc->dev, ((d_inter || ctx->vec_inter) ? p.L : 0u) + nrep);
  if (ep->used[ep->cur]) { /* (device-built plans: the next build may not overwrite this buffer before the launch is done) */
    HIPCHK(ctx, hipEventRecord(ep->used[ep->cur], ctx->stream));
    ep->used_set[ep->cur] = true;
  }
  ctx->stats.host_ms = now_ms() - t_begin;
  return rc;
}

static int decode_host(nrq_ctx *ctx, const uint8_t *select, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                      const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                      const uint32_t *h_nrep, const uint32_t *h_avail, uint32_t *h_used, uint32_t rep_cap, const void *d_rep, size_t rep_stride, void *d_inter,
                      size_t inter_stride, int *h_status) {
  const double t_begin = now_ms();
  rq_params p;
  int rc = block_params(ctx, K, Kp, &p);
  if (rc) return rc;
  KConst *kc;
  rc = get_kconst(ctx, p.Kp, &kc);
  if (rc) return rc;
  if (!select) memset(&ctx->stats, 0, sizeof(ctx->stats));

  struct Prep {
    uint8_t *plan = nullptr;
    uint32_t plan_bytes = 0;
    std::vector<uint32_t> rowsrc, cptr, orow;
    std::vector<uint16_t> cols;
    int state = 0; /* 0 = nothing to do, 1 = solve, -1 = cannot */
    uint32_t used = 0;
    size_t off_plan = 0, off_rowsrc = 0, off_cptr = 0, off_row = 0, off_cols = 0;
  };
  std::vector<Prep> prep(nblk);
  struct PrepGuard { /* the host-built plan arenas are released on every way out of this function */
    std::vector<Prep> &v;
    ~PrepGuard() { for (auto &pr : v) if (pr.plan) { nrq_host_free(pr.plan); pr.plan = nullptr; } }
  } prep_guard{prep};
  const uint32_t pad = p.Kp - K;

  auto prepare = [&](uint32_t b) {
    Prep &pr = prep[b];
    const uint32_t nl = h_nlost[b], nr = h_nrep[b];
    if (select && !select[b]) { pr.state = 2; return; } /* not ours: leave its status alone */
    if (nl == 0) { pr.state = 0; return; }             /* nothing missing (nanorq.c:605-606) */
    if (nr < nl || nl > lost_cap || nr > rep_cap) { pr.state = -1; return; } /* nanorq.c:607-608 */
    const uint32_t *lost = h_lost + (size_t)b * lost_cap;
    const uint32_t *resi = h_rep_esi + (size_t)b * rep_cap;
    uint32_t avail = h_avail ? h_avail[b] : nr;
    if (avail < nr) avail = nr;
    if (avail > rep_cap) avail = rep_cap;
    /* use nr symbols; while the system is rank deficient and the caller holds more, take one more and re-plan */
    for (uint32_t use = nr;; use++) {
      const uint32_t overhead = use - nl;
      const uint32_t M = p.L + overhead;
      if (M > 65535u) { pr.state = -1; return; }
      std::vector<uint32_t> isis(p.Kp + overhead);
      for (uint32_t j = 0; j < p.Kp; j++) isis[j] = j;
      pr.rowsrc.assign(M, NRQ_ROW_ZERO);
      for (uint32_t j = 0; j < K; j++) pr.rowsrc[p.S + p.H + j] = j;
      for (uint32_t g = 0; g < nl; g++) {
        if (lost[g] >= K || (g && lost[g] <= lost[g - 1]) || resi[g] < K || resi[g] >= (1u << 24)) { pr.state = -1; return; }
        isis[lost[g]] = resi[g] + pad;
        pr.rowsrc[p.S + p.H + lost[g]] = NRQ_ROW_REP | g;
      }
      for (uint32_t e = 0; e < overhead; e++) {
        if (resi[nl + e] < K || resi[nl + e] >= (1u << 24)) { pr.state = -1; return; }
        isis[p.Kp + e] = resi[nl + e] + pad;
        pr.rowsrc[p.L + e] = NRQ_ROW_REP | (nl + e);
      }
      if (pr.plan) { nrq_host_free(pr.plan); pr.plan = nullptr; }
      if (nrq_host_plan_build(p.Kp, p.Kp + overhead, isis.data(), kc->host, &pr.plan, &pr.plan_bytes) != 0) {
        pr.state = -1;
        return;
      }
      if (reinterpret_cast<const nrq_plan_hdr *>(pr.plan)->status == 0) { pr.used = use; break; }
      if (use + 1 > avail) { pr.state = -1; return; } /* rank(A) < L with everything the caller holds */
    }
    build_out_lists(p, reinterpret_cast<const uint16_t *>(pr.plan + reinterpret_cast<const nrq_plan_hdr *>(pr.plan)->off_colslot),
                    nl, lost, pr.cptr, pr.cols); /* ISI of a source symbol is its ESI */
    pr.orow.assign(lost, lost + nl);
    pr.state = 1;
  };

  const double t_plan0 = now_ms();
  {
    uint32_t nth = ctx->threads > 0 ? (uint32_t)ctx->threads : std::thread::hardware_concurrency();
    if (nth == 0) nth = 1;
    if (nth > nblk) nth = nblk;
    if ((uint64_t)K * nblk < 920u) nth = 1; /* (a few small blocks: ~0.4 us x K each -- less than starting and joining threads, ~200 us) */
    std::atomic<uint32_t> next(0);
    auto worker = [&]() {
      for (;;) {
        uint32_t b = next.fetch_add(1);
        if (b >= nblk) break;
        prepare(b);
      }
    };
    if (nth <= 1) worker();
    else {
      std::vector<std::thread> pool;
      for (uint32_t t = 0; t < nth; t++) pool.emplace_back(worker);
      for (auto &t : pool) t.join();
    }
  }
  ctx->stats.plan_ms += now_ms() - t_plan0;

  /* pack everything the kernel reads into one staging image */
  size_t off = r16((size_t)nblk * sizeof(nrq_job));
  /* a shared dummy header (status=1) for blocks that need no launch work */
  const size_t off_dummy = off;
  off = r16(off + sizeof(nrq_plan_hdr));
  uint32_t nsolve = 0;
  for (uint32_t b = 0; b < nblk; b++) {
    Prep &pr = prep[b];
    if (pr.state != 2) { h_status[b] = pr.state >= 0 ? 1 : 0; if (h_used) h_used[b] = pr.state == 1 ? pr.used : 0; }
    if (pr.state != 1) continue;
    nsolve++;
    pr.off_plan = off;   off = r16(off + pr.plan_bytes);
    pr.off_rowsrc = off; off = r16(off + pr.rowsrc.size() * 4);
    pr.off_cptr = off;   off = r16(off + pr.cptr.size() * 4);
    pr.off_row = off;    off = r16(off + pr.orow.size() * 4);
    pr.off_cols = off;   off = r16(off + pr.cols.size() * 2 + NRQ_STORE_SLACK);
  }
  const size_t total = off;
  int result = 0;
  if (nsolve) {
    const int f = ctx->flip;
    ctx->flip ^= 1;
    HIPCHK(ctx, hipEventSynchronize(ctx->staged[f]));
    if ((rc = ensure_pin(ctx, ctx->staging[f], total))) return rc;
    if ((rc = ensure_dev(ctx, ctx->scratch[f], total))) return rc;
    uint8_t *hs = ctx->staging[f].p, *ds = ctx->scratch[f].p;
    nrq_plan_hdr dummy;
    memset(&dummy, 0, sizeof(dummy));
    dummy.magic = NRQ_PLAN_MAGIC;
    dummy.status = 1;
    memcpy(hs + off_dummy, &dummy, sizeof(dummy));
    nrq_job *jobs = reinterpret_cast<nrq_job *>(hs);
    std::vector<const nrq_plan_hdr *> hdrs;
    std::vector<uint32_t> hblk;
    for (uint32_t b = 0; b < nblk; b++) {
      Prep &pr = prep[b];
      nrq_job &j = jobs[b];
      memset(&j, 0, sizeof(j));
      if (pr.state != 1) { j.plan = (uint64_t)(uintptr_t)(ds + off_dummy); continue; }
      hblk.push_back(b);
      memcpy(hs + pr.off_plan, pr.plan, pr.plan_bytes);
      memcpy(hs + pr.off_rowsrc, pr.rowsrc.data(), pr.rowsrc.size() * 4);
      memcpy(hs + pr.off_cptr, pr.cptr.data(), pr.cptr.size() * 4);
      memcpy(hs + pr.off_row, pr.orow.data(), pr.orow.size() * 4);
      if (!pr.cols.empty()) memcpy(hs + pr.off_cols, pr.cols.data(), pr.cols.size() * 2);
      const nrq_plan_hdr *h = reinterpret_cast<const nrq_plan_hdr *>(hs + pr.off_plan);
      hdrs.push_back(h);
      if (ctx->stats.npiv == 0) {
        ctx->stats.npiv = h->npiv; ctx->stats.u = h->u; ctx->stats.nlev = h->nlev; ctx->stats.nfree = h->nfree;
      }
      ctx->stats.xor_ops += h->n_xor_ops;
      ctx->stats.plan_bytes += pr.plan_bytes;
      j.plan = (uint64_t)(uintptr_t)(ds + pr.off_plan);
      j.rowsrc = (uint64_t)(uintptr_t)(ds + pr.off_rowsrc);
      j.src = (ctx->vec_src ? ctx->vec_src[b] : (uint64_t)(uintptr_t)((uint8_t *)d_src + (size_t)b * src_stride));
      j.rep = (ctx->vec_rep ? ctx->vec_rep[b] : (uint64_t)(uintptr_t)((const uint8_t *)d_rep + (size_t)b * rep_stride));
      j.inter = d_inter ? (uint64_t)(uintptr_t)((uint8_t *)d_inter + (size_t)b * inter_stride) : 0;
      j.out = j.src; /* recovered symbols go back into the block's own rows */
      j.out_cptr = (uint64_t)(uintptr_t)(ds + pr.off_cptr);
      j.out_slots = (uint64_t)(uintptr_t)(ds + pr.off_cols);
      j.out_row = (uint64_t)(uintptr_t)(ds + pr.off_row);
      j.nout = (uint32_t)pr.orow.size();
    }
    hipError_t e = hipMemcpyAsync(ds, hs, total, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->staged[f], ctx->stream);
    if (e != hipSuccess) result = fail(ctx, -10, "staging copy failed: %s", hipGetErrorString(e));
    else {
      uint32_t max_out = 0;
      for (uint32_t b = 0; b < nblk; b++)
        if (prep[b].state == 1 && prep[b].orow.size() > max_out) max_out = (uint32_t)prep[b].orow.size();
      result = pick_and_launch(ctx, hdrs, reinterpret_cast<const nrq_job *>(ds), nblk, T, kc->dev, (d_inter ? p.L : 0u) + max_out, &hblk);
    }
  }
  ctx->stats.host_ms += now_ms() - t_begin;
  return result;
}


/* One planner run of a batch of decode blocks: what its launch leaves for the half that waits for it and launches the solve.
 * Kept in the context between nrq_decode_plan_ahead and the decode call it was issued for (`key_*`: that call's arguments). */
struct PlanRun {
  rq_params p;
  KConst *kc = nullptr;
  uint32_t K = 0, Kp = 0, T = 0, nblk = 0, lost_cap = 0, rep_cap = 0;
  uint32_t ucap = 0, Mcap = 0, npcap = 0, arena_cap = 0, max_nl = 0;
  size_t off_hdrs = 0;
  int f = 0, ab = 0; /* staging set, arena set */
  hipStream_t ps = nullptr;
  double t_begin = 0;
  /* the call this run was issued for */
  const void *d_src = nullptr, *d_rep = nullptr, *d_inter = nullptr;
  size_t src_stride = 0, rep_stride = 0, inter_stride = 0;
  bool has_avail = false;
  std::vector<uint32_t> lost, nlost, resi, nrep, avail;
};

static void plan_ahead_drop(nrq_ctx *ctx) {
  for (PlanRun *r : ctx->ahead) {
    (void)hipEventSynchronize(ctx->planned[r->ab]); /* (its arena set is free again once it has run) */
    delete r;
  }
  ctx->ahead.clear();
  ctx->ahead_hint = 0;
}

/* first half: inputs down, planner kernels and the headers' way back enqueued on the planner stream */
static int plan_launch(nrq_ctx *ctx, PlanRun &r, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                       const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                       const uint32_t *h_nrep, uint32_t rep_cap, const void *d_rep, size_t rep_stride, void *d_inter,
                       size_t inter_stride, const uint32_t *h_avail) {
  r.t_begin = now_ms();
  rq_params &p = r.p;
  int rc = block_params(ctx, K, Kp, &p);
  if (rc) return rc;
  KConst *kc;
  rc = get_kconst(ctx, p.Kp, &kc);
  if (rc) return rc;
  r.kc = kc;
  r.K = K; r.Kp = Kp; r.T = T; r.nblk = nblk; r.lost_cap = lost_cap; r.rep_cap = rep_cap;
  const nrq_kconst_hdr *kh = reinterpret_cast<const nrq_kconst_hdr *>(kc->host);
  uint32_t max_oh = 0, max_nrep = 0, max_nl = 0;
  for (uint32_t b = 0; b < nblk; b++) {
    const uint32_t nl = h_nlost[b];
    uint32_t nr = h_nrep[b];
    if (nl == 0 || nr < nl || nl > lost_cap || nr > rep_cap) continue;
    if (h_avail && h_avail[b] > nr) nr = h_avail[b] < rep_cap ? h_avail[b] : rep_cap; /* sizing: everything it may use */
    if (nr - nl > max_oh) max_oh = nr - nl;
    if (nr > max_nrep) max_nrep = nr;
    if (nl > max_nl) max_nl = nl;
  }
  r.max_nl = max_nl;
  uint32_t ucap = p.P + 768u;
  if (ucap > 1280u) ucap = 1280u; /* 40 words per W row at most */
  if (ctx->tune.plan_ucap && ctx->tune.plan_ucap < ucap) ucap = ctx->tune.plan_ucap;
  if (ucap < p.P + 32u) return fail(ctx, -5, "K'=%u has too many permanently inactive columns for the device planner", p.Kp);
  const uint32_t Mcap = p.L + max_oh + PL_EXTRA_ROWS + 8u, npcap = max_nrep + PL_EXTRA_ROWS + 8u;
  const pl_work_layout wl = pl_work_plan(p.L, Mcap, npcap, ucap, kh->nnz + npcap * PL_PATCH_STRIDE);
  const uint32_t arena_cap = pl_arena_bound(p.L, Mcap, ucap, kh->nnz + npcap * PL_PATCH_STRIDE, max_nl + 8u);
  r.ucap = ucap; r.Mcap = Mcap; r.npcap = npcap; r.arena_cap = arena_cap;
  const int ab = ctx->aflip;
  ctx->aflip = (ctx->aflip + 1) % 3;
  r.ab = ab;
  const uint32_t run_no = ctx->prun++;
  DevBuf &work = ctx->plan_work[run_no & 1u];
  /* (growing a buffer frees the old one: nothing may still be running in it) */
  if (work.cap < (size_t)nblk * wl.total || ctx->plan_arena[ab].cap < (size_t)nblk * arena_cap ||
      ctx->plan_jobs[ab].cap < (size_t)nblk * sizeof(nrq_job)) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->plan_stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->plan_stream_b));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  if ((rc = ensure_dev(ctx, work, (size_t)nblk * wl.total))) return rc;
  if ((rc = ensure_dev(ctx, ctx->plan_arena[ab], (size_t)nblk * arena_cap))) return rc;
  if ((rc = ensure_dev(ctx, ctx->plan_jobs[ab], (size_t)nblk * sizeof(nrq_job)))) return rc;

  /* inputs of the planner: [planjobs][lost lists][repair ESI lists]; headers come back after them */
  const size_t off_pj = 0;
  const size_t off_lost = r16(off_pj + (size_t)nblk * sizeof(nrq_planjob));
  const size_t off_resi = r16(off_lost + (size_t)nblk * lost_cap * 4);
  const size_t in_bytes = r16(off_resi + (size_t)nblk * rep_cap * 4);
  const size_t off_hdrs = in_bytes;
  const size_t total = r16(off_hdrs + (size_t)nblk * sizeof(nrq_plan_hdr));
  r.off_hdrs = off_hdrs;
  /* everything up to the headers' way back runs on the planner stream (see nrq_ctx::plan_stream) */
  /* (only while the planner workgroups leave at least half of the CUs alone: with one per CU the two kernels just take
   * turns, and planner workgroups that get in first delay the persistent workgroups of the solve -- measured at 256
   * blocks of K=8192: encode solve 7.6 -> 10.4 ms, step +0.4 ms; at 64 blocks of K=20000 the overlap is worth 10 %) */
  hipStream_t ps = (ctx->tune.no_plan_stream || nblk * 2u > (uint32_t)ctx->ncu) ? ctx->stream
                   : (run_no & 1u)                                                 ? ctx->plan_stream_b
                                                                                   : ctx->plan_stream;
  r.ps = ps;
  const int f = ctx->pflip;
  ctx->pflip = (ctx->pflip + 1) & 3;
  r.f = f;
  HIPCHK(ctx, hipEventSynchronize(ctx->pstaged[f]));
  if ((rc = ensure_pin(ctx, ctx->pstaging[f], total))) return rc;
  if ((rc = ensure_dev(ctx, ctx->pscratch[f], total))) return rc; /* (inputs, and the headers side by side behind them) */
  uint8_t *hs = ctx->pstaging[f].p, *ds = ctx->pscratch[f].p;
  if (ctx->arena_busy[ab] && ps != ctx->stream) HIPCHK(ctx, hipStreamWaitEvent(ps, ctx->arena_free[ab], 0));
  memcpy(hs + off_lost, h_lost, (size_t)nblk * lost_cap * 4);
  memcpy(hs + off_resi, h_rep_esi, (size_t)nblk * rep_cap * 4);
  nrq_planjob *pj = reinterpret_cast<nrq_planjob *>(hs + off_pj);
  for (uint32_t b = 0; b < nblk; b++) {
    nrq_planjob &j = pj[b];
    memset(&j, 0, sizeof(j));
    const bool sane = h_nlost[b] <= lost_cap && h_nrep[b] <= rep_cap;
    j.lost = (uint64_t)(uintptr_t)(ds + off_lost + (size_t)b * lost_cap * 4);
    j.rep_esi = (uint64_t)(uintptr_t)(ds + off_resi + (size_t)b * rep_cap * 4);
    j.work = (uint64_t)(uintptr_t)(work.p + (size_t)b * wl.total);
    j.arena = (uint64_t)(uintptr_t)(ctx->plan_arena[ab].p + (size_t)b * arena_cap);
    j.hdr_out = (uint64_t)(uintptr_t)(ds + off_hdrs + (size_t)b * sizeof(nrq_plan_hdr));
    j.src = (ctx->vec_src ? ctx->vec_src[b] : (uint64_t)(uintptr_t)((uint8_t *)d_src + (size_t)b * src_stride));
    j.rep = (ctx->vec_rep ? ctx->vec_rep[b] : (uint64_t)(uintptr_t)((const uint8_t *)d_rep + (size_t)b * rep_stride));
    j.inter = d_inter ? (uint64_t)(uintptr_t)((uint8_t *)d_inter + (size_t)b * inter_stride) : 0;
    j.nlost = sane ? h_nlost[b] : 0;
    j.nrep = sane ? h_nrep[b] : 0;
    j.nrep_avail = j.nrep;
    if (sane && h_avail && h_avail[b] > j.nrep) j.nrep_avail = h_avail[b] < rep_cap ? h_avail[b] : rep_cap;
    j.arena_cap = arena_cap;
    j.mode = plan_is_segmented(ctx, p, Mcap) ? (ctx->tune.no_wentry ? 0x100u : 0x300u) : 0u; /* bit 8: segmented run, bit 9: entry pass by nrq_wentry_kernel */
  }
  { /* (in_bytes is a multiple of 16; both buffers are 16-byte aligned allocations) */
    const uint32_t n16 = (uint32_t)(in_bytes / 16u);
    uint32_t g = (n16 + 255u) / 256u;
    if (g > 64u) g = 64u;
    hipLaunchKernelGGL(nrq_ctl_copy_kernel, dim3(g ? g : 1u), dim3(256), 0, ps, reinterpret_cast<uint4 *>(ds), reinterpret_cast<const uint4 *>(hs), n16);
    HIPCHK(ctx, hipGetLastError());
  }
  unsigned long long *pprof = nullptr;
  if (ctx->tune.prof) {
    HIPCHK(ctx, hipMalloc((void **)&pprof, 64 * 8));
    HIPCHK(ctx, hipMemsetAsync(pprof, 0, 64 * 8, ps));
  }
  hipEvent_t pe0 = nullptr, pe1 = nullptr;
  if (ctx->ktime_on) {
    if (ctx->ptime_used == ctx->ptime_pool.size()) {
      hipEvent_t a, b;
      HIPCHK(ctx, hipEventCreate(&a));
      HIPCHK(ctx, hipEventCreate(&b));
      ctx->ptime_pool.emplace_back(a, b);
    }
    pe0 = ctx->ptime_pool[ctx->ptime_used].first;
    pe1 = ctx->ptime_pool[ctx->ptime_used].second;
    ctx->ptime_used++;
    HIPCHK(ctx, hipEventRecord(pe0, ps));
  }
  if ((rc = launch_plan_kernel(ctx, ps, p, kc->dev, reinterpret_cast<const nrq_planjob *>(ds + off_pj),
                               reinterpret_cast<nrq_job *>(ctx->plan_jobs[ab].p), nblk, Mcap, npcap, ucap, pprof,
                               kh->nnz + npcap * PL_PATCH_STRIDE)))
    return rc;
  if (pe1) HIPCHK(ctx, hipEventRecord(pe1, ps));
  if (pprof) {
    unsigned long long hp[64];
    HIPCHK(ctx, hipMemcpyAsync(hp, pprof, sizeof(hp), hipMemcpyDeviceToHost, ps));
    HIPCHK(ctx, hipStreamSynchronize(ps));
    static const char *nm[16] = {"init", "claim", "Wrun", "drop", "Wmove", "ifind", "iapply", "lev", "W", "low", "ops", "mh",
                                 "gj", "bin", "dense", "final"};
    unsigned long long tot = 0;
    for (int k = 0; k < 16; k++) tot += hp[k];
    fprintf(stderr, "[NRQ_PROF] planner nblk=%u total=%llu clk:", nblk, tot);
    for (int k = 0; k < 16; k++) fprintf(stderr, " %s=%llu(%llu)", nm[k], hp[k], hp[16 + k]);
    fprintf(stderr, "\n");
#ifdef PL_STAMP
    fprintf(stderr, "[NRQ_PROF] round stamps (clocks up to point i, summed over the rounds):");
    for (int k = 0; k < 24; k++) fprintf(stderr, " %d:%llu", k, hp[32 + k]);
    fprintf(stderr, "\n[NRQ_PROF] chained peel: entries %llu, group clocks busy %llu (waiting for the rows %llu), claims %llu, phases %llu clocks %llu; flags trip %llu, subtraction trip %llu, claims %llu\n",
            hp[56], hp[57], hp[58], hp[59], hp[61], hp[60], hp[62], hp[63], hp[54]);
#endif
    (void)hipFree(pprof);
  }
  HIPCHK(ctx, hipMemcpyAsync(hs + off_hdrs, ds + off_hdrs, (size_t)nblk * sizeof(nrq_plan_hdr), hipMemcpyDeviceToHost, ps)); /* (pl_final_d's second copies) */
  HIPCHK(ctx, hipEventRecord(ctx->pstaged[f], ps));
  HIPCHK(ctx, hipEventRecord(ctx->planned[ab], ps));
  return 0;
}

/* was this run issued for exactly this call? */
static bool plan_run_matches(const nrq_ctx *ctx, const PlanRun &r, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_src,
                             size_t src_stride, const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap,
                             const uint32_t *h_rep_esi, const uint32_t *h_nrep, uint32_t rep_cap, const void *d_rep, size_t rep_stride,
                             const void *d_inter, size_t inter_stride, const uint32_t *h_avail) {
  if (ctx->vec_src || ctx->vec_rep || ctx->chunk_blocks) return false;
  if (r.K != K || r.Kp != Kp || r.T != T || r.nblk != nblk || r.lost_cap != lost_cap || r.rep_cap != rep_cap) return false;
  if (r.d_src != d_src || r.src_stride != src_stride || r.d_rep != d_rep || r.rep_stride != rep_stride || r.d_inter != d_inter ||
      r.inter_stride != inter_stride || r.has_avail != (h_avail != nullptr))
    return false;
  return memcmp(r.nlost.data(), h_nlost, (size_t)nblk * 4) == 0 && memcmp(r.nrep.data(), h_nrep, (size_t)nblk * 4) == 0 &&
         (!h_avail || memcmp(r.avail.data(), h_avail, (size_t)nblk * 4) == 0) &&
         memcmp(r.lost.data(), h_lost, (size_t)nblk * lost_cap * 4) == 0 && memcmp(r.resi.data(), h_rep_esi, (size_t)nblk * rep_cap * 4) == 0;
}

/* decode with the symbolic stage on the GPU: one planner workgroup per block, then the solve */
static int decode_device(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                         const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                         const uint32_t *h_nrep, uint32_t rep_cap, const void *d_rep, size_t rep_stride, void *d_inter,
                         size_t inter_stride, int *h_status, std::vector<uint8_t> *fallback, const uint32_t *h_avail,
                         uint32_t *h_used) {
  const double t_begin = now_ms();
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats.planner = 1;
  PlanRun local, *run = &local;
  bool ahead = false;
  while (!ctx->ahead.empty()) {
    PlanRun *f = ctx->ahead.front();
    ctx->ahead.pop_front();
    if (plan_run_matches(ctx, *f, K, Kp, T, nblk, d_src, src_stride, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep, rep_cap, d_rep, rep_stride,
                         d_inter, inter_stride, h_avail)) {
      run = f; /* the planner run of this very call is already on its way (or done) */
      ahead = true;
      break;
    }
    /* runs are consumed in the order they were issued: one this call was not issued for is over (its arena set is free again
     * once it has run); a later one may still be this call's (a caller that issued a run per reception state) */
    (void)hipEventSynchronize(ctx->planned[f->ab]);
    delete f;
    if (ctx->ahead.empty()) ctx->ahead_hint = 0;
  }
  struct Owner { PlanRun *r; ~Owner() { delete r; } } owner{ahead ? run : nullptr};
  if (!ahead) {
    const int rc0 = plan_launch(ctx, *run, K, Kp, T, nblk, d_src, src_stride, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep, rep_cap, d_rep,
                                rep_stride, d_inter, inter_stride, h_avail);
    if (rc0) return rc0;
  }
  const rq_params &p = run->p;
  KConst *kc = run->kc;
  const int ab = run->ab;
  const uint32_t arena_cap = run->arena_cap, max_nl = run->max_nl;
  hipStream_t ps = run->ps;
  uint8_t *hs = ctx->pstaging[run->f].p;
  const size_t off_hdrs = run->off_hdrs;
  (void)arena_cap;
  HIPCHK(ctx, hipEventSynchronize(ctx->planned[ab]));
  ctx->stats.plan_ms = now_ms() - t_begin;
  ctx->stats.plan_ahead = ahead ? 1 : 0;
  const nrq_plan_hdr *hd = reinterpret_cast<const nrq_plan_hdr *>(hs + off_hdrs);
  std::vector<const nrq_plan_hdr *> hdrs;
  std::vector<uint32_t> hblk; /* block of every header in hdrs (pick_and_launch: the batch's two block lists) */
  bool need_fallback = false;
  for (uint32_t b = 0; b < nblk; b++) {
    if (h_used) h_used[b] = 0;
    if (h_nlost[b] == 0) { h_status[b] = 1; continue; } /* nothing missing (nanorq.c:605-606) */
    if (hd[b].magic != NRQ_PLAN_MAGIC) return fail(ctx, -11, "device planner produced no header for block %u", b);
    if (hd[b].status == 0) {
      h_status[b] = 1;
      if (h_used) h_used[b] = h_nrep[b] + hd[b].reserved[1];
      hdrs.push_back(&hd[b]);
      hblk.push_back(b);
      if (ctx->stats.npiv == 0) {
        ctx->stats.npiv = hd[b].npiv; ctx->stats.u = hd[b].u; ctx->stats.nlev = hd[b].nlev; ctx->stats.nfree = hd[b].nfree;
      }
      ctx->stats.xor_ops += hd[b].n_xor_ops;
      ctx->stats.plan_bytes += hd[b].total_bytes;
    } else if (hd[b].reserved[0] == PL_FAIL_CAPACITY) {
      if (ctx->tune.prof || ctx->tune.diag) fprintf(stderr, "[NRQ_PROF] block %u: device planner capacity exceeded at planner_body.h:%u (npiv %u u %u nlev %u nrows %u M %u nlost %u)\n", b, hd[b].fail_site, hd[b].npiv, hd[b].u, hd[b].nlev, hd[b].nrows, hd[b].M, h_nlost[b]);
      (*fallback)[b] = 1;
      need_fallback = true;
      h_status[b] = 0;
    } else {
      if (ctx->tune.prof || ctx->tune.diag)
        fprintf(stderr, "[NRQ_PROF] block %u: not decodable: status %u reason %u site %u npiv %u u %u nlev %u nlow %u r2 %u nfree %u taken %u\n", b, hd[b].status,
                hd[b].reserved[0], hd[b].fail_site, hd[b].npiv, hd[b].u, hd[b].nlev, hd[b].nlow, hd[b].r2, hd[b].nfree, hd[b].reserved[1]);
      h_status[b] = 0;
    }
  }
  int result = 0;
  if (ctx->chunk_blocks) {
    /* one planner run, the solve chunk by chunk (nrq_decode_blocks_vc): an event per chunk for the caller's copy streams */
    if (ps != ctx->stream) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->planned[ab], 0));
    const uint32_t cb = ctx->chunk_blocks;
    for (uint32_t c0 = 0, ci = 0; c0 < nblk && !result; c0 += cb, ci++) {
      const uint32_t m = nblk - c0 < cb ? nblk - c0 : cb;
      std::vector<const nrq_plan_hdr *> hc;
      std::vector<uint32_t> hcb;
      for (uint32_t b = c0; b < c0 + m; b++)
        if (h_nlost[b] != 0 && hd[b].magic == NRQ_PLAN_MAGIC && hd[b].status == 0) { hc.push_back(&hd[b]); hcb.push_back(b - c0); }
      if (ctx->chunk_up && ctx->chunk_up[ci]) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)ctx->chunk_up[ci], 0));
      if (!hc.empty())
        result = pick_and_launch(ctx, hc, reinterpret_cast<const nrq_job *>(ctx->plan_jobs[ab].p) + c0, m, T, kc->dev, (d_inter ? p.L : 0u) + max_nl, &hcb);
      if (ctx->chunk_done && ctx->chunk_done[ci]) HIPCHK(ctx, hipEventRecord((hipEvent_t)ctx->chunk_done[ci], ctx->stream));
    }
    HIPCHK(ctx, hipEventRecord(ctx->arena_free[ab], ctx->stream));
    ctx->arena_busy[ab] = true;
  } else if (!hdrs.empty()) {
    if (ps != ctx->stream) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->planned[ab], 0));
    result = pick_and_launch(ctx, hdrs, reinterpret_cast<const nrq_job *>(ctx->plan_jobs[ab].p), nblk, T, kc->dev,
                             (d_inter ? p.L : 0u) + max_nl, &hblk);
    /* the launch reads the plan arenas and job records: a later planner run may not overwrite this set before it is done */
    HIPCHK(ctx, hipEventRecord(ctx->arena_free[ab], ctx->stream));
    ctx->arena_busy[ab] = true;
  }
  ctx->stats.host_ms = now_ms() - t_begin;
  if (result) return result;
  return need_fallback ? 1 : 0;
}

/* Issue the planner run of a decode call AHEAD of the call: the symbolic stage needs the reception pattern only, not the
 * symbols -- a receiver that knows which symbols of a batch of blocks it holds (or a pipeline that decodes batch after batch)
 * can have the plan built while earlier work is still being solved; the nrq_decode_blocks / _lazy call with the same arguments
 * then only waits for it.  A call with other arguments discards it. */
int nrq_decode_plan_ahead(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                          const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                          const uint32_t *h_nrep, const uint32_t *h_nrep_avail, uint32_t rep_cap, const void *d_rep, size_t rep_stride,
                          void *d_inter, size_t inter_stride) {
  if (!ctx) return -1;
  if (!d_src || T == 0 || nblk == 0 || !h_nlost || !h_nrep || !h_lost || !h_rep_esi) return fail(ctx, -1, "bad arguments");
  if (!ctx->planner) return 0; /* host planner: nothing to issue ahead */
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->ahead.size() >= NRQ_PLAN_AHEAD_MAX) return fail(ctx, -6, "nrq_decode_plan_ahead: %u runs are already waiting for their decode calls", (unsigned)NRQ_PLAN_AHEAD_MAX);
  PlanRun *r = new PlanRun();
  const int rc = plan_launch(ctx, *r, K, Kp, T, nblk, d_src, src_stride, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep, rep_cap, d_rep,
                             rep_stride, d_inter, inter_stride, h_nrep_avail);
  if (rc) { delete r; return rc; }
  r->d_src = d_src; r->src_stride = src_stride; r->d_rep = d_rep; r->rep_stride = rep_stride; r->d_inter = d_inter; r->inter_stride = inter_stride;
  r->has_avail = h_nrep_avail != nullptr;
  r->lost.assign(h_lost, h_lost + (size_t)nblk * lost_cap);
  r->resi.assign(h_rep_esi, h_rep_esi + (size_t)nblk * rep_cap);
  r->nlost.assign(h_nlost, h_nlost + nblk);
  r->nrep.assign(h_nrep, h_nrep + nblk);
  if (h_nrep_avail) r->avail.assign(h_nrep_avail, h_nrep_avail + nblk);
  ctx->ahead.push_back(r);
  if (ctx->ahead.size() > ctx->ahead_hint) ctx->ahead_hint = (uint32_t)ctx->ahead.size();
  return 0;
}

int nrq_decode_blocks_lazy(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                           const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                           const uint32_t *h_nrep, const uint32_t *h_nrep_avail, uint32_t rep_cap, const void *d_rep,
                           size_t rep_stride, void *d_inter, size_t inter_stride, int *h_status, uint32_t *h_used) {
  if (!ctx) return -1;
  if (!d_src || T == 0 || nblk == 0 || !h_nlost || !h_nrep || !h_status) return fail(ctx, -1, "bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->io_aligned = (ctx->vec_src ? vec_aligned(ctx->vec_src, nblk, T) : rows_aligned(d_src, src_stride, T)) &&
                    (ctx->vec_rep ? vec_aligned(ctx->vec_rep, nblk, T) : rows_aligned(d_rep, rep_stride, T)) &&
                    (!d_inter || rows_aligned(d_inter, inter_stride, T));
  /* A call with one or two SMALL blocks (the reference's own harness: one block per call): the planner kernel is a chain of ~120
   * phases that a lone block cannot fill -- 200 us at K=100, 460 at K=1000, whatever the block count up to one per CU -- while the
   * host planner, sequential per block, needs ~0.45 us per source symbol on the GPU box's CPU.  Measured with the reference's
   * benchmark.c (decode column, Gbit/s, device / host planner): K=100 3.5 / 9.3, K=500 8.6 / 23.8, K=1000 16.7 / 23.2, K=1500
   * 20.8 / 24.1, K=2000 22.3 / 21.8, K=2500 25.5 / 23.6, K=3000 25.7 / 23.1, K=4000 34.8 / 17.9.  So the host plans when its estimate is the shorter
   * one (option "host_plan_auto" / NRQ_HOST_PLAN_AUTO=0: never), unless plans were issued ahead or the call solves in chunks.
   * (Several small blocks: decode_host plans them one after the other -- its worker threads cost more to start than such plans take.) */
  const bool host_small = ctx->planner && ctx->tune.host_plan_auto && ctx->ahead.empty() && !ctx->chunk_blocks &&
                          (uint64_t)38u * nblk * K < (uint64_t)20000u + (uint64_t)28u * K; /* (tools/small_calls.py: host call ~65 us + 0.3-0.4 us x K per block, planner-kernel
                                                                                                      * call ~270 us + 0.28 us x K: one block of K < 2000, two of K < 416, four of K < 161) */
  if (!ctx->planner || host_small) {
    const int rc_ = decode_host(ctx, nullptr, K, Kp, T, nblk, d_src, src_stride, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep,
                                h_nrep_avail, h_used, rep_cap, d_rep, rep_stride, d_inter, inter_stride, h_status);
    if (host_small) ctx->stats.host_planned = 0; /* (a choice, not a fallback: tests read host_planned as "the device planner gave up") */
    return rc_;
  }
  std::vector<uint8_t> fallback(nblk, 0);
  int rc = decode_device(ctx, K, Kp, T, nblk, d_src, src_stride, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep, rep_cap, d_rep,
                         rep_stride, d_inter, inter_stride, h_status, &fallback, h_nrep_avail, h_used);
  if (rc <= 0) return rc;
  /* blocks that exceeded a device-planner capacity are planned on the host (rare) */
  uint32_t nfb = 0;
  for (uint8_t f : fallback) nfb += f;
  rc = decode_host(ctx, fallback.data(), K, Kp, T, nblk, d_src, src_stride, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep,
                     h_nrep_avail, h_used, rep_cap, d_rep, rep_stride, d_inter, inter_stride, h_status);
  ctx->stats.host_planned = nfb;
  if (ctx->tune.prof) fprintf(stderr, "[NRQ_PROF] %u of %u blocks re-planned on the host (device planner capacity)\n", nfb, nblk);
  return rc;
}

int nrq_encode_blocks_v(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_src, size_t src_stride,
                        const uint64_t *d_inter_v) {
  if (!ctx || !d_inter_v) return -1;
  ctx->vec_inter = d_inter_v;
  const int rc = nrq_encode_blocks(ctx, K, Kp, T, nblk, d_src, src_stride, nullptr, 0, 0, nullptr, nullptr, 0);
  ctx->vec_inter = nullptr;
  return rc;
}

int nrq_decode_blocks_v(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const uint64_t *d_src_v, const uint32_t *h_lost,
                        const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi, const uint32_t *h_nrep,
                        const uint32_t *h_nrep_avail, uint32_t rep_cap, const uint64_t *d_rep_v, int *h_status, uint32_t *h_used) {
  if (!ctx || !d_src_v || !d_rep_v) return -1;
  ctx->vec_src = d_src_v;
  ctx->vec_rep = d_rep_v;
  const int rc = nrq_decode_blocks_lazy(ctx, K, Kp, T, nblk, (void *)(uintptr_t)16, 0, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep, h_nrep_avail,
                                        rep_cap, (const void *)(uintptr_t)16, 0, nullptr, 0, h_status, h_used);
  ctx->vec_src = ctx->vec_rep = nullptr;
  return rc;
}

int nrq_decode_blocks_vc(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const uint64_t *d_src_v, const uint32_t *h_lost,
                         const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi, const uint32_t *h_nrep,
                         const uint32_t *h_nrep_avail, uint32_t rep_cap, const uint64_t *d_rep_v, int *h_status, uint32_t *h_used,
                         uint32_t chunk_blocks, void *const *chunk_done, void *const *upload_done) {
  if (!ctx || !chunk_blocks || !chunk_done) return -1;
  const uint32_t nchunks = (nblk + chunk_blocks - 1u) / chunk_blocks;
  if (!ctx->planner) {
    /* host planner: no chunks -- everything is solved by one launch, after all uploads; every event is recorded behind it */
    for (uint32_t i = 0; upload_done && i < nchunks; i++)
      if (upload_done[i]) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)upload_done[i], 0));
  } else {
    ctx->chunk_blocks = chunk_blocks; ctx->chunk_done = chunk_done; ctx->chunk_up = upload_done;
  }
  const int rc = nrq_decode_blocks_v(ctx, K, Kp, T, nblk, d_src_v, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep, h_nrep_avail, rep_cap, d_rep_v,
                                     h_status, h_used);
  const bool chunked = ctx->chunk_blocks != 0 && ctx->stats.host_planned == 0;
  ctx->chunk_blocks = 0; ctx->chunk_done = ctx->chunk_up = nullptr;
  if (!chunked) /* (also when blocks were re-planned on the host and solved by a later launch: the events say "all done") */
    for (uint32_t i = 0; i < nchunks; i++)
      if (chunk_done[i]) HIPCHK(ctx, hipEventRecord((hipEvent_t)chunk_done[i], ctx->stream));
  return rc;
}

int nrq_decode_blocks(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                      const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                      const uint32_t *h_nrep, uint32_t rep_cap, const void *d_rep, size_t rep_stride, void *d_inter,
                      size_t inter_stride, int *h_status) {
  return nrq_decode_blocks_lazy(ctx, K, Kp, T, nblk, d_src, src_stride, h_lost, h_nlost, lost_cap, h_rep_esi, h_nrep, nullptr,
                                rep_cap, d_rep, rep_stride, d_inter, inter_stride, h_status, nullptr);
}

int nrq_gen_symbols(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_inter, size_t inter_stride,
                    uint32_t n, const uint32_t *h_isi, void *d_out, size_t out_stride) {
  if (!ctx) return -1;
  if (!d_inter || !d_out || !h_isi || T == 0 || nblk == 0) return fail(ctx, -1, "bad arguments");
  if (n == 0) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  rq_params p;
  int rc = block_params(ctx, K, Kp, &p);
  if (rc) return rc;
  const int f = ctx->flip;
  ctx->flip ^= 1;
  HIPCHK(ctx, hipEventSynchronize(ctx->staged[f]));
  if ((rc = ensure_pin(ctx, ctx->staging[f], (size_t)n * 4))) return rc;
  if ((rc = ensure_dev(ctx, ctx->scratch[f], (size_t)n * 4))) return rc;
  memcpy(ctx->staging[f].p, h_isi, (size_t)n * 4);
  HIPCHK(ctx, hipMemcpyAsync(ctx->scratch[f].p, ctx->staging[f].p, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->staged[f], ctx->stream));
  hipLaunchKernelGGL(nrq_gen_kernel, dim3(n, nblk), dim3(NRQ_GEN_WG), 0, ctx->stream, p, T, (const uint8_t *)d_inter,
                     inter_stride, (const uint32_t *)ctx->scratch[f].p, (uint8_t *)d_out, out_stride);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

int nrq_gen_symbols_dev(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_inter, size_t inter_stride,
                        uint32_t n, const uint32_t *d_isi, void *d_out, size_t out_stride) {
  if (!ctx) return -1;
  if (!d_inter || !d_out || !d_isi || T == 0 || nblk == 0) return fail(ctx, -1, "bad arguments");
  if (n == 0) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  rq_params p;
  int rc = block_params(ctx, K, Kp, &p);
  if (rc) return rc;
  hipLaunchKernelGGL(nrq_gen_kernel, dim3(n, nblk), dim3(NRQ_GEN_WG), 0, ctx->stream, p, T, (const uint8_t *)d_inter,
                     inter_stride, d_isi, (uint8_t *)d_out, out_stride);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

int nrq_dev_alloc(nrq_ctx *ctx, size_t bytes, void **out) {
  if (!ctx || !out) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t want = bytes ? bytes : 16;
  want = (want + 4095) & ~(size_t)4095;
  auto it = ctx->pool_free.lower_bound(want);
  if (it != ctx->pool_free.end() && it->first <= want + want / 4 + 65536) {
    *out = it->second;
    ctx->pool_cached -= it->first;
    ctx->pool_free.erase(it);
    return 0;
  }
  if (nrq_inject(ctx)) { *out = nullptr; return fail(ctx, -10, "hipMalloc(%zu) failed: injected fault", want); }
  hipError_t e = hipMalloc(out, want);
  if (e != hipSuccess && !ctx->pool_free.empty()) { /* give the cached blocks back and try once more */
    (void)hipGetLastError();
    nrq_dev_trim(ctx);
    e = hipMalloc(out, want);
  }
  if (e != hipSuccess) { *out = nullptr; return fail(ctx, -10, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
  ctx->pool_size[*out] = want;
  return 0;
}
int nrq_dev_free(nrq_ctx *ctx, void *p) {
  if (!ctx) return -1;
  if (!p) return 0;
  auto it = ctx->pool_size.find(p);
  if (it == ctx->pool_size.end()) return fail(ctx, -1, "nrq_dev_free: not a block of this context");
  ctx->pool_free.emplace(it->second, p);
  ctx->pool_cached += it->second;
  if (ctx->pool_cached > ((size_t)24 << 30)) nrq_dev_trim(ctx); /* keep the cache bounded */
  return 0;
}
int nrq_dev_trim(nrq_ctx *ctx) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  for (auto &kv : ctx->pool_free) {
    (void)hipFree(kv.second);
    ctx->pool_size.erase(kv.second);
  }
  ctx->pool_free.clear();
  ctx->pool_cached = 0;
  return 0;
}

/* page-locked host memory (what the copy engines read and write at PCIe speed without a staging pass) */
int nrq_host_alloc_pinned(size_t bytes, void **out) {
  if (!out) return -1;
  *out = nullptr;
  return hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocPortable) == hipSuccess ? 0 : -10; /* (every device of the process may DMA it) */
}
void nrq_host_free_pinned(void *p) {
  if (p) (void)hipHostFree(p);
}
int nrq_host_register(void *p, size_t bytes) { return hipHostRegister(p, bytes, hipHostRegisterPortable) == hipSuccess ? 0 : -10; }
void nrq_host_unregister(void *p) {
  if (p) (void)hipHostUnregister(p);
}
int nrq_host_is_pinned(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return a.type == hipMemoryTypeHost ? 1 : 0;
}
/* the whole range [p, p + bytes): its first and last byte and a probe every 2 MiB in between (registrations are page
 * granular; a range that is page-locked at both ends and every 2 MiB is taken as page-locked throughout) */
int nrq_host_range_is_pinned(const void *p, size_t bytes) {
  if (!p) return 0;
  if (!bytes) return nrq_host_is_pinned(p);
  const uint8_t *q = static_cast<const uint8_t *>(p);
  if (!nrq_host_is_pinned(q) || !nrq_host_is_pinned(q + bytes - 1)) return 0;
  const size_t step = (size_t)2 << 20;
  for (size_t o = step - (reinterpret_cast<uintptr_t>(q) & (step - 1)); o < bytes; o += step)
    if (!nrq_host_is_pinned(q + o)) return 0;
  return 1;
}

/* the address a kernel reaches page-locked host memory at (hipHostMalloc'ed or hipHostRegister'ed); 0 = it cannot */
uint64_t nrq_host_device_address(const void *p) {
  if (!p || !nrq_host_is_pinned(p)) return 0;
  void *d = nullptr;
  if (hipHostGetDevicePointer(&d, const_cast<void *>(p), 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return (uint64_t)(uintptr_t)d;
}

/* streams and events of the object layer's copy pipeline; stream selector: 0 = the context's stream, 1 = upload
 * stream, 2 = download stream */
static hipStream_t sel_stream(nrq_ctx *ctx, int which) {
  return which == 1 ? ctx->aux[0] : which == 2 ? ctx->aux[1] : which == 3 ? ctx->aux[2] : ctx->stream;
}
int nrq_ctl_copy(nrq_ctx *ctx, int stream, void *d_dst, const void *h_pinned, size_t bytes) {
  if (!ctx || !d_dst || !h_pinned || (bytes & 15u) || (((uintptr_t)d_dst | (uintptr_t)h_pinned) & 15u)) return -1;
  if (!bytes) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t n16 = (uint32_t)(bytes / 16u);
  uint32_t g = (n16 + 255u) / 256u;
  if (g > 256u) g = 256u;
  hipLaunchKernelGGL(nrq_ctl_copy_kernel, dim3(g), dim3(256), 0, sel_stream(ctx, stream), reinterpret_cast<uint4 *>(d_dst),
                     reinterpret_cast<const uint4 *>(h_pinned), n16);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}
int nrq_copy_on(nrq_ctx *ctx, int stream, void *dst, const void *src, size_t bytes) {
  if (!ctx) return -1;
  if (!bytes) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, sel_stream(ctx, stream)));
  return 0;
}
int nrq_memset_on(nrq_ctx *ctx, int stream, void *d_dst, int value, size_t bytes) {
  if (!ctx) return -1;
  if (!bytes) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemsetAsync(d_dst, value, bytes, sel_stream(ctx, stream)));
  return 0;
}
int nrq_event_new(nrq_ctx *ctx, void **out) {
  if (!ctx || !out) return -1;
  hipEvent_t e;
  HIPCHK(ctx, hipSetDevice(ctx->device)); /* (an event belongs to the device that is current when it is created) */
  HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *out = e;
  return 0;
}
void nrq_event_free(void *ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
int nrq_event_record(nrq_ctx *ctx, void *ev, int stream) {
  if (!ctx || !ev) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipEventRecord((hipEvent_t)ev, sel_stream(ctx, stream)));
  return 0;
}
int nrq_stream_wait(nrq_ctx *ctx, int stream, void *ev) {
  if (!ctx || !ev) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamWaitEvent(sel_stream(ctx, stream), (hipEvent_t)ev, 0));
  return 0;
}
int nrq_event_sync(nrq_ctx *ctx, void *ev) {
  if (!ctx || !ev) return -1;
  HIPCHK(ctx, hipEventSynchronize((hipEvent_t)ev));
  return 0;
}
int nrq_stream_sync(nrq_ctx *ctx, int stream) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipStreamSynchronize(sel_stream(ctx, stream)));
  return 0;
}

int nrq_scatter_symbols(nrq_ctx *ctx, int stream, const void *d_blob, uint32_t n, uint32_t T, const uint64_t *h_dst) {
  if (!ctx || !d_blob || !h_dst || T == 0) return -1;
  if (n == 0) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int f = ctx->scat_flip;
  ctx->scat_flip ^= 1;
  HIPCHK(ctx, hipEventSynchronize(ctx->scat_ev[f])); /* the copy out of this pinned image two calls ago */
  int rc;
  if ((rc = ensure_pin(ctx, ctx->scat_pin[f], (size_t)n * 8))) return rc;
  if ((rc = ensure_dev(ctx, ctx->scat_dev[f], (size_t)n * 8))) return rc;
  memcpy(ctx->scat_pin[f].p, h_dst, (size_t)n * 8);
  hipStream_t st = sel_stream(ctx, stream);
  HIPCHK(ctx, hipMemcpyAsync(ctx->scat_dev[f].p, ctx->scat_pin[f].p, (size_t)n * 8, hipMemcpyHostToDevice, st));
  HIPCHK(ctx, hipEventRecord(ctx->scat_ev[f], st));
  hipLaunchKernelGGL(nrq_scatter_kernel, dim3(n), dim3(128), 0, st, (const uint8_t *)d_blob, T, (const uint64_t *)ctx->scat_dev[f].p, n);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

int nrq_scatter_symbols_dev(nrq_ctx *ctx, int stream, const void *d_blob, uint32_t n, uint32_t T, const uint64_t *d_dst) {
  if (!ctx || !d_blob || !d_dst || T == 0) return -1;
  if (n == 0) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(nrq_scatter_kernel, dim3(n), dim3(128), 0, sel_stream(ctx, stream), (const uint8_t *)d_blob, T, d_dst, n);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

int nrq_move_rows_dev(nrq_ctx *ctx, int stream, const uint64_t *d_pairs, uint32_t n, uint32_t T) {
  if (!ctx || !d_pairs || T == 0) return -1;
  if (n == 0) return 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(nrq_move_rows_kernel, dim3(n), dim3(128), 0, sel_stream(ctx, stream), d_pairs, T, n);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

int nrq_dev_upload(nrq_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
int nrq_dev_download(nrq_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
int nrq_dev_upload_async(nrq_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}
int nrq_dev_download_async(nrq_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}
int nrq_dev_copy(nrq_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
int nrq_dev_memset(nrq_ctx *ctx, void *d_dst, int value, size_t bytes) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
  return 0;
}

int nrq_ktime_enable(nrq_ctx *ctx, int on) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->ktime_on = on != 0;
  ctx->ktime_used = 0;
  ctx->ptime_used = 0;
  if (on) {
    if (!ctx->ktime_base) HIPCHK(ctx, hipEventCreate(&ctx->ktime_base));
    HIPCHK(ctx, hipEventRecord(ctx->ktime_base, ctx->stream));
  }
  return 0;
}
/* launch intervals [start, start+dur) in ms relative to ref's nrq_ktime_enable (ref may be another context of
 * the same device: kernels of several streams overlap and the caller wants the union of their busy time) */
int nrq_ktime_read_intervals(nrq_ctx *ctx, nrq_ctx *ref, float *start_ms, float *dur_ms, uint32_t cap, uint32_t *count) {
  if (!ctx || !ref || !count || !ref->ktime_base) return -1;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipEventSynchronize(ref->ktime_base));
  uint32_t n = (uint32_t)ctx->ktime_used;
  for (uint32_t k = 0; k < n && k < cap; k++) {
    HIPCHK(ctx, hipEventElapsedTime(&start_ms[k], ref->ktime_base, ctx->ktime_pool[k].first));
    HIPCHK(ctx, hipEventElapsedTime(&dur_ms[k], ctx->ktime_pool[k].first, ctx->ktime_pool[k].second));
  }
  *count = n;
  ctx->ktime_used = 0;
  return 0;
}
int nrq_ptime_read(nrq_ctx *ctx, float *ms_out, uint32_t cap, uint32_t *count) {
  if (!ctx || !count) return -1;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->plan_stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->plan_stream_b));
  uint32_t n = (uint32_t)ctx->ptime_used;
  for (uint32_t k = 0; k < n && k < cap; k++)
    HIPCHK(ctx, hipEventElapsedTime(&ms_out[k], ctx->ptime_pool[k].first, ctx->ptime_pool[k].second));
  *count = n;
  ctx->ptime_used = 0;
  return 0;
}
int nrq_ktime_read(nrq_ctx *ctx, float *ms_out, uint32_t cap, uint32_t *count) {
  if (!ctx || !count) return -1;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  uint32_t n = (uint32_t)ctx->ktime_used;
  for (uint32_t k = 0; k < n && k < cap; k++)
    HIPCHK(ctx, hipEventElapsedTime(&ms_out[k], ctx->ktime_pool[k].first, ctx->ktime_pool[k].second));
  *count = n;
  ctx->ktime_used = 0;
  return 0;
}

int nrq_timer_start(nrq_ctx *ctx) {
  if (!ctx) return -1;
  HIPCHK(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return 0;
}
int nrq_timer_stop_ms(nrq_ctx *ctx, float *ms) {
  if (!ctx || !ms) return -1;
  HIPCHK(ctx, hipEventRecord(ctx->t1, ctx->stream));
  HIPCHK(ctx, hipEventSynchronize(ctx->t1));
  HIPCHK(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return 0;
}

} /* extern "C" */
