/*
 * plan.h -- the device "plan": everything the symbolic stage hands to the data stage.
 *
 * It plays the role of the reference's `schedule` (sched.h:6-27: op list + c/d permutations +
 * marks) but is shaped for the LDS-resident column-strip solver, not
 * for a serial replay:
 *   - the forward substitution through the triangular block X is a list of XOR ops
 *     (dst ^= src on symbol rows) grouped by dependency level; ops of one level are independent
 *     of each other up to XOR-accumulation.  The stream is cut into rows of NRQ_ROW ops that ONE
 *     wave of the solve workgroup executes as a fixed software pipeline: the sources of row
 *     k+NRQ_PIPE are read before row k is applied (the LDS pipeline keeps one wave's accesses in
 *     order, so nothing but program order is needed -- no barriers).  Hence a row may only read
 *     slots last written NRQ_PIPE or more rows earlier: every level group is followed by
 *     NRQ_PIPE-1 all-NOP rows;
 *   - the fill-in block of the reference's U_upper is kept as a bit matrix W (i x u) and
 *     applied after the dense stage instead of re-running the sparse passes;
 *   - the dense stage is pre-inverted: a GF(2) combination matrix for the binary rows, GF(256)
 *     coefficient columns for the H HDPC rows, a tiny GF(256) inverse for the free columns.
 * One plan is one contiguous arena: this header, then the arrays at the byte offsets below
 * (all 16-byte aligned).  Slots are constraint-row indices of A (0..M-1), i.e. rows of the
 * reference's D matrix (nanorq.c:137-142, :179-180): [0,S) LDPC, [S,S+H) HDPC, [S+H,L) LT rows
 * of ISI 0..K'-1 (or the repair symbol that replaced a missing one), [L,M) surplus repair rows.
 */
#ifndef NRQ_PLAN_H
#define NRQ_PLAN_H

#include <stdint.h>

#define NRQ_PLAN_MAGIC 0x4e525131u /* "NRQ1" */
#define NRQ_ROW 64u                /* ops per row == lanes of a wave */
#ifndef NRQ_PIPE
#define NRQ_PIPE 2u                /* rows between the read of a row's sources and its application (measured on
                                    * MI355X: 2 beats 3 and 4 -- a row costs ~63 clocks of issue either way, spacer rows included) */
#endif
/* The row pipeline holds the op words of the next NRQ_RING rows (or 2 * NRQ_RING: the big solve workgroup) in registers,
 * fetched FOUR ROWS AT A TIME: the stream is stored quad-interleaved -- the words of rows 4g .. 4g+3 of a lane lie next to
 * each other (one 16-byte load per lane and quad), word (row, lane) at NRQ_OP_INDEX(row, lane).  Why: a wave may have 63
 * vector-memory instructions in flight, no more (vmcnt is 6 bits), and the CU's vector L1 returns loads in order, so while
 * the data movers of the workgroup have HBM misses in flight every op word waits an HBM latency D behind them: one dword
 * load per row gives at most 63 rows per D (measured: rows of 62-70 clocks as soon as the movers keep loads in flight all the
 * time, at 44 without them; tools/microbench/fwd_loop.hip), a quad load per four rows four times that. */
#ifndef NRQ_RING_MULT
#define NRQ_RING_MULT 5u
#endif
#define NRQ_RING (NRQ_RING_MULT * 4u * (NRQ_PIPE + 1u)) /* 60 rows: a multiple of the quad and of the pipeline's value sets */
#define NRQ_RING_MAX (2u * NRQ_RING)
#define NRQ_PAD_ROWS (2u * NRQ_RING_MAX + NRQ_PIPE + 4u) /* all-NOP rows after the stream: op words are fetched ahead unconditionally,
                                    * and the pipeline runs whole trips of its ring */
#define NRQ_OP_INDEX(row, lane) ((((size_t)(row) >> 2) * NRQ_ROW + (size_t)(lane)) * 4u + ((size_t)(row) & 3u))
#define NRQ_OP_LANE_OF_INDEX(i) ((uint32_t)((i) >> 2) & (NRQ_ROW - 1u))
#define NRQ_STREAM_ROWS(rows) (((rows) + 3u) & ~3u) /* rows the stream's memory holds: whole quads */
/* Op word: dst | src << 16, both as slot + NRQ_SCRATCH.  The first NRQ_SCRATCH slots of the LDS image are
 * per-lane scratch: the padding op of lane l reads and writes scratch slot l, so padding needs no branch. */
#define NRQ_SCRATCH 64u
#define NRQ_OP(dst, src) (((uint32_t)(dst) + NRQ_SCRATCH) | (((uint32_t)(src) + NRQ_SCRATCH) << 16))
#define NRQ_NOP_AT(pos) (((uint32_t)(pos) & (NRQ_ROW - 1u)) * 0x10001u) /* padding op at stream position pos */
#define NRQ_OP_IS_NOP(op) (((op) & 0xFFFFu) < NRQ_SCRATCH)
#define NRQ_OP_DST(op) (((op) & 0xFFFFu) - NRQ_SCRATCH)
#define NRQ_OP_SRC(op) (((op) >> 16) - NRQ_SCRATCH)
#define NRQ_NOSLOT 0xFFFFu
#define NRQ_MAX_FREE 32u

/* Lane placement inside a level group.  The ops of a group are independent of each other, so which lane of which row an
 * op takes is free -- and it decides the LDS bank conflicts of the row pipeline: a 64-lane LDS atomic (ds_xor_b64, like
 * ds_write_b64) is served in four groups of 16 CONTIGUOUS lanes with bank = dword address mod 32
 * (MI355X_MICROARCH.md, LDS), so with 16-byte slots the targets of a 16-lane block fall into 8 classes (slot mod 8) and a
 * block costs as many LDS cycles as its busiest class has different slots: ~4.1 with random lanes, 2 at best.  Measured
 * (tools/microbench/lds_half.hip, two waves on 8-byte halves): 54.6 clocks per row with random lanes, 35 with two targets
 * per class and block, 28 with no conflict at all.  So: an op of class d that is the r-th of its class in the group --
 * finishing ops counted first, they must stay out of the group's last NRQ_PIPE-1 rows -- takes lane 2d + (r & 1) of the
 * group's 16-lane block r >> 1; what a class has beyond two per block goes to the lanes other classes leave empty, class
 * by class.  Both planners place ops this way (the device planner when its per-class counters fit the LDS). */
#define NRQ_LANE_CLASSES 8u
#if defined(__HIPCC__)
#define NRQ_PLAN_FN __host__ __device__ static inline
#else
#define NRQ_PLAN_FN static inline
#endif
NRQ_PLAN_FN uint32_t nrq_op_class(uint32_t op) { return op & (NRQ_LANE_CLASSES - 1u); } /* (NRQ_SCRATCH is a multiple of 8) */
/* rows a group takes: its n ops, and its finishing ops (cf[d] of class d) in all but the last NRQ_PIPE-1 rows */
NRQ_PLAN_FN uint32_t nrq_group_span(uint32_t n, const uint32_t *cf) {
  uint32_t mf = 0;
  for (uint32_t d = 0; d < NRQ_LANE_CLASSES; d++) mf = cf[d] > mf ? cf[d] : mf;
  const uint32_t need = (mf + NRQ_LANE_CLASSES - 1u) / NRQ_LANE_CLASSES + (NRQ_PIPE - 1u); /* (2 per block, 4 blocks per row) */
  const uint32_t have = (n + NRQ_ROW - 1u) / NRQ_ROW;
  return have > need ? have : need;
}
/* place (offset from the group's first op word) of the op of class d with rank r; ct[] = ops per class of the whole group */
NRQ_PLAN_FN uint32_t nrq_lane_place(uint32_t span, const uint32_t *ct, uint32_t d, uint32_t r) {
  const uint32_t cap = span * NRQ_LANE_CLASSES; /* two per block, span * 4 blocks */
  if (r >= cap) { /* beyond the class's own lanes: the o-th lane left empty by the classes, in class order */
    uint32_t o = r - cap;
    for (uint32_t e = 0; e < d; e++) o += ct[e] > cap ? ct[e] - cap : 0u;
    for (uint32_t e = 0; e < NRQ_LANE_CLASSES; e++) {
      const uint32_t holes = ct[e] < cap ? cap - ct[e] : 0u;
      if (o < holes) { d = e; r = ct[e] + o; break; }
      o -= holes;
    }
  }
  return (r >> 1) * 16u + 2u * d + (r & 1u);
}
/* Early ops by release.  An early op may run in any group of its window [level(src)+1, level(dst)-1].  Those whose window is at
 * most NRQ_EARLY_NARROW groups wide take one by hash; the others are served in the order of their release, each group taking
 * what its rows have lanes for (host planner: encode plans of L < 12000 and the fallback; the device planner takes every early
 * op's group by hash).  Measured at K=8192: 878 rows instead of 1086 for the same 55.6 k ops -- and an encode forward window of
 * 68 k clocks instead of 71-75 k: with full rows the pass is bound by the LDS time of its real ops (bank conflicts of 63 random
 * slots per row), not by the row count, so the device planner was left as it is. */
#ifndef NRQ_EARLY_NARROW
#define NRQ_EARLY_NARROW 128u
#endif
/* When a peeling round has no row of weight 1, this many open rows (sparsest first) are resolved by
 * inactivation before peeling resumes: fewer, wider cascades -> fewer rounds in the planner and fewer dependency
 * levels in the plan, for a few more inactive columns (which the back-substitution pays for).  Measured on the
 * headline workload: 2 / 3 / 4 / 6 / 8 / 12 / 16 -> 1061 / 1074 / 1080 / 1084 / 1074 / 1070 / 1051 Gbit/s (round 2, rows taken one
 * after the other).  With the rows of an event taken at once (round 5: pl_event_*, pl_inact_apply_a) 4 / 6 / 8 / 12 / 16 / 24 ->
 * 1514 / 1521 / 1541 / 1540 / 1540 / 1293: more rows an event shorten the planner (1.80 / 1.70 / 1.64 / 1.61 / 1.62 ms) at the same
 * decode solve until the inactive columns of SOME block of the launch no longer fit the 16-byte strip image and the whole launch
 * falls to 8-byte strips (24: always; 12: one launch in 40 at 10 % loss, 13 in 40 at 30 %; 8: none / 4; 6: none / 2).  K=2000:
 * 1352 / 1367 / 1354 for 6 / 8 / 12. */
#ifndef NRQ_MULTI_INACT
#define NRQ_MULTI_INACT 8u
#endif

/* From this many intermediate symbols on the GF(2) combinations of the dense stage are a bit matrix (off_augt), below it ops
 * of the stream: the table phase costs a strip ~7 k clocks of trips and barriers whatever the size, the ops r2 * nlow / 2 / 64
 * rows of the forward wave (measured: 16 k clocks at K=8192, 1 k at K=1000, where four 256-thread workgroups share a CU). */
#ifndef NRQ_AUG_MATRIX_MIN_L
#define NRQ_AUG_MATRIX_MIN_L 5000u
#endif
typedef struct nrq_plan_hdr {
  uint32_t magic;
  uint32_t status; /* 0 = solvable, 1 = rank(A) < L (decode must fail, nanorq.c:620-623) */
  uint32_t K, Kp, J, S, H, W, L, P, P1, B;
  uint32_t M;       /* slots = L + overhead rows */
  uint32_t npiv;    /* i: rows/columns resolved by peeling */
  uint32_t u;       /* inactivated columns (u = L - i) */
  uint32_t nlow;    /* binary rows left without a pivot */
  uint32_t r2;      /* GF(2) rank reached on the u inactive columns with those rows */
  uint32_t nfree;   /* u - r2: columns that need the HDPC rows */
  uint32_t nlev;    /* dependency depth of the peeled block */
  uint32_t nrows;   /* rows of the op stream (pivot levels, leftover rows, spare rows) */
  uint32_t pipe;    /* NRQ_PIPE the stream was laid out for */
  uint32_t wpr;     /* 32-bit words per W row: ceil(u/32) */
  uint32_t npiv_pad;/* stride (in pivots) of the transposed W image, multiple of 64 */
  uint32_t n_xor_ops; /* real (non-padding) ops in both passes, for statistics */

  uint32_t off_ops;     /* u32[(nrows+NRQ_PAD_ROWS)*NRQ_ROW]: op words (above), padding = NRQ_NOP_AT.  (Slots >= M are
                         * the r2 scratch rows E_p, slot M+p, of the dense stage: see off_augt) */
  uint32_t off_pivslot; /* u16[npiv]: slot of pivot k */
  uint32_t off_pivcol;  /* u16[npiv]: column of pivot k */
  uint32_t off_wt;      /* u32[wpr*npiv_pad]: word w of W row k at [w*npiv_pad + k] */
  uint32_t off_lowslot; /* u16[nlow] */
  uint32_t off_pivx;    /* u16[r2]: inactive-column index solved by reduced row p */
  uint32_t off_fbits;   /* u32[r2]: bit f set -> C_u[pivx[p]] ^= C_free[f] */
  uint32_t off_mh;      /* u8[H*r2], [h][p]: R_h ^= mh * E_p */
  uint32_t off_freex;   /* u16[nfree] */
  uint32_t off_hinv;    /* u8[nfree*H], [f][h]: C_free[f] = SUM_h hinv * R_h */
  uint32_t off_colslot; /* u16[L]: slot that finally holds intermediate symbol C[c] */
  uint32_t off_pivof;   /* u16[Kp+S]: slot of the pivot row of column c, NRQ_NOSLOT if inactive */
  uint32_t off_uslot;   /* u16[u]: slot that receives inactive column x */
  uint32_t total_bytes;
  uint32_t reserved[2]; /* device planner: [0] = PL_FAIL_* reason, [1] = repair symbols taken beyond the initial ones */
  uint32_t fail_site;   /* device planner: planner_body.h line that raised a capacity failure (diagnostics) */
  /* The GF(2) combinations of the dense stage, E_p = XOR of the leftover rows named by the augmented part of reduced row p
   * (slot M+p), as a bit matrix: bit j of the row = leftover row j (slot lowslot[j]) takes part.  The solve kernel applies it
   * with 16-entry XOR tables over groups of four leftover rows (solve_body.h ph_low_tables / ph_combine) -- as ops of the
   * stream these ~r2 * nlow / 2 terms were a quarter of all row operations, run by the single forward wave. */
  uint32_t off_augt;    /* u32[lpr * aug_stride]: word w of reduced row p at [w * aug_stride + p] */
  uint32_t lpr;         /* words per row: ceil(nlow / 32); 0: the combinations are ops of the stream (blocks of L < NRQ_AUG_MATRIX_MIN_L) */
  uint32_t aug_stride;  /* >= r2, multiple of 4 */
} nrq_plan_hdr;

/* Per-K' constants shared by every plan of that K': the HDPC block (RFC 6330 section 5.3.3.3) and the
 * "base" constraint structure -- the S LDPC rows and the LT rows of ISI 0..K'-1 (reference
 * precode_matrix_gen, precode.c:90-97) in CSR and CSC form -- which a decode only patches in the rows
 * whose source symbol was replaced by a repair symbol (reference patch_precode_matrix, nanorq.c:527-547). */
#define NRQ_CHEAD 16u /* entries of a column head (off_chead) */
typedef struct nrq_kconst_hdr {
  uint32_t Kp, S, H, n; /* n = Kp + S */
  uint32_t off_g;       /* u8[H*n] row-major: the HDPC block (reference precode.c:60-83) */
  uint32_t off_b12;     /* u8[n]: b1 | b2<<4 of column c (two unit entries of MT), c < n-1 */
  uint32_t total_bytes;
  uint32_t L, W, P, nnz;
  uint32_t off_rptr;    /* u32[L+1]: base CSR row pointers (rows [S,S+H) are empty) */
  uint32_t off_cidx;    /* u16[nnz]: base CSR column indices */
  uint32_t off_cptr;    /* u32[L+1]: base CSC column pointers */
  uint32_t off_ridx;    /* u16[nnz]: base CSC row indices, ascending per column */
  uint32_t off_state;   /* u32[L]: per base row, (count << 24 | sum) of its column ids below W */
  uint32_t off_gt;      /* u8[n*16]: the HDPC block transposed, 16 bytes per column (rows >= H are 0) */
  uint32_t off_erow;    /* u16[nnz]: the row of every base CSR entry (lets a pass run over entries, not rows) */
  uint32_t off_chead;   /* u16[L*16]: the first 16 row indices of every base CSC column, 0xFFFF behind the last (one trip to a
                         * column's rows instead of pointer-then-list: the chained peel, planner_body.h) */
} nrq_kconst_hdr;

#endif
