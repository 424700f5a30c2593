/*
 * plan.h -- the device "plan": everything the symbolic stage hands to the data stage.
 *
 * It plays the role of the reference's `schedule` (sched.h:6-27: op list + c/d permutations +
 * marks) but is shaped for the LDS-resident column-strip solver, not
 * for a serial replay:
 *   - the forward substitution through the triangular block X is a list of XOR ops
 *     (dst ^= src on symbol rows) grouped by dependency level and padded to NRQ_CHUNK-op
 *     chunks; ops inside a chunk are independent of each other up to XOR-accumulation;
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
#define NRQ_CHUNK 256u             /* ops per chunk == threads of the solve workgroup */
#define NRQ_NOP 0xFFFFFFFFu        /* padding op */
#define NRQ_NOSLOT 0xFFFFu
#define NRQ_MAX_FREE 32u
/* When a peeling round has no row of weight 1, this many open rows (sparsest first) are resolved by
 * inactivation before peeling resumes: fewer, wider cascades -> ~40 % fewer rounds in the planner and ~15 %
 * fewer dependency levels in the plan, for ~4 % more inactive columns. */
#define NRQ_MULTI_INACT 8u

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
  uint32_t nchunk1; /* chunks of the pivot forward pass */
  uint32_t nchunk2; /* chunks of the low-row pass and of the GF(2) combination pass (E rows) */
  uint32_t wpr;     /* 32-bit words per W row: ceil(u/32) */
  uint32_t lpr;     /* reserved (was: words per G2 row) */
  uint32_t npiv_pad;/* stride (in pivots) of the transposed W image, multiple of 64 */
  uint32_t n_xor_ops; /* real (non-padding) ops in both passes, for statistics */

  uint32_t off_ops;     /* u32[(nchunk1+nchunk2)*NRQ_CHUNK]: dst | src<<16, NRQ_NOP = padding
                         * (8 more all-NOP chunks follow the last one: prefetch slack).  Slots >= M are the
                         * r2 scratch rows E_p (slot M+p) of the dense stage: E_p = XOR of leftover rows */
  uint32_t off_pivslot; /* u16[npiv]: slot of pivot k */
  uint32_t off_pivcol;  /* u16[npiv]: column of pivot k */
  uint32_t off_wt;      /* u32[wpr*npiv_pad]: word w of W row k at [w*npiv_pad + k] */
  uint32_t off_lowslot; /* u16[nlow] */
  uint32_t off_g2;      /* reserved (the GF(2) combinations are part of the op stream) */
  uint32_t off_pivx;    /* u16[r2]: inactive-column index solved by reduced row p */
  uint32_t off_fbits;   /* u32[r2]: bit f set -> C_u[pivx[p]] ^= C_free[f] */
  uint32_t off_mh;      /* u8[H*r2], [h][p]: R_h ^= mh * E_p */
  uint32_t off_freex;   /* u16[nfree] */
  uint32_t off_hinv;    /* u8[nfree*H], [f][h]: C_free[f] = SUM_h hinv * R_h */
  uint32_t off_colslot; /* u16[L]: slot that finally holds intermediate symbol C[c] */
  uint32_t off_pivof;   /* u16[Kp+S]: slot of the pivot row of column c, NRQ_NOSLOT if inactive */
  uint32_t off_uslot;   /* u16[u]: slot that receives inactive column x */
  uint32_t off_sync;    /* u32[ceil(nchunk/32)+2]: bit c set -> workgroup barrier after chunk c */
  uint32_t total_bytes;
  uint32_t reserved[2];
} nrq_plan_hdr;

/* Per-K' constants shared by every plan of that K': the HDPC block (RFC 6330 section 5.3.3.3) and the
 * "base" constraint structure -- the S LDPC rows and the LT rows of ISI 0..K'-1 (reference
 * precode_matrix_gen, precode.c:90-97) in CSR and CSC form -- which a decode only patches in the rows
 * whose source symbol was replaced by a repair symbol (reference patch_precode_matrix, nanorq.c:527-547). */
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
  uint32_t reserved[2];
} nrq_kconst_hdr;

#endif
