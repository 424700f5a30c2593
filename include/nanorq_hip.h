/*
 * nanorq_hip.h -- thin C ABI into the gfx950 (MI355X) RaptorQ precode-solve / symbol-generation
 * path.  Plain C types only: device and host pointers as void*, sizes, int status returns
 * (0 = ok, negative = error, text via nrq_ctx_error()); no C++ exceptions cross this line.
 *
 * This is the seam the drop-in library (include/nanorq.h, include/io.h) sits on, and the
 * entry points a binding of the reference would call instead of its CPU solver.  Reference
 * interfaces replaced (file:line in sleepybishop/nanorq):
 *   precode_matrix_gen / precode_matrix_invert / precode_matrix_intermediate   include/precode.h:10-12
 *   nanorq_generate_symbols  (load + plan + replay)                            lib/nanorq.c:206-232
 *   nanorq_repair_block      (patch + plan + replay + regenerate gaps)         lib/nanorq.c:591-631
 *   decode_row / nanorq_encode (LT symbol generation)                          lib/nanorq.c:184-204, :403-435
 *   oblas oaxpy/oscal/oswaprow row kernels (absent submodule deps/oblas)       precode.c:7,18,20
 * The unit of work is a batch of independent source blocks of equal (K, T) that stay resident in
 * HBM; persistent workgroups solve one 16/8/4/2-byte column strip of one block at a time out of LDS.
 */
#ifndef NANORQ_HIP_H
#define NANORQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nrq_ctx nrq_ctx;

/* statistics of the last encode/decode call (host side, for benchmarks and tests) */
typedef struct nrq_call_stats {
  double plan_ms;      /* symbolic stage wall time (all blocks) */
  double host_ms;      /* whole host side of the call before the launch returns */
  uint32_t strip_bytes;/* column-strip width chosen (16/12/8/4/2) */
  uint32_t lds_bytes;  /* dynamic LDS per workgroup */
  uint32_t grid;       /* (persistent) workgroups launched */
  uint32_t planner;    /* 0 = host planner, 1 = device planner */
  uint64_t plan_bytes; /* plan bytes resident on the device for this call */
  uint64_t xor_ops;    /* row XOR ops in the forward passes, summed over blocks */
  uint32_t npiv, u, nlev, nfree; /* of block 0 */
  uint32_t wg_threads; /* threads per workgroup of the solve launch */
  uint32_t strips_per_slot; /* strips a work slot holds (a whole 128-byte line group unless work is scarce) */
  uint32_t wg_waves_per_simd; /* register budget of the solve kernel variant launched: waves per SIMD it was compiled for */
  uint32_t host_planned; /* decode blocks whose plan exceeded a device-planner capacity and was rebuilt on the host */
  uint32_t movers_aligned; /* 1: the solve kernel variant without byte-wise paths in its movers (all rows aligned, T a multiple of the strip) */
  uint32_t plan_ahead;  /* 1: the decode found its planner run already issued (nrq_decode_plan_ahead) */
  uint32_t strip_bytes_b, blocks_b; /* the batch's SECOND block list: blocks whose LDS image does not fit at strip_bytes run in a launch
                                     * of their own at this narrower width (0 / 0: one list) */
} nrq_call_stats;

/* One context per GPU (one process per GPU: no cross-device state).  `stream` is a hipStream_t the
 * caller owns (NULL = the default stream); all work of later calls is enqueued on it. */
int nrq_ctx_create(int device, void *stream, nrq_ctx **out);
void nrq_ctx_destroy(nrq_ctx *ctx);
int nrq_ctx_set_stream(nrq_ctx *ctx, void *stream);
const char *nrq_ctx_error(nrq_ctx *ctx);
int nrq_ctx_sync(nrq_ctx *ctx);
void nrq_ctx_last_stats(nrq_ctx *ctx, nrq_call_stats *out);
/* where decode plans are built: 1 = on the GPU (default; planner kernel, one workgroup per block),
 * 0 = on the host (thread pool).  The environment variable NRQ_HOST_PLANNER=1 selects 0 at creation. */
int nrq_ctx_set_planner(nrq_ctx *ctx, int device_planner);
/* Tuning / test knobs of the launch path (the NRQ_* environment variables read at nrq_ctx_create set the same
 * fields): "max_wb" widest column strip considered (16/12/8/4/2), "no_wb12" (round 5's widths: no 12-byte strip), "host_plan_auto" (1: a decode call of one or two small blocks is planned on the host -- faster than the
 * planner kernel's latency for a lone block of K < 2000; 0: never), "plan_pack" (1: small blocks' planner workgroups share a CU whatever the batch size;
 * default: only when the batch has more blocks than the device has compute units), "no_split", "no_balance", "no_plan_stream",
 * "reserve_cus", "solve_grid", "big_wg", "map_spread", "no_tiny", "tiny_div", "wide_g", "small_waves4", "no_plan_split",
 * "plan_split_force", "plan_small_state", "plan_big_wg", "encplan_dev_min_l", "no_lists" (one solve launch per batch at the width every block fits, no second block list), "lds_max" (tests: LDS bytes a strip
 * image may take when the block lists are formed), "plan_ucap" (inactive columns the device
 * planner has room for; a block that needs more is re-planned on the host).  Results never depend on them, only speed.
 * Fault injection for tests: "fail_after" n -- the n-th checked runtime call of the context from now on (allocation, copy,
 * event / stream operation, the error check behind a launch) is not made and fails instead, once (0 = off);
 * "faults_injected" returns the number of failures injected so far.  "fail_after" exists only on a context created with
 * NANORQ_HIP_FAULT_INJECT=1 in the environment (else: unknown option, -1), and not at all in a -DNRQ_NO_FAULT_INJECT build. */
int nrq_ctx_set_option(nrq_ctx *ctx, const char *name, long long value);
/* threads used for host-side planning (0 = hardware concurrency) */
int nrq_ctx_set_threads(nrq_ctx *ctx, int n);

/* Parameters of RFC 6330 section 5.3.1.2 for K source symbols: out = {K',J,S,H,W,L,P,P1,U,B}. */
int nrq_params(uint32_t K, uint32_t out[10]);

/* In every call below K is the number of source symbols of a block and Kp the RFC 6330 Table 2 row (K')
 * it is coded with: 0 = the row the RFC assigns to K; nanorq passes block 0's K' for every block of an
 * object (lib/nanorq.c:289, :372), which can exceed a short block's own row.
 *
 * Build (or fetch the cached) plan for encoding blocks of K symbols: the counterpart of
 * nanorq_precalculate (lib/nanorq.c:393-401).  Implied by nrq_encode_blocks. */
int nrq_precalculate(nrq_ctx *ctx, uint32_t K, uint32_t Kp);
/* First-use costs of blocks of (K, K') paid now instead of inside the first encode / decode: the code object is loaded, the
 * per-K' constants are built and uploaded, and with encode_plan != 0 the encode plan is built (1: enqueued, for big K', as by
 * nrq_precalculate; 2: waited for).  The object layer calls it from nanorq_encoder_new* / nanorq_decoder_new* -- the reference pays nothing
 * comparable (its tables are static), so the constructors are where a drop-in can put it without a timed call seeing it. */
int nrq_warm(nrq_ctx *ctx, uint32_t K, uint32_t Kp, int encode_plan);
/* drop cached encode plans (so that a benchmark can time plan generation) */
void nrq_plan_cache_clear(nrq_ctx *ctx);

/* Encode nblk source blocks (asynchronous on the context's stream).
 *   d_src    device: block b at d_src + b*src_stride, source symbol j at + j*T        (K*T bytes)
 *   d_inter  device or NULL: block b's L intermediate symbols at d_inter + b*inter_stride
 *   h_esis   host: nrep repair ESIs (each K <= esi < 2^24), the same list for every block
 *   d_rep    device: block b's repair symbol q at d_rep + b*rep_stride + q*T
 * Bit-exact with nanorq_generate_symbols + nanorq_encode(esi) of the reference. */
int nrq_encode_blocks(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_src, size_t src_stride,
                      void *d_inter, size_t inter_stride, uint32_t nrep, const uint32_t *h_esis, void *d_rep,
                      size_t rep_stride);

/* Decode nblk source blocks (asynchronous on the context's stream; h_status is final on return).
 *   d_src    device, in/out: block b at d_src + b*src_stride; received source symbols are in place at
 *            row esi, rows of missing symbols are overwritten with the recovered symbols
 *   h_lost   host: missing source ESIs of block b, ascending, at h_lost[b*lost_cap .. + h_nlost[b])
 *   h_rep_esi host: ESIs (>= K) of block b's received repair symbols in ARRIVAL order at
 *            h_rep_esi[b*rep_cap .. + h_nrep[b]); symbol q at d_rep + b*rep_stride + q*T
 *   d_inter  device or NULL: intermediate symbols out
 *   h_status host out: 1 = block recovered, 0 = not decodable (fewer repair symbols than gaps, or
 *            rank deficient: the caller may add symbols and retry, as with nanorq_repair_block)
 * The i-th missing ESI takes the i-th repair symbol, surplus symbols become extra constraint rows
 * (lib/nanorq.c:527-565). */
int nrq_decode_blocks(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                      const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                      const uint32_t *h_nrep, uint32_t rep_cap, const void *d_rep, size_t rep_stride, void *d_inter,
                      size_t inter_stride, int *h_status);

/* Same, using spare symbols only when needed.  Block b's system is first built from its first h_nrep[b]
 * repair symbols (>= h_nlost[b]); if it is rank deficient the planner takes further symbols from the list,
 * one at a time up to h_nrep_avail[b], as additional constraint rows -- on the GPU without redoing the
 * elimination -- instead of failing.  This is what a receiver does that calls nanorq_repair_block again
 * after one more packet (lib/nanorq.c:620-623), minus the second pass.  h_nrep_avail may be NULL (= h_nrep);
 * h_used (nullable) receives the number of repair symbols each recovered block consumed. */
int nrq_decode_blocks_lazy(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                           const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                           const uint32_t *h_nrep, const uint32_t *h_nrep_avail, uint32_t rep_cap, const void *d_rep,
                           size_t rep_stride, void *d_inter, size_t inter_stride, int *h_status, uint32_t *h_used);

/* Issue the planner run of a later nrq_decode_blocks / nrq_decode_blocks_lazy call now (same arguments, without the result
 * arrays): the symbolic stage (reference precode_matrix_invert's planning half, lib/precode.c:347-377) needs the reception
 * pattern only, not the symbols, so it can run on the planner stream while earlier batches are still being solved.  The
 * decode call with identical arguments then only waits for it (nrq_call_stats::plan_ahead = 1).  Up to two runs may be waiting
 * (-6 beyond that); they are consumed in the order they were issued: a decode call discards the runs in front of the one
 * issued for it (and all of them if none was).  Two runs issued back to back execute side by side (two planner streams, two workspaces, three sets of
 * plan arenas): a planner workgroup is latency bound on one compute unit, so a pipeline that keeps two batches' plans in
 * flight gets them at twice the rate.  Not for the per-block-address variants (_v, _vc). */
int nrq_decode_plan_ahead(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, void *d_src, size_t src_stride,
                          const uint32_t *h_lost, const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi,
                          const uint32_t *h_nrep, const uint32_t *h_nrep_avail, uint32_t rep_cap, const void *d_rep, size_t rep_stride,
                          void *d_inter, size_t inter_stride);

/* nrq_encode_blocks (no repair symbols) with block b's intermediate symbols going to device address d_inter_v[b]. */
int nrq_encode_blocks_v(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_src, size_t src_stride,
                        const uint64_t *d_inter_v);
/* nrq_decode_blocks_lazy for blocks that do not lie at a fixed stride: block b's source rows at device address
 * d_src_v[b], its repair symbols at d_rep_v[b] (host arrays of device addresses).  No intermediate symbols out. */
int nrq_decode_blocks_v(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const uint64_t *d_src_v, const uint32_t *h_lost,
                        const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi, const uint32_t *h_nrep,
                        const uint32_t *h_nrep_avail, uint32_t rep_cap, const uint64_t *d_rep_v, int *h_status, uint32_t *h_used);

/* nrq_decode_blocks_v with ONE planner run for all blocks and the solve in chunks of `chunk_blocks` blocks, in order: after
 * chunk i has been solved the event chunk_done[i] (nrq_event_new; ceil(nblk / chunk_blocks) of them) is recorded on the
 * context's stream, so that a caller can start moving the first blocks (nrq_stream_wait on a copy stream) while the later
 * ones are still being solved -- the planner's fixed cost (a launch the host waits for, ~3 ms) is paid once, not per chunk.
 * The events are valid once the call has RETURNED (the caller enqueues its waits afterwards): with the host planner, or when
 * blocks had to be re-planned on the host, all chunks are solved by one later launch and every event is recorded behind that;
 * on a negative return value the events are undefined.
 * upload_done (nullable): events the solve of chunk i waits for before it reads the chunk's symbols (their upload).
 * Replaces nanorq_repair_block per block (reference lib/nanorq.c:591-631). */
int nrq_decode_blocks_vc(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const uint64_t *d_src_v, const uint32_t *h_lost,
                         const uint32_t *h_nlost, uint32_t lost_cap, const uint32_t *h_rep_esi, const uint32_t *h_nrep,
                         const uint32_t *h_nrep_avail, uint32_t rep_cap, const uint64_t *d_rep_v, int *h_status, uint32_t *h_used,
                         uint32_t chunk_blocks, void *const *chunk_done, void *const *upload_done);

/* Generate encoding symbols from intermediate symbols already in HBM (after encode/decode with
 * d_inter != NULL): symbol q of block b = LT(C_b, isi[q]) -> d_out + b*out_stride + q*T.
 * h_isi are INTERNAL symbol ids (esi for esi < K, esi + K' - K for repair symbols). */
int nrq_gen_symbols(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_inter, size_t inter_stride,
                    uint32_t n, const uint32_t *h_isi, void *d_out, size_t out_stride);

/* the same with the list of internal symbol ids already in device memory (enqueue only: nothing is staged, nothing waited for;
 * a pipeline that generates the same symbols for block after block uploads the list once) */
int nrq_gen_symbols_dev(nrq_ctx *ctx, uint32_t K, uint32_t Kp, uint32_t T, uint32_t nblk, const void *d_inter, size_t inter_stride,
                        uint32_t n, const uint32_t *d_isi, void *d_out, size_t out_stride);

/* Raw device memory helpers for hosts without a HIP binding of their own (the drop-in C library and
 * the ctypes tests use them; PyTorch callers pass tensor data pointers instead). */
int nrq_dev_alloc(nrq_ctx *ctx, size_t bytes, void **out);
int nrq_dev_free(nrq_ctx *ctx, void *p);
int nrq_dev_upload(nrq_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);   /* synchronous */
int nrq_dev_download(nrq_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* synchronous */
int nrq_dev_memset(nrq_ctx *ctx, void *d_dst, int value, size_t bytes);
/* enqueue-only variants (ordered on the context's stream; complete after nrq_ctx_sync) */
int nrq_dev_upload_async(nrq_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int nrq_dev_download_async(nrq_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int nrq_dev_copy(nrq_ctx *ctx, void *d_dst, const void *d_src, size_t bytes); /* device to device, enqueue-only */

/* nrq_dev_alloc / nrq_dev_free come out of a caching pool of the context (no hipMalloc / hipFree per call, no
 * synchronisation on free); nrq_dev_trim hands the cached blocks back to the driver. */
int nrq_dev_trim(nrq_ctx *ctx);

/* ---- what the object layer's streaming path is made of (reference: transfer_esi / load_symbol_matrix,
 * lib/nanorq.c:148-182, and the seek+read/write loops of lib/io.c behind them) ----
 * Page-locked host memory: the copy engines move it at PCIe speed and asynchronously. */
int nrq_host_alloc_pinned(size_t bytes, void **out);
void nrq_host_free_pinned(void *p);
int nrq_host_register(void *p, size_t bytes);   /* page-lock caller memory in place */
void nrq_host_unregister(void *p);
int nrq_host_is_pinned(const void *p);          /* 1 if p lies in page-locked (allocated or registered) host memory */
int nrq_host_range_is_pinned(const void *p, size_t bytes); /* 1 if all of [p, p + bytes) does (ends + a probe every 2 MiB) */
/* Copies and ordering on the context's streams.  `stream`: 0 = the context's stream (kernels), 1 = its upload stream,
 * 2 = its download stream.  Everything is enqueue-only; events order the streams among each other. */
int nrq_copy_on(nrq_ctx *ctx, int stream, void *dst, const void *src, size_t bytes); /* direction from the pointers */
int nrq_memset_on(nrq_ctx *ctx, int stream, void *d_dst, int value, size_t bytes);
int nrq_event_new(nrq_ctx *ctx, void **out);
void nrq_event_free(void *ev);
int nrq_event_record(nrq_ctx *ctx, void *ev, int stream);
int nrq_stream_wait(nrq_ctx *ctx, int stream, void *ev);  /* later work on `stream` waits for ev */
int nrq_event_sync(nrq_ctx *ctx, void *ev);                /* the host waits */
int nrq_stream_sync(nrq_ctx *ctx, int stream);
/* Symbol ingestion on the device: symbol k (T bytes at d_blob + k*T, device memory) is copied to device address
 * h_dst[k] (0 = skip) -- received packets go up in one piece and are put into their rows by a kernel. */
int nrq_scatter_symbols(nrq_ctx *ctx, int stream, const void *d_blob, uint32_t n, uint32_t T, const uint64_t *h_dst);
/* Control data (KBs to a few MB, a multiple of 16 bytes, both ends 16-byte aligned) from page-locked host memory into a device
 * buffer by a kernel that reads the host memory itself: unlike a copy it does not queue behind the bulk uploads the copy
 * engine is busy with.  stream selector as in nrq_copy_on, plus 3 = the stream of the sorting kernels. */
int nrq_ctl_copy(nrq_ctx *ctx, int stream, void *d_dst, const void *h_pinned, size_t bytes);
/* the same with the destination addresses already on the device (enqueue only: no staging of the list, nothing waited for) */
int nrq_scatter_symbols_dev(nrq_ctx *ctx, int stream, const void *d_blob, uint32_t n, uint32_t T, const uint64_t *d_dst);

/* Rows by address pair (enqueue only): row k, T bytes, from address d_pairs[2k] to address d_pairs[2k+1] (0 = skip); either side
 * may be page-locked host memory the device can address.  The receiver's repaired symbols go from their device rows straight to
 * their places in the caller's page-locked output buffer this way (nanorq_repair_all): no staging, no copy per row. */
int nrq_move_rows_dev(nrq_ctx *ctx, int stream, const uint64_t *d_pairs, uint32_t n, uint32_t T);
/* the address at which a kernel reaches page-locked host memory (hipHostMalloc'ed or hipHostRegister'ed); 0 = it cannot */
uint64_t nrq_host_device_address(const void *p);

/* Per-launch duration of the solve kernel, measured with HIP events recorded on the launch stream
 * immediately around each launch (bench.py's roofline leg).  enable(1) starts collecting; read()
 * synchronises, returns the durations of the launches since the last read/enable in launch order. */
int nrq_ktime_enable(nrq_ctx *ctx, int on);
int nrq_ktime_read(nrq_ctx *ctx, float *ms_out, uint32_t cap, uint32_t *count);
/* same for the decode planner (all kernels of a planner run, on the stream they run on) */
int nrq_ptime_read(nrq_ctx *ctx, float *ms_out, uint32_t cap, uint32_t *count);
/* same as intervals [start, start+dur) in ms after ref's nrq_ktime_enable(1); ref may be another context of the
 * same GPU, so that launches of several streams can be put on one time axis */
int nrq_ktime_read_intervals(nrq_ctx *ctx, nrq_ctx *ref, float *start_ms, float *dur_ms, uint32_t cap, uint32_t *count);

/* Generic stream timer (HIP events on the context's stream). */
int nrq_timer_start(nrq_ctx *ctx);
int nrq_timer_stop_ms(nrq_ctx *ctx, float *ms); /* synchronises */

#ifdef __cplusplus
}
#endif
#endif /* NANORQ_HIP_H */
