/*
 * nanorq_batch.h -- batched variants of the per-block / per-symbol calls of nanorq.h.
 *
 * The reference API moves one block per nanorq_generate_symbols / nanorq_repair_block call and one symbol per
 * nanorq_encode / nanorq_decoder_add_symbol call (include/nanorq.h:41-80 of sleepybishop/nanorq); behind a GPU each
 * of those is a PCIe round trip and a kernel launch for ONE block.  These calls do the same work for every block
 * of the object in one device batch.  They are additions, not replacements: state,
 * return conventions and the bytes produced are those of the per-block calls, and the two families can be mixed
 * on one object.
 */
#ifndef NANORQ_BATCH_H
#define NANORQ_BATCH_H

#include "nanorq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Encoder: nanorq_generate_symbols (reference lib/nanorq.c:206-232) for every block of the object that is not solved
 * yet -- source symbols read from `io`, all blocks of equal size solved by one batched launch.  Returns the number of
 * blocks that are solved afterwards (== nanorq_blocks(rq) on success). */
size_t nanorq_generate_symbols_all(nanorq *rq, struct ioctx *io);

/* Encoder: nanorq_encode (reference lib/nanorq.c:403-435) for the n consecutive encoding symbols esi0 .. esi0+n-1 of
 * block `sbn`; `data` receives n * nanorq_symbol_size(rq) bytes.  Returns the bytes written (0 on failure). */
size_t nanorq_encode_range(nanorq *rq, void *data, uint32_t esi0, uint32_t n, uint8_t sbn, struct ioctx *io);

/* Encoder: the n consecutive REPAIR symbols esi0 .. esi0+n-1 (esi0 >= the symbols of every block) of ALL blocks: `data`
 * receives, block after block, n * nanorq_symbol_size(rq) bytes each (solving what is not solved yet).  One download per
 * device instead of a launch, a wait and a download per block.  Called on an object none of whose blocks is solved yet, with
 * `data` in page-locked memory, it is the whole sender as one pipeline: a chunk of blocks travels up, is solved, its repair
 * symbols are generated and travel down while the next chunk travels up -- both directions of the link at once (128 blocks of
 * K=8192, T=1280: 26.8 ms against 30.9 ms for nanorq_generate_symbols_all followed by this call).  Returns the bytes written
 * (0 on failure). */
size_t nanorq_encode_range_all(nanorq *rq, void *data, uint32_t esi0, uint32_t n, struct ioctx *io);

/* Decoder: nanorq_decoder_add_symbol (reference lib/nanorq.c:478-509) for n symbols: symbol k is the T bytes at
 * data + k*T with tag tags[k].  results[k] (may be NULL) receives the NANORQ_SYM_* code of symbol k.  Returns the
 * number of symbols stored (NANORQ_SYM_ADDED). */
size_t nanorq_decoder_add_symbols(nanorq *rq, const void *data, const uint32_t *tags, uint32_t n, int *results, struct ioctx *io);

/* Decoder: nanorq_decoder_add_symbols that only ENQUEUES the way of the bytes to the GPU (page-locked packet buffer; with any
 * other buffer it is the call above).  Bookkeeping and result codes are final on return -- they are those of
 * nanorq_decoder_add_symbol (reference lib/nanorq.c:478-509) -- but `data` must stay untouched until nanorq_repair_all,
 * nanorq_decoder_flush, nanorq_repair_block of a block the batch feeds, or nanorq_free has returned.  The receiver's two
 * stations then overlap: nanorq_repair_all plans at once (the symbolic stage needs the reception pattern, not the symbols),
 * every chunk of blocks is solved as soon as the upload piece that completes it has landed, and decoded blocks travel
 * back while later pieces still travel up. */
size_t nanorq_decoder_add_symbols_async(nanorq *rq, const void *data, const uint32_t *tags, uint32_t n, int *results, struct ioctx *io);

/* Decoder: nanorq_repair_block (reference lib/nanorq.c:591-631) for every block that misses source symbols and holds
 * at least as many repair symbols as it misses, in one device batch per block size.  Returns the number of blocks of
 * the object that are complete afterwards; blocks whose system is rank deficient stay incomplete and retryable. */
size_t nanorq_repair_all(nanorq *rq, struct ioctx *io);

/* ---- streaming I/O: page-locked memory (second half) ----
 * A GPU behind PCIe is fed fastest by asynchronous DMA copies straight out of (and into) page-locked host memory.  With
 * the contexts below the batched calls above move whole blocks that way, overlapped with the solve of the neighbouring
 * blocks: nanorq_generate_symbols_all reads the object from the context's memory without a host-side copy;
 * nanorq_decoder_add_symbols, given a page-locked packet buffer, uploads it in one piece and sorts the symbols into their
 * rows on the GPU; nanorq_repair_all writes whole decoded blocks into the context's memory.  Results are the bytes the
 * per-symbol calls produce.  One difference in timing: on this path the source symbols a decoder receives reach the
 * output context at the next nanorq_repair_all / nanorq_repair_block of their block (or nanorq_decoder_flush), not
 * inside the add call (reference: write-through at lib/nanorq.c:498). */
/* memory context over `sz` bytes of page-locked memory it allocates and owns (freed by destroy) */
struct ioctx *ioctx_from_pinned_mem(size_t sz);
/* memory context over caller memory that is page-locked in place for the life of the context */
struct ioctx *ioctx_from_registered_mem(uint8_t *ptr, size_t sz);
/* the bytes of a memory context (any of ioctx_from_mem / _pinned_mem / _registered_mem), NULL for other contexts */
uint8_t *ioctx_mem_base(struct ioctx *io);
/* page-locked buffers for packets (what a receive path hands to nanorq_decoder_add_symbols) */
void *nanorq_pinned_alloc(size_t bytes);
void nanorq_pinned_free(void *p);
/* Decoder: write the received source symbols that have not reached `io` yet (blocks fed through the page-locked path
 * and not repaired since).  Returns the number of blocks written. */
size_t nanorq_decoder_flush(nanorq *rq, struct ioctx *io);


/* ---- devices ----
 * NANORQ_HIP_DEVICES=0,1,... (read once, at the first call that needs a GPU) names the GPUs of the process; source blocks
 * share nothing (reference lib/nanorq.c:97-112), so block sbn of every object lives on device number sbn mod N and the
 * batched calls above run one host thread per device.  The packets and symbols produced do not depend on N. */
size_t nanorq_devices(void); /* contexts in use (0: no GPU could be opened -- every call that needs one fails) */
/* give back the page-locked host rows and device pool blocks cached from freed objects */
void nanorq_trim(void);
/* nrq_ctx_set_option (include/nanorq_hip.h) on the context of device number `dev` of the process (0 .. nanorq_devices() - 1):
 * tuning switches and the fault injection the tests use ("fail_after").  Returns what nrq_ctx_set_option returns, -1 without
 * such a device.  Three switches belong to the object layer itself (any `dev`): "host_rows" 0 / 1 -- with a page-locked output
 * context given to nanorq_decoder_add_symbols(_async), the received SOURCE symbols of device-resident blocks are written to
 * their places in the output by the host as they arrive (what the reference does per symbol, nanorq.c:478-509) and
 * nanorq_repair_all brings down the repaired rows only (default 1; NANORQ_HIP_HOST_ROWS=0: whole blocks come down) --,
 * "book_threads" n -- host threads that book a packet
 * batch of nanorq_decoder_add_symbols(_async), each the blocks sbn mod n (0 = default: NANORQ_HIP_BOOK_THREADS, else half the
 * cores the process may use, at most 16; up to 32 when set) -- and "book_min" -- symbols from which a batch is booked by more than one thread
 * (0 = default, 65536).  Result codes and bytes do not depend on any of them. */
int nanorq_hip_option(size_t dev, const char *name, long long value);

#ifdef __cplusplus
}
#endif
#endif /* NANORQ_BATCH_H */
