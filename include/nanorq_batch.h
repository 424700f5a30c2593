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

/* Decoder: nanorq_decoder_add_symbol (reference lib/nanorq.c:478-509) for n symbols: symbol k is the T bytes at
 * data + k*T with tag tags[k].  results[k] (may be NULL) receives the NANORQ_SYM_* code of symbol k.  Returns the
 * number of symbols stored (NANORQ_SYM_ADDED). */
size_t nanorq_decoder_add_symbols(nanorq *rq, const void *data, const uint32_t *tags, uint32_t n, int *results, struct ioctx *io);

/* Decoder: nanorq_repair_block (reference lib/nanorq.c:591-631) for every block that misses source symbols and holds
 * at least as many repair symbols as it misses, in one device batch per block size.  Returns the number of blocks of
 * the object that are complete afterwards; blocks whose system is rank deficient stay incomplete and retryable. */
size_t nanorq_repair_all(nanorq *rq, struct ioctx *io);

#ifdef __cplusplus
}
#endif
#endif /* NANORQ_BATCH_H */
