/*
 * nanorq.h -- RaptorQ (RFC 6330) object encoder / decoder, MI355X-native build.
 *
 * Drop-in for the public API of sleepybishop/nanorq (include/nanorq.h:19-83 of the reference):
 * every function below has the reference's name, signature, argument meaning and error
 * convention (NULL / false / 0 / NANORQ_SYM_ERR, no errno, nothing printed), so that encode.c,
 * decode.c and benchmark.c of the reference compile and link against libnanorq_hip.so unchanged.
 * What differs is below the API: the precode solve and the symbol generation run as HIP kernels
 * on gfx950 (include/nanorq_hip.h); the object, its partitioning and the symbol bookkeeping stay
 * on the host in C (nanorq_amd/csrc/nanorq_api.c).  There is no CPU solver in this library:
 * without a usable GPU the calls that need the solve report failure.
 *
 * Conventions kept from the reference:
 *   - `nanorq` is opaque; one object is not thread-safe, distinct objects are independent;
 *   - `data` buffers are caller-owned and exactly nanorq_symbol_size() bytes;
 *   - the OTI words are nanorq's own packing (T-1, Z-1, N-1 stored), not RFC 6330 section 3.3.
 *
 * What a GPU behind the API adds, and its switches (environment, read once):
 *   - the constructors pay the first-use costs -- GPU context, code object, the per-K' constants and (encoders) the encode plan:
 *     tens of milliseconds (the host planner takes 25 ms at K=27000, the device-built plan of K'=56403 is waited for ~15 ms);
 *     NANORQ_HIP_LAZY=1 leaves them to the first call that needs the GPU, for callers that construct an object only to read
 *     its OTI;
 *   - an ioctx_from_mem region of >= 1 MiB handed to nanorq_generate_symbols is page-locked in place on first use and stays
 *     so until its destroy() (the block is then read by DMA, with no host copy); NANORQ_HIP_AUTOPIN=0 switches that off;
 *   - nanorq_repair_block decodes, in the same device batch, the other decodable blocks of the object and hands their rows
 *     out when their own call comes (same verdicts, same bytes); NANORQ_HIP_REPAIR_AHEAD=0 decodes one block per call;
 *   - NANORQ_HIP_DEVICES=0,1,... / NANORQ_HIP_DEVICE=n choose the GPUs (nanorq_batch.h);
 *   - the library runs six streams beside the caller's and the HIP runtime multiplexes a process's streams over 4 hardware
 *     queues by default: a host process should export GPU_MAX_HW_QUEUES=8 before its first HIP call.  The library leaves the
 *     process environment alone; NANORQ_HIP_SET_ENV=1 asks it to set the variable (if unset) when it is loaded.
 */
#ifndef NANORQ_H
#define NANORQ_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "io.h"

#ifdef __cplusplus
extern "C" {
#endif

/* results of nanorq_decoder_add_symbol */
#define NANORQ_SYM_ERR -1  /* unknown block or ESI above the configured maximum */
#define NANORQ_SYM_ADDED 0 /* stored */
#define NANORQ_SYM_IGN 1   /* block already complete, symbol not needed */
#define NANORQ_SYM_DUP 2   /* this ESI was seen before */

/* largest object: 256 source blocks x 56403 symbols x 65535 bytes... capped as in the reference */
#define NANORQ_MAX_TRANSFER 946270874880ULL

typedef struct nanorq nanorq;

/* ---------------------------------------------------------------- object construction ---- */

/* Encoder for an object of `len` bytes cut into symbols of T bytes (T is snapped down to a multiple of
 * the alignment Al in {1,2,4,8}).  Exactly one of K (symbols per source block) and Z (number of source
 * blocks) may be given, the other being 0; with both 0 at least 16 blocks are used.  NULL if the
 * object cannot be represented (len too large, more than 256 blocks, more than 56403 symbols per block). */
nanorq *nanorq_encoder_new_ex(size_t len, uint16_t T, uint16_t K, uint16_t Z, uint8_t Al);
/* same with K = Z = 0 */
nanorq *nanorq_encoder_new(size_t len, uint16_t T, uint8_t Al);

/* Decoder from the two OTI words an encoder reports.  The largest accepted ESI defaults to 2*K'. */
nanorq *nanorq_decoder_new(uint64_t common, uint32_t specific);

/* releases the object, its cached precode plan and every block */
void nanorq_free(nanorq *rq);

/* ---------------------------------------------------------------------- object facts ---- */

uint64_t nanorq_oti_common(nanorq *rq);          /* transfer length and symbol size */
uint32_t nanorq_oti_scheme_specific(nanorq *rq); /* blocks, sub-blocks, alignment */
size_t nanorq_transfer_length(nanorq *rq);       /* F */
size_t nanorq_symbol_size(nanorq *rq);           /* T */
size_t nanorq_blocks(nanorq *rq);                /* Z */
size_t nanorq_block_symbols(nanorq *rq, uint8_t sbn); /* K of that block (0 for an unknown block) */
size_t nanorq_max_blocks(nanorq *rq);            /* 256 */
/* packet tag: sbn in the top 8 bits, esi in the low 24 */
uint32_t nanorq_tag(uint8_t sbn, uint32_t esi);

/* ------------------------------------------------------------------------- encoding ---- */

/* Build the precode plan for this object's block size once so that every later block reuses it. */
bool nanorq_precalculate(nanorq *rq);

/* Read the block's source symbols from `io` and compute its intermediate symbols (on the GPU).
 * True on success; a second call for a block that is already solved is a no-op. */
bool nanorq_generate_symbols(nanorq *rq, uint8_t sbn, struct ioctx *io);

/* Write encoding symbol `esi` of block `sbn` into `data`; returns the bytes written (T) or 0.
 * esi < K before the block is solved copies the source symbol; esi >= K (repair, < 2^24) solves the block
 * on first use. */
size_t nanorq_encode(nanorq *rq, void *data, uint32_t esi, uint8_t sbn, struct ioctx *io);

/* forget what was computed for a block but keep its buffers */
void nanorq_encoder_reset(nanorq *rq, uint8_t sbn);
/* release a block's buffers */
void nanorq_encoder_cleanup(nanorq *rq, uint8_t sbn);

/* ------------------------------------------------------------------------- decoding ---- */

/* Largest ESI add_symbol accepts for blocks created afterwards; K' <= max_esi < 2^24. */
bool nanorq_set_max_esi(nanorq *rq, uint32_t max_esi);

/* Hand one received symbol (tag = nanorq_tag(sbn, esi)) to the decoder; `data` is copied.  Source symbols
 * are written through to `io` at their place in the object. Returns a NANORQ_SYM_* code. */
int nanorq_decoder_add_symbol(nanorq *rq, void *data, uint32_t tag, struct ioctx *io);

size_t nanorq_num_missing(nanorq *rq, uint8_t sbn); /* source symbols still missing */
size_t nanorq_num_repair(nanorq *rq, uint8_t sbn);  /* repair symbols held */

/* Recover the missing source symbols of a block (on the GPU) and write them to `io`.  True when the block is
 * complete afterwards; false when there are fewer repair symbols than gaps or the system is rank
 * deficient -- add more symbols and call again. */
bool nanorq_repair_block(nanorq *rq, struct ioctx *io, uint8_t sbn);

#ifdef __cplusplus
}
#endif
#endif /* NANORQ_H */
