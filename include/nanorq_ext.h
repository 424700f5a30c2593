/*
 * nanorq_ext.h -- RFC 6330 conformance options behind explicit flags.
 *
 * nanorq deviates from RFC 6330 in three places; this library follows nanorq by default (so that it interoperates with
 * nanorq peers) and offers the RFC's behaviour through the flags below:
 *   - OTI packing.  nanorq: common = F << 24 | (T - 1), scheme-specific = (Z - 1) << 24 | (N - 1) << 8 | Al
 *     (reference lib/nanorq.c:309-324, decoded at :336-376).  RFC 6330 section 3.3.2 / 3.3.3: common = F (40 bits) |
 *     reserved (8) | T (16), scheme-specific = Z (8) | N (16) | Al (8), values as they are (so Z <= 255, T <= 65535).
 *   - K' of a short block.  nanorq codes every block of an object with block 0's table row (lib/nanorq.c:289, :372);
 *     RFC 6330 section 5.3.1.2 gives each source block the smallest row >= its own K.
 *   - Sub-blocking.  nanorq forces N = 1 (lib/nanorq.c:78; the offset arithmetic for N > 1 is there, :114-128, but
 *     unreachable).  RFC 6330 section 4.4.1.2 splits every symbol into N sub-symbols and lays the object out
 *     sub-block by sub-block.
 * Objects created through nanorq.h are unaffected.  An encoder and its decoder must use the same flags.
 */
#ifndef NANORQ_EXT_H
#define NANORQ_EXT_H

#include "nanorq.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NANORQ_EXT_RFC_OTI 1u      /* nanorq_oti_common / _scheme_specific (and the decoder constructor) use RFC 6330 section 3.3 */
#define NANORQ_EXT_PER_BLOCK_KP 2u /* each block is coded with the table row of its own K */
#define NANORQ_EXT_SUBBLOCKS 4u    /* N > 1 is honoured (encoder: the N passed in; decoder: the N of the OTI) */

/* nanorq_encoder_new_ex with N sub-blocks per source block (1 <= N <= T / Al; ignored, i.e. 1, without
 * NANORQ_EXT_SUBBLOCKS) and the flags above.  Returns NULL where nanorq_encoder_new_ex does, if N is out of range,
 * or if NANORQ_EXT_RFC_OTI cannot represent the parameters (Z > 255). */
nanorq *nanorq_encoder_new_ext(size_t len, uint16_t T, uint16_t K, uint16_t Z, uint16_t N, uint8_t Al, uint32_t flags);
/* nanorq_decoder_new for OTI words packed according to `flags` */
nanorq *nanorq_decoder_new_ext(uint64_t common, uint32_t specific, uint32_t flags);
/* the flags an object was created with; its N; the table row (K') block `sbn` is coded with */
uint32_t nanorq_ext_flags(nanorq *rq);
size_t nanorq_sub_blocks(nanorq *rq);
size_t nanorq_block_kprime(nanorq *rq, uint8_t sbn);

#ifdef __cplusplus
}
#endif
#endif /* NANORQ_EXT_H */
