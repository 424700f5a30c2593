/* io_priv.h -- what nanorq_api.c knows about io.c beyond include/io.h */
#ifndef NRQ_IO_PRIV_H
#define NRQ_IO_PRIV_H
#include "../../include/io.h"
/* true (and the region) if `io` is a page-locked memory context of this library whose vtable is untouched: its bytes
 * can be the source or the target of an asynchronous DMA copy */
bool ioctx_dma_region(struct ioctx *io, uint8_t **base, size_t *len);
/* the same, and an ordinary ioctx_from_mem context of at least 1 MiB is page-locked in place on first use */
bool ioctx_dma_region_auto(struct ioctx *io, uint8_t **base, size_t *len);
#endif
