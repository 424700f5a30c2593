/*
 * io.h -- byte-stream context the nanorq API reads source data from and writes decoded data to.
 *
 * Binary-compatible with the reference's `struct ioctx` (sleepybishop/nanorq include/io.h:7-16: same
 * members, same order, same types), so callers that fill or poke the vtable themselves keep working.
 * The three constructors below are this library's own implementations (nanorq_amd/csrc/io.c) of
 * ioctx_from_file / ioctx_mmap_file / ioctx_from_mem (reference lib/io.c:58-80, :138-157, :333-388).
 */
#ifndef NANORQ_IOCTX_H
#define NANORQ_IOCTX_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct ioctx {
  /* copy up to `len` bytes from the cursor into `buf`, advance, return bytes copied */
  size_t (*read)(struct ioctx *, uint8_t *, size_t);
  /* copy up to `len` bytes from `buf` to the cursor, advance, return bytes copied */
  size_t (*write)(struct ioctx *, const uint8_t *, size_t);
  /* move the cursor to an absolute byte offset; false if the offset is not addressable */
  bool (*seek)(struct ioctx *, const size_t);
  /* total size in bytes */
  size_t (*size)(struct ioctx *);
  /* cursor position */
  long (*tell)(struct ioctx *);
  /* release the context (never the caller's memory buffer) */
  void (*destroy)(struct ioctx *);
  bool seekable;
  bool writable;
};

/* stdio file.  t != 0: open `fn` for reading (an encoder's source); t == 0: create/truncate `fn`
 * for reading and writing (a decoder's sink). */
struct ioctx *ioctx_from_file(const char *fn, int t);
/* memory-mapped file, same meaning of t. */
struct ioctx *ioctx_mmap_file(const char *fn, int t);
/* caller-owned memory of `t` bytes; reads and writes are clipped at the end of the buffer. */
struct ioctx *ioctx_from_mem(const uint8_t *ptr, size_t t);

#ifdef __cplusplus
}
#endif
#endif /* NANORQ_IOCTX_H */
