/*
 * io.c -- the three ioctx back-ends of include/io.h: stdio file, caller memory, memory-mapped file.
 * Same contract as the reference's lib/io.c (:12-80 file, :82-157 memory, :159-388 mmap): `t != 0`
 * opens an existing file read-only, `t == 0` creates/truncates a read-write file; the memory
 * context never owns the caller's buffer.  The mmap back-end maps the whole file (growing it in
 * 64 MiB steps when written past the end and trimming it on destroy) instead of sliding a 64 KiB
 * window; a failed mmap returns NULL / false instead of aborting the process.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/io.h"
#include "../../include/nanorq_batch.h"
#include "../../include/nanorq_hip.h"
#include "io_priv.h"

/* ------------------------------------------------------------------------------ stdio file ---- */
struct file_io { struct ioctx io; FILE *fp; };

static size_t f_read(struct ioctx *io, uint8_t *buf, size_t len) { return fread(buf, 1, len, ((struct file_io *)io)->fp); }
static size_t f_write(struct ioctx *io, const uint8_t *buf, size_t len) { return fwrite(buf, 1, len, ((struct file_io *)io)->fp); }
static bool f_seek(struct ioctx *io, const size_t off) { return fseeko(((struct file_io *)io)->fp, (off_t)off, SEEK_SET) == 0; }
static long f_tell(struct ioctx *io) { return (long)ftello(((struct file_io *)io)->fp); }
static size_t f_size(struct ioctx *io) {
  FILE *fp = ((struct file_io *)io)->fp;
  off_t here = ftello(fp);
  fseeko(fp, 0, SEEK_END);
  off_t end = ftello(fp);
  fseeko(fp, here, SEEK_SET);
  return end < 0 ? 0 : (size_t)end;
}
static void f_destroy(struct ioctx *io) {
  struct file_io *f = (struct file_io *)io;
  fclose(f->fp);
  free(f);
}

struct ioctx *ioctx_from_file(const char *fn, int t) {
  FILE *fp = fopen(fn, t ? "r" : "w+");
  if (!fp) return NULL;
  struct file_io *f = calloc(1, sizeof(*f));
  if (!f) { fclose(fp); return NULL; }
  f->fp = fp;
  f->io.read = f_read; f->io.write = f_write; f->io.seek = f_seek; f->io.size = f_size;
  f->io.tell = f_tell; f->io.destroy = f_destroy;
  f->io.seekable = true;
  f->io.writable = (t == 0);
  return &f->io;
}

/* --------------------------------------------------------------------------- caller memory ---- */
struct mem_io { struct ioctx io; uint8_t *base; size_t pos, len; int pinned; /* 0 = caller memory, 1 = page-locked and owned, 2 = caller memory registered, 3 = registered on first use by the object layer (ioctx_dma_region_auto), -1 = that was tried and refused */ };

static size_t m_clip(struct mem_io *m, size_t len) { return (m->pos + len > m->len) ? m->len - m->pos : len; }
static size_t m_read(struct ioctx *io, uint8_t *buf, size_t len) {
  struct mem_io *m = (struct mem_io *)io;
  size_t n = m_clip(m, len);
  memcpy(buf, m->base + m->pos, n);
  m->pos += n;
  return n;
}
static size_t m_write(struct ioctx *io, const uint8_t *buf, size_t len) {
  struct mem_io *m = (struct mem_io *)io;
  size_t n = m_clip(m, len);
  memcpy(m->base + m->pos, buf, n);
  m->pos += n;
  return n;
}
static bool m_seek(struct ioctx *io, const size_t off) {
  struct mem_io *m = (struct mem_io *)io;
  if (off >= m->len) return false;
  m->pos = off;
  return true;
}
static long m_tell(struct ioctx *io) { return (long)((struct mem_io *)io)->pos; }
static size_t m_size(struct ioctx *io) { return ((struct mem_io *)io)->len; }
static void m_destroy(struct ioctx *io) {
  struct mem_io *m = (struct mem_io *)io;
  if (m->pinned == 3) nrq_host_unregister(m->base); /* (the page lock the object layer took; the memory stays the caller's) */
  free(io);
}

struct ioctx *ioctx_from_mem(const uint8_t *ptr, size_t sz) {
  struct mem_io *m = calloc(1, sizeof(*m));
  if (!m) return NULL;
  m->base = (uint8_t *)ptr;
  m->len = sz;
  m->io.read = m_read; m->io.write = m_write; m->io.seek = m_seek; m->io.size = m_size;
  m->io.tell = m_tell; m->io.destroy = m_destroy;
  m->io.seekable = true;
  m->io.writable = true;
  return &m->io;
}

/* page-locked variants (include/nanorq_batch.h): the object layer moves whole blocks between such a context and the GPU
 * with asynchronous DMA copies -- no staging copy on the host, no seek/read loop (reference: transfer_esi and
 * load_symbol_matrix, lib/nanorq.c:148-182, over lib/io.c:82-157) */
static void pm_destroy(struct ioctx *io) {
  struct mem_io *m = (struct mem_io *)io;
  if (m->pinned == 1) nrq_host_free_pinned(m->base);
  else if (m->pinned == 2) nrq_host_unregister(m->base);
  free(m);
}
struct ioctx *ioctx_from_pinned_mem(size_t sz) {
  void *p = NULL;
  if (nrq_host_alloc_pinned(sz ? sz : 1, &p) != 0) return NULL;
  struct ioctx *io = ioctx_from_mem(p, sz);
  if (!io) { nrq_host_free_pinned(p); return NULL; }
  ((struct mem_io *)io)->pinned = 1;
  io->destroy = pm_destroy;
  return io;
}
struct ioctx *ioctx_from_registered_mem(uint8_t *ptr, size_t sz) {
  if (!ptr || nrq_host_register(ptr, sz) != 0) return NULL;
  struct ioctx *io = ioctx_from_mem(ptr, sz);
  if (!io) { nrq_host_unregister(ptr); return NULL; }
  ((struct mem_io *)io)->pinned = 2;
  io->destroy = pm_destroy;
  return io;
}
uint8_t *ioctx_mem_base(struct ioctx *io) {
  if (!io || io->read != m_read) return NULL;
  return ((struct mem_io *)io)->base;
}
bool ioctx_dma_region(struct ioctx *io, uint8_t **base, size_t *len) {
  if (!io || io->read != m_read || io->write != m_write) return false; /* (a caller that replaced the vtable gets the generic path) */
  struct mem_io *m = (struct mem_io *)io;
  if (m->pinned <= 0) return false;
  *base = m->base;
  *len = m->len;
  return true;
}

/* An ordinary memory context (ioctx_from_mem, the only kind the reference's callers make) whose blocks the per-block calls
 * move: the first use page-locks the caller's region in place, so that nanorq_generate_symbols can hand the block's bytes to
 * the copy engine instead of copying them into a staging buffer first (reference load_symbol_matrix, lib/nanorq.c:175-182).
 * The lock is dropped in destroy(); NANORQ_HIP_AUTOPIN=0 switches this off; regions below 1 MiB are not worth the call. */
bool ioctx_dma_region_auto(struct ioctx *io, uint8_t **base, size_t *len) {
  if (ioctx_dma_region(io, base, len)) return true;
  if (!io || io->read != m_read || io->write != m_write || io->destroy != m_destroy) return false;
  struct mem_io *m = (struct mem_io *)io;
  if (m->pinned != 0 || m->len < ((size_t)1 << 20)) return false;
  const char *e = getenv("NANORQ_HIP_AUTOPIN");
  if ((e && *e == '0') || nrq_host_register(m->base, m->len) != 0) { m->pinned = -1; return false; }
  m->pinned = 3;
  *base = m->base;
  *len = m->len;
  return true;
}

/* ------------------------------------------------------------------------------ mmap file ---- */
#define MAP_STEP ((size_t)64 << 20)
struct map_io { struct ioctx io; int fd; uint8_t *base; size_t mapped, logical, pos; };

static bool map_grow(struct map_io *m, size_t need) {
  if (need <= m->mapped) return true;
  if (!m->io.writable) return false;
  size_t want = ((need + MAP_STEP - 1) / MAP_STEP) * MAP_STEP;
  if (ftruncate(m->fd, (off_t)want) != 0) return false;
  void *p = m->base ? mremap(m->base, m->mapped, want, MREMAP_MAYMOVE)
                    : mmap(NULL, want, PROT_READ | PROT_WRITE, MAP_SHARED, m->fd, 0);
  if (p == MAP_FAILED) return false;
  m->base = p;
  m->mapped = want;
  return true;
}
static size_t mp_read(struct ioctx *io, uint8_t *buf, size_t len) {
  struct map_io *m = (struct map_io *)io;
  if (m->pos >= m->logical) return 0;
  size_t n = (m->pos + len > m->logical) ? m->logical - m->pos : len;
  memcpy(buf, m->base + m->pos, n);
  m->pos += n;
  return n;
}
static size_t mp_write(struct ioctx *io, const uint8_t *buf, size_t len) {
  struct map_io *m = (struct map_io *)io;
  if (!m->io.writable || !map_grow(m, m->pos + len)) return 0;
  memcpy(m->base + m->pos, buf, len);
  m->pos += len;
  if (m->pos > m->logical) m->logical = m->pos;
  return len;
}
static bool mp_seek(struct ioctx *io, const size_t off) {
  struct map_io *m = (struct map_io *)io;
  if (!m->io.writable && off >= m->logical) return false;
  if (m->io.writable && !map_grow(m, off + 1)) return false;
  m->pos = off;
  return true;
}
static long mp_tell(struct ioctx *io) { return (long)((struct map_io *)io)->pos; }
static size_t mp_size(struct ioctx *io) { return ((struct map_io *)io)->logical; }
static void mp_destroy(struct ioctx *io) {
  struct map_io *m = (struct map_io *)io;
  if (m->base) munmap(m->base, m->mapped);
  if (m->io.writable) { if (ftruncate(m->fd, (off_t)m->logical) != 0) { /* best effort */ } }
  close(m->fd);
  free(m);
}

struct ioctx *ioctx_mmap_file(const char *fn, int t) {
  int fd = t ? open(fn, O_RDONLY) : open(fn, O_RDWR | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return NULL;
  struct map_io *m = calloc(1, sizeof(*m));
  if (!m) { close(fd); return NULL; }
  m->fd = fd;
  m->io.writable = (t == 0);
  if (t) {
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); free(m); return NULL; }
    m->logical = (size_t)sb.st_size;
    if (m->logical) {
      void *p = mmap(NULL, m->logical, PROT_READ, MAP_SHARED, fd, 0);
      if (p == MAP_FAILED) { close(fd); free(m); return NULL; }
      m->base = p;
      m->mapped = m->logical;
    }
  } else if (!map_grow(m, 1)) {
    close(fd); free(m);
    return NULL;
  }
  m->io.read = mp_read; m->io.write = mp_write; m->io.seek = mp_seek; m->io.size = mp_size;
  m->io.tell = mp_tell; m->io.destroy = mp_destroy;
  m->io.seekable = true;
  return &m->io;
}
