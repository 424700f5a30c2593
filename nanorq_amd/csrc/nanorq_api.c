/*
 * nanorq_api.c -- the public nanorq object API (include/nanorq.h) on top of the HIP path.
 *
 * Host-side mirror of the reference's lib/nanorq.c: object transmission information, source-block
 * partitioning, per-block state, symbol ingestion and the ioctx traffic are plain C here, with the
 * same observable behaviour (citations per function, file:line in sleepybishop/nanorq).  The three
 * places where the reference does arithmetic on symbols --
 *     precode_matrix_gen/invert/intermediate      (nanorq.c:217-225, :616-624)
 *     decode_row for repair symbols and for gaps  (nanorq.c:184-204, :567-577)
 * -- go through include/nanorq_hip.h to the GPU.  Nothing in this file solves or generates symbols
 * on the CPU; when no GPU context can be created those calls fail (false / 0).
 *
 * Data movement.  Device buffers come from the context's pool (no hipMalloc /
 * hipFree per call), host staging rows are page-locked, and the batched calls (include/nanorq_batch.h) run a
 * three-stream pipeline: the upload of block n+1 and the download of block n-1 run beside the solve of
 * block n.  With page-locked memory contexts (ioctx_from_pinned_mem / _registered_mem) blocks move between the
 * object's memory and the GPU by DMA with no copy on the host at all, and a decoder fed from a page-locked
 * packet buffer keeps its received symbols on the device ("device-resident" blocks below).
 *
 * Devices.  NANORQ_HIP_DEVICES=0,1,... names the GPUs of the process (default: NANORQ_HIP_DEVICE or
 * device 0); each gets a context of its own -- streams, pools, plan caches -- and a lock.  Source blocks share nothing
 * (reference get_source_block, lib/nanorq.c:97-112), so block sbn of every object lives on device sbn mod N, and the batched
 * calls run one host thread per device over that device's blocks.  A device may be named twice (two contexts on one GPU).
 *
 * Threads: distinct nanorq objects may be used from different threads (as with the reference, which has no
 * globals); the GPU contexts behind them are guarded by a lock each.  One object is not thread-safe.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/nanorq.h"
#include "../../include/nanorq_batch.h"
#include "../../include/nanorq_ext.h"
#include "../../include/nanorq_hip.h"
#include "io_priv.h"

#define NRQ_Z_MAX 256u
#define NRQ_K_MAX 56403u
#define ENC_WINDOW 512u /* repair symbols fetched from the device per miss of the encode cache */
#define PIN_MIN ((size_t)64 << 10)        /* host buffers from this size on are page-locked */
#define CHUNK_BYTES ((size_t)192 << 20)   /* source bytes per pipeline step of the batched calls */
#define REPAIR_CHUNK_BYTES ((size_t)128 << 20) /* source bytes per solve launch of nanorq_repair_all (one planner run for all) */
#define UP_PIECE_BYTES ((size_t)48 << 20)      /* packet bytes per upload piece of the deferred ingestion */
#define NRQ_UP_EV 256
#define NRQ_UP_BLOB 16
#define NRQ_MAX_DEV 16

struct part { /* RFC 6330 section 4.4.1.2 Partition[I, J] */
  size_t IL, IS, JL, JS;
};

struct blockst {
  int di;             /* the device (index into g_dev) the block lives on: sbn mod the number of devices */
  uint32_t K, Kp, L;  /* source symbols; the table row the block is coded with and its L (block 0's unless NANORQ_EXT_PER_BLOCK_KP) */
  bool loaded, inverted;
  uint8_t *src;       /* K x T: source symbols (encoder) / received source symbols (decoder); allocated on first use */
  bool src_pinned;
  void *d_src;        /* device copy (K x T) */
  void *d_inter;      /* device: L x T intermediate symbols once solved */
  /* deferred encode of the per-block call (nanorq_generate_symbols): the upload goes into one of two device copies in turn,
   * so that the next block's bytes travel while this one is being solved; the solve is only enqueued */
  void *d_src2;       /* the other device copy */
  void *ev_up;        /* the upload of the current call */
  void *ev_read[2];   /* behind the solve that read d_src / d_src2 */
  bool ev_read_set[2];
  int cur;            /* which copy the solve in flight (or the last one) read: 0 = d_src, 1 = d_src2 */
  bool pending;       /* work of this block may still be in flight on the context's streams */
  /* decoder side */
  uint32_t *mask;     /* received-ESI bitmap */
  size_t mask_words;
  uint32_t have;      /* received source symbols (K - have = gaps) */
  uint32_t *rep_esi;  /* repair symbols in arrival order */
  uint8_t *rep_data;  /* their bytes -- host-resident blocks only */
  bool rep_pinned;
  size_t nrep, rep_cap;
  size_t spare;       /* max_esi - K: rows available beyond L (reference D sizing, nanorq.c:137-142) */
  /* device-resident decoder block: the received symbols were uploaded as a packet buffer and sorted into d_src / d_rep
   * on the GPU (nanorq_decoder_add_symbols with page-locked packets); no host copy is kept */
  bool dev;
  void *d_rep;        /* device: repair symbols in arrival order, capacity d_rep_cap symbols */
  size_t d_rep_cap;
  bool dirty;         /* received source symbols have not reached the output context yet */
  uint8_t *io_base;   /* page-locked output region the HOST has written every received source symbol of this block into, at ingestion
                       * (as the reference does: nanorq.c:478-509 writes a source symbol to the output when it arrives); NULL: not so.
                       * Then only the REPAIRED rows have to come down after the decode (nanorq_repair_all), a tenth of the block */
  uint32_t up_seq;    /* deferred ingestion: 1 + index (in the object's list for the block's device) of the last upload piece that
                       * carries symbols of this block; 0 = none in flight */
  /* decoded ahead of its nanorq_repair_block call, in the device batch of an earlier block's call (repair_ahead below):
   * 1 = recovered -- the rows are in `src`, the output context and the bitmap have not seen them yet --, 2 = rank deficient
   * with the symbols held then (a new symbol clears it) */
  int pre_state;
  uint32_t *pre_lost; /* the ESIs that were missing when the block was decoded ahead */
  uint32_t pre_n;
  /* encoder side: window of generated symbols */
  uint8_t *win;
  uint32_t win_isi0, win_n;
};

struct nanorq {
  size_t F, T, Al;         /* common OTI */
  size_t Z, N, Kt;         /* scheme specific */
  struct part src_part, sub_part;
  uint32_t Kp, S, H, L;    /* parameters of block 0, shared by every block (nanorq.c:289, :372) unless NANORQ_EXT_PER_BLOCK_KP */
  uint32_t max_esi;
  uint32_t flags;          /* NANORQ_EXT_* */
  bool precalc;
  pthread_mutex_t io_lock; /* an ioctx has one cursor: the device threads of a batched call on THIS object take turns at it
                            * (was process-wide: unrelated objects' output waited for each other) */
  struct blockst *blocks[NRQ_Z_MAX];
  /* deferred ingestion (nanorq_decoder_add_symbols_async), per device: the events behind the upload pieces still in
   * flight (a piece = copy of a stretch of the packet buffer + the kernel that sorts it into rows, on the upload stream)
   * and the staging buffers they go through; released by settle_uploads() */
  struct upstate {
    void *ev[NRQ_UP_EV];
    unsigned nev;
    void *blob[NRQ_UP_BLOB];
    unsigned nblob;
    void *pin[NRQ_UP_BLOB]; /* page-locked host arrays (destination addresses) the pieces' copies read */
    bool pin_cached[NRQ_UP_BLOB]; /* from host_alloc (goes back to its cache) rather than nrq_host_alloc_pinned */
    unsigned npin;
  } up[NRQ_MAX_DEV];
};

/* --------------------------------------------------------------- GPU contexts (process-wide) ---- */
struct devctx {
  nrq_ctx *c;
  int device;
  pthread_mutex_t lock; /* recursive: every entry point that touches the context holds it */
};
static struct devctx g_dev[NRQ_MAX_DEV];
static int g_ndev;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void ctx_init(void) {
  int devs[NRQ_MAX_DEV], n = 0;
  const char *e = getenv("NANORQ_HIP_DEVICES");
  if (e && *e) {
    while (*e && n < NRQ_MAX_DEV) {
      char *end = NULL;
      const long v = strtol(e, &end, 10);
      if (end == e) break;
      if (v >= 0) devs[n++] = (int)v;
      e = end;
      while (*e == ',' || *e == ' ') e++;
    }
  }
  if (n == 0) {
    e = getenv("NANORQ_HIP_DEVICE");
    devs[n++] = (e && *e) ? atoi(e) : 0;
  }
  pthread_mutexattr_t a;
  pthread_mutexattr_init(&a);
  pthread_mutexattr_settype(&a, PTHREAD_MUTEX_RECURSIVE);
  for (int i = 0; i < n; i++) {
    nrq_ctx *c = NULL;
    if (nrq_ctx_create(devs[i], NULL, &c) != 0 || !c) continue; /* a device that cannot be opened is left out */
    g_dev[g_ndev].c = c;
    g_dev[g_ndev].device = devs[i];
    pthread_mutex_init(&g_dev[g_ndev].lock, &a);
    g_ndev++;
  }
  pthread_mutexattr_destroy(&a);
}
static int ndev(void) {
  pthread_once(&g_once, ctx_init);
  return g_ndev;
}
static nrq_ctx *ctx(void) { return ndev() ? g_dev[0].c : NULL; } /* "is there a GPU at all" */
static nrq_ctx *dctx(int di) { return di < ndev() ? g_dev[di].c : NULL; }
static void gpu_lock(int di) { pthread_mutex_lock(&g_dev[di].lock); }
static void gpu_unlock(int di) { pthread_mutex_unlock(&g_dev[di].lock); }
size_t nanorq_devices(void) { return (size_t)ndev(); }
/* object-layer switches of nanorq_hip_option (not the context's): threads that book a packet batch, and from how many symbols on */
#define NRQ_BOOK_THREADS_MAX 32
static int g_book_threads = -1;        /* -1: NANORQ_HIP_BOOK_THREADS (up to 32), else half the cores this process may use, at most 16 */
static uint32_t g_book_min = 65536u;
static int g_host_rows = -1;          /* -1: NANORQ_HIP_HOST_ROWS (default on), see host_rows_on() */
int nanorq_hip_option(size_t dev, const char *name, long long value) {
  if (name && !strcmp(name, "book_threads")) { g_book_threads = value > 0 ? (int)(value > (long long)NRQ_BOOK_THREADS_MAX ? (long long)NRQ_BOOK_THREADS_MAX : value) : -1; return 0; }
  if (name && !strcmp(name, "book_min")) { g_book_min = value > 0 ? (uint32_t)value : 65536u; return 0; }
  if (name && !strcmp(name, "host_rows")) { g_host_rows = value != 0; return 0; }
  if (dev >= (size_t)ndev()) return -1;
  gpu_lock((int)dev);
  const int rc = nrq_ctx_set_option(g_dev[dev].c, name, value);
  gpu_unlock((int)dev);
  return rc;
}
void nanorq_trim(void); /* (below) */

/* -------------------------------------------------------------------------- small helpers ---- */
static size_t ceil_div(size_t a, size_t b) { return a / b + (a % b ? 1 : 0); }

static struct part partition(size_t I, size_t J) { /* nanorq.c:83-95 */
  struct part p = {0, 0, 0, 0};
  if (J == 0) return p;
  p.IL = ceil_div(I, J);
  p.IS = I / J;
  p.JL = I - p.IS * J;
  p.JS = J - p.JL;
  if (p.JL == 0) p.IL = 0;
  return p;
}

/* Page-locking memory costs about as much as copying it three times (hipHostMalloc pins at ~3 GB/s), so page-locked
 * host rows are recycled through a small cache: an object that is freed leaves its buffers to the next one. */
#define PIN_CACHE_SLOTS 512
#define PIN_CACHE_BYTES ((size_t)8 << 30)
static struct { void *p; size_t cap; } g_pin_cache[PIN_CACHE_SLOTS];
static size_t g_pin_cached;
static pthread_mutex_t g_pin_lock = PTHREAD_MUTEX_INITIALIZER;

static void *host_alloc(size_t bytes, bool *pinned) { /* zeroed; page-locked from PIN_MIN on when there is a GPU */
  void *p = NULL;
  *pinned = false;
  if (bytes >= PIN_MIN && ctx()) {
    pthread_mutex_lock(&g_pin_lock);
    int best = -1;
    for (int i = 0; i < PIN_CACHE_SLOTS; i++)
      if (g_pin_cache[i].p && g_pin_cache[i].cap >= bytes && g_pin_cache[i].cap <= bytes + bytes / 4 &&
          (best < 0 || g_pin_cache[i].cap < g_pin_cache[best].cap))
        best = i;
    if (best >= 0) {
      p = g_pin_cache[best].p;
      g_pin_cached -= g_pin_cache[best].cap;
      g_pin_cache[best].p = NULL;
    }
    pthread_mutex_unlock(&g_pin_lock);
    if (!p) {
      /* the capacity rides in front of the block (64 bytes keep the rows' alignment) */
      void *raw = NULL;
      if (nrq_host_alloc_pinned(bytes + 64, &raw) == 0 && raw) {
        *(size_t *)raw = bytes;
        p = (uint8_t *)raw + 64;
      }
    }
    if (p) {
      memset(p, 0, bytes);
      *pinned = true;
      return p;
    }
  }
  return calloc(bytes ? bytes : 1, 1);
}
static void host_free(void *p, bool pinned) {
  if (!p) return;
  if (!pinned) { free(p); return; }
  const size_t cap = *(size_t *)((uint8_t *)p - 64);
  pthread_mutex_lock(&g_pin_lock);
  if (g_pin_cached + cap <= PIN_CACHE_BYTES)
    for (int i = 0; i < PIN_CACHE_SLOTS; i++)
      if (!g_pin_cache[i].p) {
        g_pin_cache[i].p = p;
        g_pin_cache[i].cap = cap;
        g_pin_cached += cap;
        p = NULL;
        break;
      }
  pthread_mutex_unlock(&g_pin_lock);
  if (p) nrq_host_free_pinned((uint8_t *)p - 64);
}

static bool mask_get(const struct blockst *b, size_t id) {
  size_t w = id / 32;
  return w < b->mask_words && ((b->mask[w] >> (id % 32)) & 1u);
}
static void mask_set(struct blockst *b, size_t id) {
  size_t w = id / 32;
  if (w >= b->mask_words) {
    size_t nw = w + 1;
    uint32_t *m = realloc(b->mask, nw * sizeof(uint32_t));
    if (!m) return;
    memset(m + b->mask_words, 0, (nw - b->mask_words) * sizeof(uint32_t));
    b->mask = m;
    b->mask_words = nw;
  }
  if (id < b->K && !((b->mask[w] >> (id % 32)) & 1u)) b->have++;
  b->mask[w] |= 1u << (id % 32);
}
static void mask_clear(struct blockst *b, size_t id) {
  const size_t w = id / 32;
  if (w >= b->mask_words || !((b->mask[w] >> (id % 32)) & 1u)) return;
  b->mask[w] &= ~(1u << (id % 32));
  if (id < b->K) b->have--;
}
static size_t mask_gaps(const struct blockst *b, size_t until) { /* zero bits below `until` (bitmask_gaps, bitmask.c:47-77) */
  if (until == b->K) return b->K - b->have;
  size_t set = 0, full = until / 32;
  for (size_t w = 0; w < full && w < b->mask_words; w++) set += (size_t)__builtin_popcount(b->mask[w]);
  if ((until % 32) && full < b->mask_words) set += (size_t)__builtin_popcount(b->mask[full] & ((1u << (until % 32)) - 1u));
  return until - set;
}

size_t nanorq_block_symbols(nanorq *rq, uint8_t sbn) { /* nanorq.c:379-385 */
  if (sbn < rq->src_part.JL) return rq->src_part.IL;
  if (sbn - rq->src_part.JL < rq->src_part.JS) return rq->src_part.IS;
  return 0;
}
size_t nanorq_blocks(nanorq *rq) { return rq->src_part.JL + rq->src_part.JS; }
size_t nanorq_max_blocks(nanorq *rq) { (void)rq; return NRQ_Z_MAX; }
size_t nanorq_transfer_length(nanorq *rq) { return rq->F; }
size_t nanorq_symbol_size(nanorq *rq) { return rq->T; }
uint32_t nanorq_ext_flags(nanorq *rq) { return rq->flags; }
size_t nanorq_sub_blocks(nanorq *rq) { return rq->N; }

uint64_t nanorq_oti_common(nanorq *rq) {
  if (rq->flags & NANORQ_EXT_RFC_OTI) return ((uint64_t)rq->F << 24) | (rq->T & 0xffff); /* RFC 6330 section 3.3.2 */
  return ((uint64_t)rq->F << 24) | ((rq->T - 1) & 0xffff);                                  /* nanorq.c:309-315 */
}
uint32_t nanorq_oti_scheme_specific(nanorq *rq) {
  if (rq->flags & NANORQ_EXT_RFC_OTI) /* RFC 6330 section 3.3.3: Z | N | Al */
    return (uint32_t)((rq->Z & 0xff) << 24) | (uint32_t)((rq->N & 0xffff) << 8) | (uint32_t)rq->Al;
  return (uint32_t)((rq->Z - 1) << 24) | (uint32_t)((rq->N - 1) << 8) | (uint32_t)rq->Al; /* nanorq.c:317-324 */
}
uint32_t nanorq_tag(uint8_t sbn, uint32_t esi) { return ((uint32_t)sbn << 24) | (esi & 0x00ffffffu); }

static bool set_block_params(nanorq *rq) {
  uint32_t pr[10];
  size_t k0 = nanorq_block_symbols(rq, 0);
  if (k0 == 0) k0 = 1; /* an empty object still gets the smallest parameter row, as in the reference */
  if (nrq_params((uint32_t)k0, pr) != 0) return false;
  rq->Kp = pr[0]; rq->S = pr[2]; rq->H = pr[3]; rq->L = pr[5];
  return true;
}
size_t nanorq_block_kprime(nanorq *rq, uint8_t sbn) {
  const size_t k = nanorq_block_symbols(rq, sbn);
  uint32_t pr[10];
  if ((rq->flags & NANORQ_EXT_PER_BLOCK_KP) && k && nrq_params((uint32_t)k, pr) == 0) return pr[0];
  return rq->Kp;
}

/* ---------------------------------------------------------------------- construction ---- */
/* What the first block of an object would otherwise pay inside its first timed call: context creation, the code object, the
 * per-K' constants and (encoder) the encode plan -- on every device that will hold a block.  Failures are ignored here: the
 * calls that need the GPU report them.  NANORQ_HIP_LAZY=1 leaves all of it to first use. */
static void warm(nanorq *rq, int encoder) {
  const char *e = getenv("NANORQ_HIP_LAZY");
  if (e && *e == '1') return;
  const size_t k0 = nanorq_block_symbols(rq, 0), Z = nanorq_blocks(rq);
  if (!ndev() || k0 == 0) return;
  for (int d = 0; d < g_ndev && (size_t)d < (Z ? Z : 1); d++) {
    gpu_lock(d);
    (void)nrq_warm(g_dev[d].c, (uint32_t)k0, rq->Kp, encoder ? 2 : 0); /* (the encode plan of a big block is built on the device: waited for) */
    gpu_unlock(d);
  }
}

nanorq *nanorq_encoder_new_ext(size_t len, uint16_t T16, uint16_t K, uint16_t Z16, uint16_t N16, uint8_t Al8, uint32_t flags) {
  /* nanorq.c:241-296 */
  static const uint8_t aligns[4] = {8, 4, 2, 1};
  size_t T = T16, Z = Z16, Al = Al8;
  if (len > NANORQ_MAX_TRANSFER) return NULL;
  for (int a = 0; a < 4; a++)
    if (Al >= aligns[a]) { Al = aligns[a]; break; }
  if (Al == 0) Al = 1;
  if (T < Al) T = Al; else T -= T % Al;
  while (ceil_div(len, T) > (size_t)NRQ_Z_MAX * NRQ_K_MAX) {
    if (Al == 1 || T * Al > 0xffff) return NULL; /* the reference would spin / overflow here */
    T *= Al;
  }
  size_t Kt = ceil_div(len, T), Kn = K;
  if (Kt == 0) return NULL;
  if (K == 0) {
    Kn = Kt;
    if (Z == 0) {
      Z = 16;
      while (ceil_div(Kt, Z) > NRQ_K_MAX) Z++;
    }
    Kn = ceil_div(Kt, Z);
  }
  Z = ceil_div(Kt, Kn);
  if (Z == 0 || Z > NRQ_Z_MAX || ceil_div(Kt, Z) > NRQ_K_MAX) return NULL;
  size_t N = 1; /* nanorq.c:78 "disable interleaving" */
  if (flags & NANORQ_EXT_SUBBLOCKS) {
    N = N16 ? N16 : 1;
    if (N > T / Al) return NULL; /* a sub-symbol is at least Al bytes (RFC 6330 section 4.4.1.2) */
  }
  if ((flags & NANORQ_EXT_RFC_OTI) && (Z > 255 || T > 0xffff)) return NULL;
  nanorq *rq = calloc(1, sizeof(nanorq));
  if (!rq) return NULL;
  pthread_mutex_init(&rq->io_lock, NULL);
  rq->F = len; rq->T = T; rq->Al = Al; rq->Z = Z; rq->N = N; rq->Kt = Kt; rq->flags = flags;
  rq->src_part = partition(Kt, Z);
  rq->sub_part = partition(T / Al, rq->N);
  if (!set_block_params(rq)) { free(rq); return NULL; }
  warm(rq, 1);
  return rq;
}
nanorq *nanorq_encoder_new_ex(size_t len, uint16_t T, uint16_t K, uint16_t Z, uint8_t Al) {
  return nanorq_encoder_new_ext(len, T, K, Z, 1, Al, 0);
}
nanorq *nanorq_encoder_new(size_t len, uint16_t T, uint8_t Al) { return nanorq_encoder_new_ex(len, T, 0, 0, Al); }

nanorq *nanorq_decoder_new_ext(uint64_t common, uint32_t specific, uint32_t flags) { /* nanorq.c:336-377 */
  uint64_t F = common >> 24;
  size_t T, Z, N, Al = specific & 0xff;
  if (flags & NANORQ_EXT_RFC_OTI) {
    T = (size_t)(common & 0xffff);
    Z = (specific >> 24) & 0xff;
    N = (specific >> 8) & 0xffff;
  } else {
    T = (size_t)((common & 0xffff) + 1) & 0xffff; /* 16-bit wrap as in the reference */
    Z = ((specific >> 24) & 0xff) + 1;
    N = ((specific >> 8) & 0xffff) + 1;
  }
  if (F > NANORQ_MAX_TRANSFER) return NULL;
  if (T == 0 || Z == 0 || N == 0 || Al == 0 || T < Al || T % Al != 0) return NULL;
  if (!(flags & NANORQ_EXT_SUBBLOCKS) && (flags & NANORQ_EXT_RFC_OTI) && N != 1) return NULL;
  if (N > T / Al) return NULL;
  size_t Kt = ceil_div(F, T);
  if (ceil_div(Kt, Z) > NRQ_K_MAX) return NULL;
  nanorq *rq = calloc(1, sizeof(nanorq));
  if (!rq) return NULL;
  pthread_mutex_init(&rq->io_lock, NULL);
  rq->F = F; rq->T = T; rq->Al = Al; rq->Z = Z; rq->N = N; rq->Kt = Kt; rq->flags = flags;
  rq->src_part = partition(Kt, Z);
  rq->sub_part = partition(T / Al, N);
  if (!set_block_params(rq)) { free(rq); return NULL; }
  rq->max_esi = 2 * rq->Kp;
  warm(rq, 0);
  return rq;
}
nanorq *nanorq_decoder_new(uint64_t common, uint32_t specific) { return nanorq_decoder_new_ext(common, specific, 0); }

bool nanorq_set_max_esi(nanorq *rq, uint32_t max_esi) { /* nanorq.c:471-476 */
  if (!rq || max_esi >= (1u << 24) || max_esi < rq->Kp) return false;
  rq->max_esi = max_esi;
  return true;
}

/* deferred ingestion: wait for device di's upload pieces and release what they used */
static void settle_uploads(nanorq *rq, int di) {
  struct upstate *u = &rq->up[di];
  if (!u->nev && !u->nblob && !u->npin) return;
  nrq_ctx *c = dctx(di);
  if (c) {
    gpu_lock(di);
    nrq_stream_sync(c, 1);
    nrq_stream_sync(c, 3);
    for (unsigned i = 0; i < u->nblob; i++) nrq_dev_free(c, u->blob[i]);
    gpu_unlock(di);
  }
  for (unsigned i = 0; i < u->npin; i++) {
    if (u->pin_cached[i]) host_free(u->pin[i], true); else nrq_host_free_pinned(u->pin[i]);
  }
  for (unsigned i = 0; i < u->nev; i++) nrq_event_free(u->ev[i]);
  u->nev = u->nblob = u->npin = 0;
  for (unsigned sbn = 0; sbn < NRQ_Z_MAX; sbn++)
    if (rq->blocks[sbn] && rq->blocks[sbn]->di == di) rq->blocks[sbn]->up_seq = 0;
}
static void settle_all_uploads(nanorq *rq) {
  for (int d = 0; d < g_ndev; d++) settle_uploads(rq, d);
}

/* -------------------------------------------------------------------- per-block state ---- */
static struct blockst *get_block(nanorq *rq, uint8_t sbn) { /* nanorq.c:130-146 */
  if (rq->blocks[sbn]) return rq->blocks[sbn];
  struct blockst *b = calloc(1, sizeof(*b));
  if (!b) return NULL;
  b->K = (uint32_t)nanorq_block_symbols(rq, sbn);
  b->di = ndev() ? (int)(sbn % (unsigned)ndev()) : 0;
  b->Kp = rq->Kp;
  b->L = rq->L;
  if ((rq->flags & NANORQ_EXT_PER_BLOCK_KP) && b->K) {
    uint32_t pr[10];
    if (nrq_params(b->K, pr) == 0) { b->Kp = pr[0]; b->L = pr[5]; }
  }
  if (rq->max_esi) {
    b->mask_words = rq->max_esi / 32 + 1;
    b->mask = calloc(b->mask_words, sizeof(uint32_t));
    b->spare = rq->max_esi - b->K;
    if (!b->mask) { free(b); return NULL; }
  }
  rq->blocks[sbn] = b;
  return b;
}
/* the host rows of a block (zeroed), allocated when first needed: device-resident blocks never need them */
static bool ensure_src(nanorq *rq, struct blockst *b) {
  if (b->src) return true;
  b->src = host_alloc((size_t)(b->K ? b->K : 1) * rq->T, &b->src_pinned);
  return b->src != NULL;
}

static void drop_device(struct blockst *b) {
  nrq_ctx *c = g_ndev ? g_dev[b->di].c : NULL;
  if (c) {
    gpu_lock(b->di);
    if (b->pending) { nrq_stream_sync(c, 1); nrq_ctx_sync(c); } /* (a freed pool block may be handed out again at once) */
    if (b->d_src) nrq_dev_free(c, b->d_src);
    if (b->d_src2) nrq_dev_free(c, b->d_src2);
    if (b->d_inter) nrq_dev_free(c, b->d_inter);
    if (b->d_rep) nrq_dev_free(c, b->d_rep);
    gpu_unlock(b->di);
  }
  b->pending = false;
  b->d_src = b->d_src2 = b->d_inter = b->d_rep = NULL;
  b->d_rep_cap = 0;
}

void nanorq_encoder_cleanup(nanorq *rq, uint8_t sbn) { /* nanorq.c:437-451 */
  struct blockst *b = rq->blocks[sbn];
  if (!b) return;
  if (b->up_seq) settle_uploads(rq, b->di);
  drop_device(b);
  nrq_event_free(b->ev_up); nrq_event_free(b->ev_read[0]); nrq_event_free(b->ev_read[1]);
  host_free(b->src, b->src_pinned);
  host_free(b->rep_data, b->rep_pinned);
  free(b->mask); free(b->rep_esi); free(b->win); free(b->pre_lost); free(b);
  rq->blocks[sbn] = NULL;
}

void nanorq_encoder_reset(nanorq *rq, uint8_t sbn) { /* nanorq.c:453-469 */
  struct blockst *b = rq->blocks[sbn];
  if (!b) return;
  /* The reference zeroes D here (nanorq.c:460).  Nothing reads a row of this block before it has been written again:
   * load_block fills every row (and zeroes what the object does not cover), the decoder's rows are written by add_symbol
   * and the rows that stay missing are named to the solver, never read -- so the host rows are left as they are (12.8 MB
   * of memset per call at K=10000), and the device copies stay allocated for the next use. */
  b->loaded = b->inverted = false;
  b->nrep = 0;
  b->win_n = 0;
  b->have = 0;
  b->pre_state = 0;
  if (b->dev && b->up_seq) settle_uploads(rq, b->di);
  if (b->dev) drop_device(b); /* a device-resident decoder block: its rows go back to the pool (the next packet batch allocates anew) */
  b->dev = b->dirty = false;
  if (b->mask) memset(b->mask, 0, b->mask_words * sizeof(uint32_t));
}

void nanorq_free(nanorq *rq) { /* nanorq.c:298-307 */
  if (!rq) return;
  settle_all_uploads(rq);
  for (unsigned sbn = 0; sbn < NRQ_Z_MAX; sbn++) nanorq_encoder_cleanup(rq, (uint8_t)sbn);
  pthread_mutex_destroy(&rq->io_lock);
  free(rq);
}

/* ---------------------------------------------------------------- object <-> symbol bytes ---- */
/* byte offset (in units of Al) of position `pos` of symbol `esi` (nanorq.c:97-128) */
static size_t symbol_offset(const nanorq *rq, uint8_t sbn, size_t pos, uint32_t K, uint32_t esi) {
  const size_t unit = rq->T / rq->Al;
  size_t sbloc = 0;
  if (sbn < rq->src_part.JL) sbloc = sbn * rq->src_part.IL * unit;
  else if (sbn - rq->src_part.JL < rq->src_part.JS)
    sbloc = rq->src_part.IL * rq->src_part.JL * unit + (sbn - rq->src_part.JL) * rq->src_part.IS * unit;
  const struct part *sp = &rq->sub_part;
  const size_t part_tot = sp->IL * sp->JL;
  if (pos < part_tot) {
    size_t sub = pos / sp->IL;
    return sbloc + sub * K * sp->IL + esi * sp->IL + pos % sp->IL;
  }
  size_t p2 = pos - part_tot, sub = p2 / sp->IS;
  return sbloc + part_tot * K + sub * K * sp->IS + esi * sp->IS + p2 % sp->IS;
}

/* move one symbol between `ptr` and the object stream (nanorq.c:148-173); bytes beyond F are skipped */
static size_t transfer_symbol(nanorq *rq, uint8_t sbn, uint32_t esi, uint32_t K, uint8_t *ptr, struct ioctx *io,
                              int out) {
  size_t moved = 0, col = 0;
  const size_t unit = rq->T / rq->Al, part_tot = rq->sub_part.IL * rq->sub_part.JL;
  for (size_t i = 0; i < unit;) {
    size_t offset = symbol_offset(rq, sbn, i, K, esi) * rq->Al;
    size_t sublen = (i < part_tot) ? rq->sub_part.IL : rq->sub_part.IS;
    size_t stride = sublen * rq->Al;
    if (sublen == 0) break;
    i += sublen;
    if (offset >= rq->F) { col += stride; continue; } /* (the piece lies beyond the object: skip it, keep the column) */
    if (io->seek(io, offset)) {
      if (offset + stride >= rq->F) stride = rq->F - offset;
      moved += out ? io->write(io, ptr + col, stride) : io->read(io, ptr + col, stride);
    }
    col += sublen * rq->Al;
  }
  return moved;
}

/* [off, off + len) of the object = the rows of block sbn, when they are one stretch (N == 1); bytes beyond F cut off */
static bool block_extent(const nanorq *rq, uint8_t sbn, uint32_t K, size_t *off, size_t *len) {
  if (rq->N != 1) return false;
  *off = symbol_offset(rq, sbn, 0, K, 0) * rq->Al;
  const size_t want = (size_t)K * rq->T;
  *len = *off >= rq->F ? 0 : (*off + want > rq->F ? rq->F - *off : want);
  return true;
}

static bool load_block(nanorq *rq, uint8_t sbn, struct blockst *b, struct ioctx *io) { /* nanorq.c:175-182 */
  if (!io || !ensure_src(rq, b)) return false;
  size_t off, len;
  if (block_extent(rq, sbn, b->K, &off, &len)) {
    /* no sub-blocking (the only case nanorq creates, nanorq.c:78): the block's symbols are one contiguous stretch
     * of the object -- one seek and one read instead of K of each; bytes beyond F (or not delivered) are zero */
    size_t got = 0;
    if (len && io->seek(io, off)) {
      while (got < len) {
        const size_t n = io->read(io, b->src + got, len - got);
        if (n == 0) break;
        got += n;
      }
    }
    memset(b->src + got, 0, (size_t)(b->K ? b->K : 1) * rq->T - got);
    return true;
  }
  memset(b->src, 0, (size_t)(b->K ? b->K : 1) * rq->T);
  for (uint32_t esi = 0; esi < b->K; esi++) transfer_symbol(rq, sbn, esi, b->K, b->src + (size_t)esi * rq->T, io, 0);
  return true;
}

/* ------------------------------------------------------------------------- encoding ---- */
bool nanorq_precalculate(nanorq *rq) { /* nanorq.c:393-401 */
  size_t k0 = nanorq_block_symbols(rq, 0);
  if (!ndev() || k0 == 0) return false;
  bool ok = true;
  const size_t Z = nanorq_blocks(rq);
  for (int d = 0; d < g_ndev && (size_t)d < (Z ? Z : 1); d++) { /* (every device that will hold a block keeps its own plan cache) */
    gpu_lock(d);
    ok = nrq_precalculate(g_dev[d].c, (uint32_t)k0, rq->Kp) == 0 && ok;
    gpu_unlock(d);
  }
  if (ok) rq->precalc = true;
  return ok;
}

/* the rows of block sbn as one stretch of a page-locked memory context (ordinary memory contexts are page-locked on first
 * use, io.c): the copy engine can read them where they are */
static bool io_dma_block(nanorq *rq, uint8_t sbn, uint32_t K, struct ioctx *io, uint8_t **p, size_t *len) {
  uint8_t *base;
  size_t rlen, off;
  if (!io || !ioctx_dma_region_auto(io, &base, &rlen)) return false;
  if (!block_extent(rq, sbn, K, &off, len) || off + *len > rlen) return false;
  *p = base + off;
  return true;
}

/* Deferred: the block's bytes are handed to the upload stream (straight from the caller's memory when that can be page-locked,
 * else from the block's page-locked host rows), the solve is enqueued behind them, and the call returns when the UPLOAD is
 * done -- the caller's bytes have been consumed, as in the reference, but nobody waits for the solve: the next call's upload
 * goes into the block's second device copy and runs beside it.  The results are waited for where bytes leave the library
 * (nanorq_encode -> fetch_symbol downloads on the context's stream, behind the solve; cleanup and free synchronise).
 * A solve cannot fail for mathematical reasons (the encode matrix is always invertible); a HIP error surfaces at those calls. */
bool nanorq_generate_symbols(nanorq *rq, uint8_t sbn, struct ioctx *io) { /* nanorq.c:206-232 */
  struct blockst *b = get_block(rq, sbn);
  if (!b) return false;
  if (b->inverted) return true;
  if (b->K == 0) return false;
  nrq_ctx *c = dctx(b->di);
  if (!c) return false; /* no GPU: no solve */
  const size_t T = rq->T, bytes = (size_t)b->K * T;
  uint8_t *hp = NULL;
  size_t hlen = 0;
  if (b->loaded || !io_dma_block(rq, sbn, b->K, io, &hp, &hlen)) {
    if (!b->loaded) b->loaded = load_block(rq, sbn, b, io);
    if (!b->loaded) return false;
    hp = b->src;
    hlen = bytes;
  }
  bool ok = false;
  gpu_lock(b->di);
  /* every block of an object is coded with block 0's K' (nanorq.c:289); a short block just has more padding */
  const int w = b->pending ? b->cur ^ 1 : 0; /* the copy the solve in flight is not reading */
  void **dp = w ? &b->d_src2 : &b->d_src;
  if (!*dp && nrq_dev_alloc(c, bytes, dp) != 0) goto out;
  if (!b->d_inter && nrq_dev_alloc(c, (size_t)b->L * T, &b->d_inter) != 0) goto out;
  if (!b->ev_up && nrq_event_new(c, &b->ev_up) != 0) goto out;
  if (!b->ev_read[w] && nrq_event_new(c, &b->ev_read[w]) != 0) goto out;
  if (b->ev_read_set[w] && nrq_stream_wait(c, 1, b->ev_read[w]) != 0) goto out; /* (the solve two calls ago read this copy) */
  if (nrq_copy_on(c, 1, *dp, hp, hlen) != 0) goto out;
  if (hlen < bytes && nrq_memset_on(c, 1, (uint8_t *)*dp + hlen, 0, bytes - hlen) != 0) goto out; /* beyond F: zero */
  if (nrq_event_record(c, b->ev_up, 1) != 0 || nrq_stream_wait(c, 0, b->ev_up) != 0) goto out;
  b->pending = true;
  if (nrq_encode_blocks(c, b->K, b->Kp, (uint32_t)T, 1, *dp, bytes, b->d_inter, (size_t)b->L * T, 0, NULL, NULL, 0) != 0) goto out;
  if (nrq_event_record(c, b->ev_read[w], 0) != 0) goto out;
  b->ev_read_set[w] = true;
  b->cur = w;
  if (nrq_event_sync(c, b->ev_up) != 0) goto out; /* the source bytes are on the device: the caller may reuse them */
  ok = true;
out:
  if (!ok && b->pending) { nrq_stream_sync(c, 1); nrq_ctx_sync(c); b->pending = false; }
  gpu_unlock(b->di);
  if (ok) { b->win_n = 0; b->inverted = true; }
  return ok;
}

/* repair symbol with internal id `isi` through a window of device-generated symbols */
static bool fetch_symbol(nanorq *rq, struct blockst *b, uint32_t isi, uint8_t *out) {
  const size_t T = rq->T;
  if (!(b->win_n && isi >= b->win_isi0 && isi < b->win_isi0 + b->win_n)) {
    nrq_ctx *c = dctx(b->di);
    if (!c || !b->d_inter) return false;
    if (!b->win && !(b->win = malloc((size_t)ENC_WINDOW * T))) return false;
    uint32_t isis[ENC_WINDOW], n = ENC_WINDOW;
    for (uint32_t k = 0; k < n; k++) isis[k] = isi + k;
    void *d_out = NULL;
    gpu_lock(b->di);
    bool ok = nrq_dev_alloc(c, (size_t)n * T, &d_out) == 0 &&
              nrq_gen_symbols(c, b->K, b->Kp, (uint32_t)T, 1, b->d_inter, (size_t)b->L * T, n, isis, d_out, (size_t)n * T) == 0 &&
              nrq_dev_download(c, b->win, d_out, (size_t)n * T) == 0;
    if (d_out) nrq_dev_free(c, d_out);
    gpu_unlock(b->di);
    if (!ok) return false;
    b->pending = false; /* (the download waited for the context's stream: the solve before it is done) */
    b->win_isi0 = isi;
    b->win_n = n;
  }
  memcpy(out, b->win + (size_t)(isi - b->win_isi0) * T, T);
  return true;
}

size_t nanorq_encode(nanorq *rq, void *data, uint32_t esi, uint8_t sbn, struct ioctx *io) { /* nanorq.c:403-435 */
  struct blockst *b = get_block(rq, sbn);
  if (!b) return 0;
  const size_t T = rq->T;
  if (esi < b->K) {
    /* before the solve: the source symbol itself (nanorq.c:414-421).  After it the reference regenerates the symbol from the
     * intermediate symbols, whatever has become of the io since (nanorq.c:410-413): a block that was solved straight out of
     * the caller's memory (no host copy) does the same -- LT(C, esi) on the device; with a host copy the loaded row is those
     * bytes (RFC 6330 is systematic). */
    if (b->inverted && !b->loaded) return fetch_symbol(rq, b, esi, data) ? T : 0;
    if (!b->loaded) b->loaded = load_block(rq, sbn, b, io);
    if (!b->loaded) return 0;
    memcpy(data, b->src + (size_t)esi * T, T);
    return T;
  }
  if (esi > (1u << 24) - 1) return 0;
  if (!b->inverted) b->inverted = nanorq_generate_symbols(rq, sbn, io);
  if (!b->inverted) return 0;
  return fetch_symbol(rq, b, esi + (b->Kp - b->K), data) ? T : 0;
}

/* ------------------------------------------------------------------------- decoding ---- */
static bool rep_reserve_host(nanorq *rq, struct blockst *b) { /* room for one more repair symbol (esi list + host bytes) */
  /* (a block that became device-resident in a batch that was then taken back is host-resident again with an ESI list but no
   * host bytes behind it: found by the fault-injection sweep under UBSan) */
  const bool no_bytes = !b->dev && !b->rep_data;
  if (b->nrep < b->rep_cap && !no_bytes) return true;
  const size_t nc = b->nrep < b->rep_cap ? b->rep_cap : (b->rep_cap ? b->rep_cap * 2 : 64), T = rq->T;
  uint32_t *e = realloc(b->rep_esi, nc * sizeof(uint32_t));
  if (!e) return false;
  b->rep_esi = e;
  if (!b->dev) {
    bool pin = false;
    uint8_t *d = host_alloc(nc * T, &pin);
    if (!d) return false;
    if (b->rep_data) memcpy(d, b->rep_data, b->nrep * T);
    host_free(b->rep_data, b->rep_pinned);
    b->rep_data = d;
    b->rep_pinned = pin;
  }
  b->rep_cap = nc;
  return true;
}

/* one symbol of a device-resident block arriving through the per-symbol call: straight to its device row */
static bool dev_put_row(nanorq *rq, struct blockst *b, void *d_row, const void *data) {
  nrq_ctx *c = dctx(b->di);
  if (!c) return false;
  gpu_lock(b->di);
  const bool ok = nrq_copy_on(c, 1, d_row, data, rq->T) == 0 && nrq_stream_sync(c, 1) == 0;
  gpu_unlock(b->di);
  return ok;
}
static bool dev_rep_reserve(nanorq *rq, struct blockst *b, size_t need, size_t keep, void **old_out); /* (below) */

int nanorq_decoder_add_symbol(nanorq *rq, void *data, uint32_t tag, struct ioctx *io) { /* nanorq.c:478-509 */
  uint8_t sbn = (uint8_t)(tag >> 24);
  uint32_t esi = tag & 0x00ffffffu;
  struct blockst *b = get_block(rq, sbn);
  if (!b || esi > rq->max_esi) return NANORQ_SYM_ERR;
  if (mask_gaps(b, b->K) == 0) return NANORQ_SYM_IGN;
  if (mask_get(b, esi)) return NANORQ_SYM_DUP;
  const size_t T = rq->T;
  if (b->dev && b->up_seq) settle_uploads(rq, b->di); /* (a deferred batch is still being sorted into this block's rows) */
  if (b->pre_state == 2) b->pre_state = 0; /* (the verdict formed ahead was for the symbols held then; a recovered block stays recovered) */
  if (esi < b->K) {
    if (b->dev) {
      if (!dev_put_row(rq, b, (uint8_t *)b->d_src + (size_t)esi * T, data)) return NANORQ_SYM_ERR;
    } else {
      if (!ensure_src(rq, b)) return NANORQ_SYM_ERR;
      memcpy(b->src + (size_t)esi * T, data, T);
    }
    if (io) transfer_symbol(rq, sbn, esi, b->K, data, io, 1);
  } else {
    if (!rep_reserve_host(rq, b)) return NANORQ_SYM_ERR;
    if (b->dev) {
      void *old = NULL;
      if (!dev_rep_reserve(rq, b, b->nrep + 1, b->nrep, &old)) return NANORQ_SYM_ERR;
      const bool ok = dev_put_row(rq, b, (uint8_t *)b->d_rep + b->nrep * T, data); /* (waits for the upload stream: `old` is idle then) */
      if (old) { gpu_lock(b->di); nrq_dev_free(dctx(b->di), old); gpu_unlock(b->di); }
      if (!ok) return NANORQ_SYM_ERR;
    } else {
      memcpy(b->rep_data + b->nrep * T, data, T);
    }
    b->rep_esi[b->nrep] = esi;
    b->nrep++;
  }
  mask_set(b, esi);
  return NANORQ_SYM_ADDED;
}

size_t nanorq_num_missing(nanorq *rq, uint8_t sbn) {
  struct blockst *b = get_block(rq, sbn);
  return b ? mask_gaps(b, b->K) : 0;
}
size_t nanorq_num_repair(nanorq *rq, uint8_t sbn) {
  struct blockst *b = get_block(rq, sbn);
  return b ? b->nrep : 0;
}

/* the missing source ESIs of a block, ascending; returns their number */
static uint32_t list_lost(const struct blockst *b, uint32_t *lost) {
  uint32_t g = 0;
  for (uint32_t w = 0; w * 32u < b->K; w++) {
    uint32_t miss = ~(w < b->mask_words ? b->mask[w] : 0u);
    if ((w + 1) * 32u > b->K) miss &= (b->K % 32u) ? ((1u << (b->K % 32u)) - 1u) : 0xFFFFFFFFu;
    while (miss) {
      const uint32_t bit = (uint32_t)__builtin_ctz(miss);
      miss &= miss - 1u;
      lost[g++] = w * 32u + bit;
    }
  }
  return g;
}
/* Repair symbols handed to the solver up front: the gaps plus a couple; the rest stays in reserve and is taken one at a
 * time only if the system turns out rank deficient (nrq_decode_blocks_lazy).  The reference makes every received
 * repair symbol a constraint row (nanorq.c:527-565: overhead = num_repair - num_gaps); the verdict is the same --
 * rows beyond rank L add nothing -- but M = L + overhead stays small: cheaper plans, and within the 16-bit slot range
 * of the planners however many surplus symbols a receiver has collected. */
static uint32_t rep_upfront(size_t gaps, size_t nrep) { return (uint32_t)(nrep - gaps > 2 ? gaps + 2 : nrep); }

/* write the rows of a device-resident block that the output context has not seen: all of them when the block is
 * complete (`all`), otherwise the runs of received rows only.  Enqueues on the download stream when `io` is a
 * page-locked context (the caller waits), else goes through the block's host rows. */
static bool flush_dev_block(nanorq *rq, uint8_t sbn, struct blockst *b, struct ioctx *io, bool all) {
  nrq_ctx *c = dctx(b->di);
  if (!c || !io || !b->d_src) return false;
  const size_t T = rq->T;
  uint8_t *base;
  size_t rlen, off, len;
  const bool dma = ioctx_dma_region(io, &base, &rlen) && block_extent(rq, sbn, b->K, &off, &len) && off + len <= rlen;
  if (!dma) {
    if (!ensure_src(rq, b) || nrq_copy_on(c, 2, b->src, b->d_src, (size_t)b->K * T) != 0 || nrq_stream_sync(c, 2) != 0) return false;
    pthread_mutex_lock(&rq->io_lock);
    for (uint32_t e = 0; e < b->K; e++)
      if (all || mask_get(b, e)) transfer_symbol(rq, sbn, e, b->K, b->src + (size_t)e * T, io, 1);
    pthread_mutex_unlock(&rq->io_lock);
    return true;
  }
  if (all) return nrq_copy_on(c, 2, base + off, b->d_src, len) == 0;
  for (uint32_t e = 0; e < b->K;) { /* runs of received rows */
    if (!mask_get(b, e)) { e++; continue; }
    uint32_t e1 = e;
    while (e1 < b->K && mask_get(b, e1)) e1++;
    const size_t o = (size_t)e * T, n0 = (size_t)(e1 - e) * T, n = o >= len ? 0 : (o + n0 > len ? len - o : n0);
    if (n && nrq_copy_on(c, 2, base + off + o, (uint8_t *)b->d_src + o, n) != 0) return false;
    e = e1;
  }
  return true;
}

/* The unchanged caller's decode loop (reference benchmark.c:143-151, decode.c: `for sbn: nanorq_repair_block`) asks for the
 * blocks of an object one call at a time; behind a GPU every call is a planner run the host waits for, a solve launch and a
 * PCIe round trip for ONE block.  So the first call decodes, in the same device batch, every other host-resident block of
 * the same size on the same device that is decodable right now -- one planner run and one solve launch for all of them --
 * and leaves their recovered rows in the blocks' host rows with the verdict (blockst::pre_state); the calls that follow
 * commit them (rows to the output context, bitmap) without touching the GPU.  What a call returns and writes is what the
 * reference's call does: the verdict is the block's own, a rank-deficient block stays retryable (a new symbol clears the
 * cached verdict), a recovered block stays recovered whatever arrives later (the solution is unique).
 * NANORQ_HIP_REPAIR_AHEAD=0 switches it off (one block per call). */
/* Received source symbols of device-resident blocks are written into a page-locked output region by the HOST at ingestion, and only
 * the repaired rows come down after the decode ("host_rows" option / NANORQ_HIP_HOST_ROWS=0: whole blocks come down, as in round 5) */
static bool host_rows_on(void) {
  if (g_host_rows < 0) {
    const char *e = getenv("NANORQ_HIP_HOST_ROWS");
    g_host_rows = !(e && *e == '0');
  }
  return g_host_rows != 0;
}
static bool repair_ahead_on(void) {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("NANORQ_HIP_REPAIR_AHEAD");
    on = (e && *e == '0') ? 0 : 1;
  }
  return on != 0;
}
static bool block_decodable(const struct blockst *b) {
  if (!b || b->K == 0) return false;
  const size_t gaps = mask_gaps(b, b->K);
  return gaps > 0 && b->nrep >= gaps && b->nrep - gaps <= b->spare;
}
/* recovered rows of a host-resident block: to the output context and into the bitmap (write_repair_rows, nanorq.c:579-589) */
static void commit_rows(nanorq *rq, uint8_t sbn, struct blockst *b, const uint32_t *lost, size_t n, struct ioctx *io) {
  for (size_t k = 0; k < n; k++) {
    if (io) transfer_symbol(rq, sbn, lost[k], b->K, b->src + (size_t)lost[k] * rq->T, io, 1);
    mask_set(b, lost[k]);
  }
}
/* decode the host-resident blocks `sbns` (equal K, one device) in one device batch; status[k] = 1 recovered (rows in the
 * block's host rows), 0 rank deficient; lost lists at lost + k * lost_cap.  false: a device error (nothing is valid). */
static bool decode_host_blocks(nanorq *rq, int di, const unsigned *sbns, unsigned n, uint32_t *lost, uint32_t *nlost, size_t lost_cap,
                               int *status) {
  nrq_ctx *c = dctx(di);
  const size_t T = rq->T;
  size_t rep_cap = 0;
  for (unsigned k = 0; k < n; k++)
    if (rq->blocks[sbns[k]]->nrep > rep_cap) rep_cap = rq->blocks[sbns[k]]->nrep;
  uint32_t *resi = calloc((size_t)n * rep_cap, sizeof(uint32_t)), *nuse = calloc(n, sizeof(uint32_t)), *navail = calloc(n, sizeof(uint32_t));
  uint64_t *sv = calloc(n, sizeof(uint64_t)), *rv = calloc(n, sizeof(uint64_t));
  void **tmp = calloc(n, sizeof(void *));
  bool ok = resi && nuse && navail && sv && rv && tmp;
  const struct blockst *b0 = rq->blocks[sbns[0]];
  const size_t bytes = (size_t)b0->K * T;
  gpu_lock(di);
  for (unsigned k = 0; k < n && ok; k++) {
    struct blockst *b = rq->blocks[sbns[k]];
    nlost[k] = list_lost(b, lost + (size_t)k * lost_cap);
    nuse[k] = rep_upfront(nlost[k], b->nrep);
    navail[k] = (uint32_t)b->nrep;
    memcpy(resi + (size_t)k * rep_cap, b->rep_esi, b->nrep * sizeof(uint32_t));
    ok = ensure_src(rq, b) && (b->d_src || nrq_dev_alloc(c, bytes, &b->d_src) == 0) && nrq_dev_alloc(c, b->nrep * T, &tmp[k]) == 0 &&
         nrq_dev_upload_async(c, b->d_src, b->src, bytes) == 0 && nrq_dev_upload_async(c, tmp[k], b->rep_data, b->nrep * T) == 0;
    sv[k] = (uint64_t)(uintptr_t)b->d_src;
    rv[k] = (uint64_t)(uintptr_t)tmp[k];
  }
  ok = ok && nrq_decode_blocks_v(c, b0->K, b0->Kp, (uint32_t)T, n, sv, lost, nlost, (uint32_t)lost_cap, resi, nuse, navail, (uint32_t)rep_cap, rv,
                                 status, NULL) == 0;
  for (unsigned k = 0; k < n && ok; k++)
    if (status[k]) ok = nrq_dev_download_async(c, rq->blocks[sbns[k]]->src, rq->blocks[sbns[k]]->d_src, bytes) == 0;
  ok = (nrq_ctx_sync(c) == 0 || nrq_ctx_sync(c) == 0) && ok; /* (also on failure, and twice: the temporaries below may be in use) */
  for (unsigned k = 0; k < n && tmp; k++)
    if (tmp[k]) nrq_dev_free(c, tmp[k]);
  /* the blocks decoded AHEAD (k >= 1) have their rows back on the host (or nothing valid): their device copies go back to the
   * pool now instead of staying allocated until cleanup -- up to 2 GiB a call otherwise */
  for (unsigned k = 1; k < n; k++) {
    struct blockst *o = rq->blocks[sbns[k]];
    if (o->d_src) { nrq_dev_free(c, o->d_src); o->d_src = NULL; }
  }
  gpu_unlock(di);
  free(resi); free(nuse); free(navail); free(sv); free(rv); free(tmp);
  return ok;
}

bool nanorq_repair_block(nanorq *rq, struct ioctx *io, uint8_t sbn) { /* nanorq.c:591-631 */
  struct blockst *b = get_block(rq, sbn);
  if (!b) return false;
  if (b->up_seq) settle_uploads(rq, b->di); /* (symbols of a deferred batch still on their way) */
  if (b->pre_state == 1) { /* decoded in the batch of an earlier call: commit */
    commit_rows(rq, sbn, b, b->pre_lost, b->pre_n, io);
    b->pre_state = 0;
    return mask_gaps(b, b->K) == 0;
  }
  const size_t gaps = mask_gaps(b, b->K);
  nrq_ctx *c = dctx(b->di);
  if (gaps == 0) {
    if (b->dev && b->dirty && io && c) { /* received through the page-locked path and not written yet */
      gpu_lock(b->di);
      if (flush_dev_block(rq, sbn, b, io, true) && nrq_stream_sync(c, 2) == 0) b->dirty = false;
      gpu_unlock(b->di);
    }
    return true;
  }
  if (b->nrep < gaps) return false;
  const size_t overhead = b->nrep - gaps;
  if (overhead > b->spare) return false; /* D.rows < L + overhead in the reference */
  if (!c) return false;
  if (b->pre_state == 2) { b->pre_state = 0; return false; } /* found rank deficient ahead, and nothing has arrived since */
  const size_t T = rq->T;
  if (b->dev) {
    uint32_t *lost = malloc(gaps * sizeof(uint32_t));
    if (!lost) return false;
    list_lost(b, lost);
    bool ok = false;
    uint32_t nlost = (uint32_t)gaps, nuse = rep_upfront(gaps, b->nrep), navail = (uint32_t)b->nrep, used = 0;
    int status = 0;
    gpu_lock(b->di);
    const uint64_t sv = (uint64_t)(uintptr_t)b->d_src, rv = (uint64_t)(uintptr_t)b->d_rep;
    if (nrq_decode_blocks_v(c, b->K, b->Kp, (uint32_t)T, 1, &sv, lost, &nlost, nlost, b->rep_esi, &nuse, &navail, navail, &rv, &status, &used) == 0 &&
        status) {
      if (nrq_ctx_sync(c) == 0) { /* the recovered rows are in d_src */
        ok = true;
        if (io) {
          ok = flush_dev_block(rq, sbn, b, io, true) && nrq_stream_sync(c, 2) == 0;
          if (ok) b->dirty = false;
        }
        if (ok) /* (only now: a block counts as complete when its rows are where the caller will look for them) */
          for (size_t k = 0; k < gaps; k++) mask_set(b, lost[k]);
      }
    }
    gpu_unlock(b->di);
    free(lost);
    return ok;
  }
  /* host-resident: this block and, in the same device batch, the other blocks of its size that can be decoded now */
  unsigned sbns[NRQ_Z_MAX], n = 0;
  size_t lost_cap = gaps;
  sbns[n++] = sbn;
  if (repair_ahead_on()) {
    const size_t Z = nanorq_blocks(rq);
    size_t batch_bytes = ((size_t)b->K + b->nrep) * T; /* source rows AND the repair rows staged beside them */
    for (unsigned s2 = 0; s2 < Z; s2++) {
      const struct blockst *o = rq->blocks[s2];
      if (s2 == sbn || !o || o->dev || o->di != b->di || o->K != b->K || o->Kp != b->Kp || o->pre_state || o->up_seq || !block_decodable(o)) continue;
      if (batch_bytes + ((size_t)o->K + o->nrep) * T > ((size_t)2 << 30)) break; /* (bounded device footprint per call) */
      batch_bytes += ((size_t)o->K + o->nrep) * T;
      const size_t g2 = mask_gaps(o, o->K);
      if (g2 > lost_cap) lost_cap = g2;
      sbns[n++] = s2;
    }
  }
  uint32_t *lost = malloc((size_t)n * lost_cap * sizeof(uint32_t)), *nlost = calloc(n, sizeof(uint32_t));
  int *status = calloc(n, sizeof(int));
  bool ok = lost && nlost && status && decode_host_blocks(rq, b->di, sbns, n, lost, nlost, lost_cap, status);
  if (!ok && n > 1 && lost && nlost && status) {
    /* the BATCH failed (an allocation, an upload, the planner of some sibling): that says nothing about this block -- a
     * one-block-per-call decoder would not have touched the others.  Once more with the requested block alone, so that
     * `false` keeps meaning what it means in the reference: this block is rank deficient (nanorq.c:620-623) or the device failed it */
    n = 1;
    status[0] = 0;
    nlost[0] = 0;
    ok = decode_host_blocks(rq, b->di, sbns, 1, lost, nlost, lost_cap, status);
  }
  if (ok) {
    for (unsigned k = 1; k < n; k++) { /* the others: rows and verdict wait for their own call */
      struct blockst *o = rq->blocks[sbns[k]];
      if (!status[k]) { o->pre_state = 2; continue; }
      uint32_t *pl = realloc(o->pre_lost, (size_t)(nlost[k] ? nlost[k] : 1) * sizeof(uint32_t));
      if (!pl) continue; /* (no memory for the list: the block is decoded again by its own call) */
      memcpy(pl, lost + (size_t)k * lost_cap, (size_t)nlost[k] * sizeof(uint32_t));
      o->pre_lost = pl;
      o->pre_n = nlost[k];
      o->pre_state = 1;
    }
    ok = status[0] != 0; /* rank deficient: retry after more symbols (nanorq.c:620-623) */
    if (ok) {
      commit_rows(rq, sbn, b, lost, nlost[0], io);
      ok = mask_gaps(b, b->K) == 0;
    }
  }
  free(lost); free(nlost); free(status);
  return ok;
}

/* =============================================================== batched variants (nanorq_batch.h) ==== */
/* blocks come in at most two sizes (RFC 6330 section 4.4.1.2 partition: JL blocks of IL symbols, JS of IS) */
static uint32_t class_of(nanorq *rq, unsigned sbn) { return sbn < rq->src_part.JL ? 0u : 1u; }

void *nanorq_pinned_alloc(size_t bytes) {
  void *p = NULL;
  if (!ctx() || nrq_host_alloc_pinned(bytes, &p) != 0) return NULL;
  return p;
}
void nanorq_pinned_free(void *p) { nrq_host_free_pinned(p); }

/* the batched calls run one host thread per device over that device's blocks */
struct all_job {
  nanorq *rq;
  struct ioctx *io;
  int di;
  bool dma;
  uint8_t *base;
  size_t rlen;
  /* nanorq_decoder_add_symbols */
  const uint8_t *pk;
  const uint32_t *tags;
  const uint32_t *rix;
  uint32_t n;
  const size_t *nrep0;
  const uint8_t *touched;
  bool deferred; /* nanorq_decoder_add_symbols_async: enqueue only */
  void *early_blob;     /* ... and the packet buffer is already on its way into this device buffer, in pieces of early_piece */
  uint32_t early_piece; /* symbols, their copies' events at rq->up[di].ev[early_ev0 ...] */
  unsigned early_ev0;
  bool ok;
  /* nanorq_encode_range_all */
  uint8_t *out;
  uint32_t esi0;
};
static void for_devices(void *(*fn)(void *), struct all_job *tmpl, int nd) {
  struct all_job jobs[NRQ_MAX_DEV];
  pthread_t th[NRQ_MAX_DEV];
  bool started[NRQ_MAX_DEV];
  for (int d = 0; d < nd; d++) { jobs[d] = *tmpl; jobs[d].di = d; jobs[d].ok = true; started[d] = false; }
  for (int d = 1; d < nd; d++) started[d] = pthread_create(&th[d], NULL, fn, &jobs[d]) == 0;
  fn(&jobs[0]);
  for (int d = 1; d < nd; d++) {
    if (started[d]) pthread_join(th[d], NULL);
    else fn(&jobs[d]); /* (no thread to be had: one after the other) */
  }
  tmpl->ok = true;
  for (int d = 0; d < nd; d++) tmpl->ok = tmpl->ok && jobs[d].ok;
}
static int devices_for(nanorq *rq) { /* devices that hold blocks of this object */
  const size_t Z = nanorq_blocks(rq);
  return (size_t)ndev() < Z ? ndev() : (int)Z;
}

/* Encoder, all blocks: per device a pipeline of chunks of blocks -- chunk n+1 goes up (upload stream; straight out of a
 * page-locked context, or out of the blocks' page-locked host rows) while chunk n is solved (the context's stream).  Every
 * block keeps its intermediate symbols on the device, like after nanorq_generate_symbols. */
static void *gen_all_worker(void *arg) {
  struct all_job *j = arg;
  nanorq *rq = j->rq;
  const int di = j->di;
  nrq_ctx *c = g_dev[di].c;
  const size_t Z = nanorq_blocks(rq), T = rq->T;
  const bool dma = j->dma;
  uint8_t *base = j->base;
  /* j->out != NULL (nanorq_encode_range_all on blocks that are not solved yet): the sender's two legs as ONE pipeline -- behind
   * the solve of a chunk its repair symbols esi0 .. esi0+n-1 are generated and travel down (download stream) while the next
   * chunk travels up and is solved; the two directions of the link work at the same time. */
  const size_t per = (size_t)j->n * T;
  gpu_lock(di);
  for (uint32_t cls = 0; cls < 2; cls++) {
    /* the blocks of this size on this device that still need the solve */
    unsigned todo[NRQ_Z_MAX], n = 0;
    uint32_t K = 0, Kp = 0, L = 0;
    for (unsigned sbn = 0; sbn < Z; sbn++) {
      if (class_of(rq, sbn) != cls) continue;
      struct blockst *b = rq->blocks[sbn];
      if (!b || b->di != di || b->K == 0 || b->inverted) continue;
      if (!dma && !b->loaded) continue;
      K = b->K; Kp = b->Kp; L = b->L;
      todo[n++] = sbn;
    }
    if (!n) continue;
    const size_t sbytes = (size_t)K * T, ibytes = (size_t)L * T;
    unsigned C = (unsigned)(CHUNK_BYTES / sbytes);
    if (C < 1) C = 1;
    if (C > n) C = n;
    void *dsrc[2] = {NULL, NULL}, *up_done[2] = {NULL, NULL}, *solved[2] = {NULL, NULL};
    void *dout[2] = {NULL, NULL}, *gen_done[2] = {NULL, NULL}, *dl_done[2] = {NULL, NULL}, *d_isi = NULL;
    uint32_t *isis = NULL;
    bool ok = true;
    for (int i = 0; i < 2 && ok; i++)
      ok = nrq_dev_alloc(c, sbytes * C, &dsrc[i]) == 0 && nrq_event_new(c, &up_done[i]) == 0 && nrq_event_new(c, &solved[i]) == 0;
    if (j->out && ok) {
      /* the list of internal symbol ids is the same for every block of the class: uploaded once */
      isis = malloc((size_t)j->n * sizeof(uint32_t));
      ok = isis != NULL && nrq_dev_alloc(c, (size_t)j->n * sizeof(uint32_t), &d_isi) == 0;
      for (uint32_t q = 0; q < j->n && ok; q++) isis[q] = j->esi0 + q + (Kp - K);
      ok = ok && nrq_dev_upload(c, d_isi, isis, (size_t)j->n * sizeof(uint32_t)) == 0;
      for (int i = 0; i < 2 && ok; i++)
        ok = nrq_dev_alloc(c, per * C, &dout[i]) == 0 && nrq_event_new(c, &gen_done[i]) == 0 && nrq_event_new(c, &dl_done[i]) == 0;
    }
    unsigned done_blocks = 0;
    for (unsigned c0 = 0, step = 0; c0 < n && ok; c0 += C, step++) {
      const int i = (int)(step & 1u);
      const unsigned m = n - c0 < C ? n - c0 : C;
      if (step >= 2) ok = nrq_stream_wait(c, 1, solved[i]) == 0; /* the solve that read this buffer two steps ago */
      for (unsigned k = 0; k < m && ok;) {
        struct blockst *b = rq->blocks[todo[c0 + k]];
        uint8_t *dst = (uint8_t *)dsrc[i] + sbytes * k;
        if (dma) {
          /* consecutive blocks of a class are consecutive in the object: one copy for the run */
          size_t off = 0, len = 0, run = 1, tot;
          block_extent(rq, (uint8_t)todo[c0 + k], K, &off, &len);
          tot = len;
          while (k + run < m && todo[c0 + k + run] == todo[c0 + k] + run && tot == run * sbytes) {
            size_t o2 = 0, l2 = 0;
            block_extent(rq, (uint8_t)todo[c0 + k + run], K, &o2, &l2);
            tot += l2;
            run++;
          }
          ok = nrq_copy_on(c, 1, dst, base + off, tot) == 0;
          if (ok && tot < run * sbytes) ok = nrq_memset_on(c, 1, dst + tot, 0, run * sbytes - tot) == 0; /* padding beyond F */
          k += (unsigned)run;
        } else {
          ok = nrq_copy_on(c, 1, dst, b->src, sbytes) == 0;
          k++;
        }
      }
      uint64_t iv[NRQ_Z_MAX];
      for (unsigned k = 0; k < m && ok; k++) {
        struct blockst *b = rq->blocks[todo[c0 + k]];
        if (!b->d_inter) ok = nrq_dev_alloc(c, ibytes, &b->d_inter) == 0;
        iv[k] = (uint64_t)(uintptr_t)b->d_inter;
      }
      ok = ok && nrq_event_record(c, up_done[i], 1) == 0 && nrq_stream_wait(c, 0, up_done[i]) == 0 &&
           nrq_encode_blocks_v(c, K, Kp, (uint32_t)T, m, dsrc[i], sbytes, iv) == 0 && nrq_event_record(c, solved[i], 0) == 0;
      if (j->out && ok) {
        if (step >= 2) ok = nrq_stream_wait(c, 0, dl_done[i]) == 0; /* (the download out of this buffer two steps ago) */
        for (unsigned k = 0; k < m && ok; k++)
          ok = nrq_gen_symbols_dev(c, K, Kp, (uint32_t)T, 1, rq->blocks[todo[c0 + k]]->d_inter, ibytes, j->n, d_isi, (uint8_t *)dout[i] + per * k, per) == 0;
        ok = ok && nrq_event_record(c, gen_done[i], 0) == 0 && nrq_stream_wait(c, 2, gen_done[i]) == 0;
        for (unsigned k = 0; k < m && ok;) { /* consecutive blocks of this device: one copy for the run */
          unsigned run = 1;
          while (k + run < m && todo[c0 + k + run] == todo[c0 + k] + run) run++;
          ok = nrq_copy_on(c, 2, j->out + per * todo[c0 + k], (uint8_t *)dout[i] + per * k, per * run) == 0;
          k += run;
        }
        ok = ok && nrq_event_record(c, dl_done[i], 2) == 0;
      }
      if (ok) done_blocks = c0 + m;
    }
    ok = nrq_ctx_sync(c) == 0 && ok;
    nrq_stream_sync(c, 1);
    if (j->out) ok = nrq_stream_sync(c, 2) == 0 && ok;
    for (unsigned k = 0; k < done_blocks && ok; k++) {
      struct blockst *b = rq->blocks[todo[k]];
      b->win_n = 0;
      b->inverted = true;
    }
    if (!ok) j->ok = false;
    for (int i = 0; i < 2; i++) {
      if (dsrc[i]) nrq_dev_free(c, dsrc[i]);
      if (dout[i]) nrq_dev_free(c, dout[i]);
      nrq_event_free(up_done[i]);
      nrq_event_free(solved[i]);
      nrq_event_free(gen_done[i]);
      nrq_event_free(dl_done[i]);
    }
    if (d_isi) nrq_dev_free(c, d_isi);
    free(isis);
  }
  gpu_unlock(di);
  return NULL;
}
/* all blocks that are not solved yet; with `out`: their repair symbols esi0 .. esi0+n-1 too, block sbn's at out + sbn * n * T */
static size_t generate_all(nanorq *rq, struct ioctx *io, uint8_t *out, uint32_t esi0, uint32_t nsym) {
  const size_t Z = nanorq_blocks(rq);
  if (!ndev() || !io) return 0;
  struct all_job j;
  memset(&j, 0, sizeof(j));
  j.rq = rq; j.io = io; j.out = out; j.esi0 = esi0; j.n = nsym;
  j.dma = ioctx_dma_region(io, &j.base, &j.rlen) && rq->N == 1 && j.rlen >= rq->F;
  /* the blocks' state (and, without a page-locked context, their source rows: the context has ONE cursor) before the
   * device threads start */
  for (unsigned sbn = 0; sbn < Z; sbn++) {
    struct blockst *b = get_block(rq, (uint8_t)sbn);
    if (!b || b->K == 0 || b->inverted) continue;
    if (!j.dma && !b->loaded) b->loaded = load_block(rq, (uint8_t)sbn, b, io);
  }
  for_devices(gen_all_worker, &j, devices_for(rq));
  size_t done = 0;
  for (unsigned sbn = 0; sbn < Z; sbn++)
    if (rq->blocks[sbn] && rq->blocks[sbn]->inverted) done++;
  return (out && !j.ok) ? 0 : done;
}
size_t nanorq_generate_symbols_all(nanorq *rq, struct ioctx *io) { return generate_all(rq, io, NULL, 0, 0); }

/* Encoder: the repair symbols esi0 .. esi0+n-1 (esi0 >= K of every block) of ALL blocks in one go: `data` receives, block
 * after block, n * T bytes each.  One generation launch per block, queued without waiting, and one download per device
 * instead of a launch, a wait and a download per block (nanorq_encode_range). */
static void *range_all_worker(void *arg) {
  struct all_job *j = arg;
  nanorq *rq = j->rq;
  const int di = j->di;
  nrq_ctx *c = g_dev[di].c;
  const size_t Z = nanorq_blocks(rq), T = rq->T, per = (size_t)j->n * T;
  uint32_t *isis = malloc((size_t)j->n * sizeof(uint32_t));
  if (!isis) { j->ok = false; return NULL; }
  gpu_lock(di);
  /* device staging for a batch of blocks at a time */
  unsigned mine[NRQ_Z_MAX], nm = 0;
  for (unsigned sbn = 0; sbn < Z; sbn++)
    if (rq->blocks[sbn] && rq->blocks[sbn]->di == di) mine[nm++] = sbn;
  unsigned B = (unsigned)(((size_t)256 << 20) / (per ? per : 1));
  if (B < 1) B = 1;
  if (B > nm) B = nm;
  void *d_out = NULL;
  bool ok = nm == 0 || nrq_dev_alloc(c, per * B, &d_out) == 0;
  for (unsigned b0 = 0; b0 < nm && ok; b0 += B) {
    const unsigned m = nm - b0 < B ? nm - b0 : B;
    for (unsigned k = 0; k < m && ok; k++) {
      struct blockst *b = rq->blocks[mine[b0 + k]];
      for (uint32_t q = 0; q < j->n; q++) isis[q] = j->esi0 + q + (b->Kp - b->K);
      ok = nrq_gen_symbols(c, b->K, b->Kp, (uint32_t)T, 1, b->d_inter, (size_t)b->L * T, j->n, isis, (uint8_t *)d_out + per * k, per) == 0;
    }
    if (!ok) break;
    /* blocks of one device are every N-th of the object: their pieces of `data` are apart, one copy each (download stream) */
    ok = nrq_ctx_sync(c) == 0;
    for (unsigned k = 0; k < m && ok; k++)
      ok = nrq_copy_on(c, 2, j->out + per * mine[b0 + k], (uint8_t *)d_out + per * k, per) == 0;
    ok = nrq_stream_sync(c, 2) == 0 && ok;
  }
  if (d_out) nrq_dev_free(c, d_out);
  gpu_unlock(di);
  free(isis);
  j->ok = ok;
  return NULL;
}
size_t nanorq_encode_range_all(nanorq *rq, void *data, uint32_t esi0, uint32_t n, struct ioctx *io) {
  const size_t Z = nanorq_blocks(rq), T = rq->T;
  if (!ndev() || !n || (uint64_t)esi0 + n > (1u << 24)) return 0;
  for (unsigned sbn = 0; sbn < Z; sbn++) {
    struct blockst *b = get_block(rq, (uint8_t)sbn);
    if (!b || esi0 < b->K) return 0; /* repair symbols only */
  }
  {
    /* no block solved yet and a page-locked target (the sender's usual case): solve and generate as one pipeline */
    bool fresh = nrq_host_range_is_pinned(data, Z * (size_t)n * T) == 1; /* (every byte the downloads will write, not just the first) */
    for (unsigned sbn = 0; sbn < Z && fresh; sbn++) fresh = !rq->blocks[sbn]->inverted && rq->blocks[sbn]->K > 0;
    if (fresh) return generate_all(rq, io, data, esi0, n) == Z ? Z * (size_t)n * T : 0;
  }
  if (nanorq_generate_symbols_all(rq, io) != Z) return 0;
  struct all_job j;
  memset(&j, 0, sizeof(j));
  j.rq = rq; j.io = io; j.out = data; j.esi0 = esi0; j.n = n;
  for_devices(range_all_worker, &j, devices_for(rq));
  return j.ok ? Z * (size_t)n * T : 0;
}

size_t nanorq_encode_range(nanorq *rq, void *data, uint32_t esi0, uint32_t n, uint8_t sbn, struct ioctx *io) {
  struct blockst *b = get_block(rq, sbn);
  const size_t T = rq->T;
  if (!b || !n || (uint64_t)esi0 + n > (1u << 24)) return 0;
  uint8_t *out = data;
  uint32_t esi = esi0, left = n;
  /* source symbols: the loaded rows (nanorq_encode, esi < K) */
  while (left && esi < b->K) {
    if (nanorq_encode(rq, out, esi, sbn, io) != T) return 0;
    out += T; esi++; left--;
  }
  if (!left) return (size_t)n * T;
  if (!b->inverted) b->inverted = nanorq_generate_symbols(rq, sbn, io);
  nrq_ctx *c = dctx(b->di);
  if (!b->inverted || !c) return 0;
  /* repair symbols: generated on the device in one go, one download */
  uint32_t *isis = malloc((size_t)left * sizeof(uint32_t));
  void *d_out = NULL;
  gpu_lock(b->di);
  bool ok = isis && nrq_dev_alloc(c, (size_t)left * T, &d_out) == 0;
  if (ok) {
    for (uint32_t k = 0; k < left; k++) isis[k] = esi + k + (b->Kp - b->K);
    ok = nrq_gen_symbols(c, b->K, b->Kp, (uint32_t)T, 1, b->d_inter, (size_t)b->L * T, left, isis, d_out, (size_t)left * T) == 0 &&
         nrq_dev_download(c, out, d_out, (size_t)left * T) == 0;
  }
  if (d_out) nrq_dev_free(c, d_out);
  gpu_unlock(b->di);
  free(isis);
  return ok ? (size_t)n * T : 0;
}

/* room for `need` repair symbols in a device-resident block's d_rep; an outgrown buffer is handed back through *old_out
 * (the caller frees it once the copy of its first `keep` symbols -- enqueued on the upload stream -- is done) */
static bool dev_rep_reserve(nanorq *rq, struct blockst *b, size_t need, size_t keep, void **old_out) {
  nrq_ctx *c = dctx(b->di);
  *old_out = NULL;
  if (need <= b->d_rep_cap) return true;
  size_t nc = b->d_rep_cap ? b->d_rep_cap * 2 : (b->K / 8 > 64 ? b->K / 8 : 64);
  while (nc < need) nc *= 2;
  void *p = NULL;
  gpu_lock(b->di);
  bool ok = nrq_dev_alloc(c, nc * rq->T, &p) == 0;
  /* a deferred batch may still be sorting symbols into the rows that are copied (on the sorting stream, which nothing else
   * orders against the upload stream): the copy waits for the block's last piece */
  if (ok && b->d_rep && keep && b->up_seq && b->up_seq <= rq->up[b->di].nev) ok = nrq_stream_wait(c, 1, rq->up[b->di].ev[b->up_seq - 1u]) == 0;
  if (ok && b->d_rep && keep) ok = nrq_copy_on(c, 1, p, b->d_rep, keep * rq->T) == 0;
  if (!ok && p) nrq_dev_free(c, p);
  gpu_unlock(b->di);
  if (!ok) return false;
  *old_out = b->d_rep;
  b->d_rep = p;
  b->d_rep_cap = nc;
  return true;
}

/* Decoder, many symbols.  With a page-locked packet buffer (and a page-locked or no output context, no sub-blocking)
 * the packets go to the GPU in one DMA copy and a kernel sorts them into their rows (nrq_scatter_symbols); the host
 * does the bookkeeping of nanorq_decoder_add_symbol (nanorq.c:478-509) and touches no symbol byte.  Otherwise: the
 * per-symbol call in a loop.  With several devices every device uploads the buffer and keeps the symbols of its own blocks. */
enum { RIX_NONE = 0xFFFFFFFFu, RIX_SRC = 0xFFFFFFFEu };
static unsigned book_threads(void);
static uint32_t book_min(void);
#define NRQ_BOOK_THREADS 32u
struct addr_job { /* destination addresses of the symbols [k0, k1) of a batch that belong to device di (add_all_worker) */
  nanorq *rq;
  const uint32_t *tags, *rix;
  uint64_t *dst;
  int di;
  size_t T;
  uint32_t k0, k1, k_lo, k_hi;
  uint32_t last_k[NRQ_Z_MAX];
  bool any;
};
static void *addr_worker(void *arg) {
  struct addr_job *a = arg;
  uint32_t k_lo = a->k_lo, k_hi = a->k_hi, last_k[NRQ_Z_MAX]; /* (locals: the jobs lie side by side in memory) */
  memset(last_k, 0xFF, sizeof(last_k));
  bool any = false;
  for (uint32_t k = a->k0; k < a->k1; k++) {
    if (a->rix[k] == RIX_NONE) continue;
    const struct blockst *b = a->rq->blocks[(uint8_t)(a->tags[k] >> 24)];
    if (b->di != a->di) continue;
    a->dst[k] = a->rix[k] == RIX_SRC ? (uint64_t)(uintptr_t)((uint8_t *)b->d_src + (size_t)(a->tags[k] & 0x00ffffffu) * a->T)
                                     : (uint64_t)(uintptr_t)((uint8_t *)b->d_rep + (size_t)a->rix[k] * a->T);
    if (k < k_lo) k_lo = k;
    k_hi = k + 1u;
    last_k[(uint8_t)(a->tags[k] >> 24)] = k;
    any = true;
  }
  a->k_lo = k_lo; a->k_hi = k_hi; a->any = any;
  memcpy(a->last_k, last_k, sizeof(last_k));
  return NULL;
}
static void *add_all_worker(void *arg) {
  struct all_job *j = arg;
  nanorq *rq = j->rq;
  const int di = j->di;
  nrq_ctx *c = g_dev[di].c;
  const size_t T = rq->T;
  const uint32_t n = j->n;
  /* (deferred ingestion with the packets already on their way: the address list is built where the GPU will read it) */
  const size_t lbytes = ((size_t)n * 8u + 15u) & ~(size_t)15u;
  bool dst_pinned = false;
  uint64_t *dst = j->early_blob ? host_alloc(lbytes >= PIN_MIN ? lbytes : PIN_MIN, &dst_pinned) : calloc(n ? n : 1, sizeof(uint64_t));
  void *olds[NRQ_Z_MAX];
  unsigned nold = 0;
  bool ok = dst != NULL && (!j->early_blob || dst_pinned), any = false;
  uint32_t last_k[NRQ_Z_MAX]; /* per block: the last symbol of the batch that is the block's */
  memset(last_k, 0xFF, sizeof(last_k));
  gpu_lock(di);
  /* every touched block's repair rows to their final size (ONCE, keeping the rows held before the batch), then addresses */
  for (unsigned sbn = 0; sbn < NRQ_Z_MAX && ok; sbn++) {
    struct blockst *b = j->touched[sbn] ? rq->blocks[sbn] : NULL;
    if (!b || b->di != di || b->nrep == j->nrep0[sbn]) continue;
    void *old = NULL;
    ok = dev_rep_reserve(rq, b, b->nrep, j->nrep0[sbn], &old);
    if (old) olds[nold++] = old; /* (at most one per block) */
  }
  uint32_t k_lo = n, k_hi = 0; /* the stretch of the buffer that holds this device's symbols */
  if (ok) {
    /* where every symbol goes: a pure function of the books, so a big batch is cut into stretches for the booking threads
     * (a million symbols: ~2.5 ms on one thread, in front of everything nanorq_repair_all can start) */
    struct addr_job aj[NRQ_BOOK_THREADS];
    pthread_t ath[NRQ_BOOK_THREADS];
    bool astarted[NRQ_BOOK_THREADS];
    const unsigned P = n >= book_min() ? book_threads() : 1u;
    for (unsigned t = 0; t < P; t++) {
      aj[t] = (struct addr_job){.rq = rq, .tags = j->tags, .rix = j->rix, .dst = dst, .di = di, .T = T,
                                .k0 = (uint32_t)((uint64_t)n * t / P), .k1 = (uint32_t)((uint64_t)n * (t + 1u) / P), .k_lo = n, .k_hi = 0, .any = false};
      memset(aj[t].last_k, 0xFF, sizeof(aj[t].last_k));
      astarted[t] = false;
    }
    for (unsigned t = 1; t < P; t++) astarted[t] = pthread_create(&ath[t], NULL, addr_worker, &aj[t]) == 0;
    addr_worker(&aj[0]);
    for (unsigned t = 1; t < P; t++) {
      if (astarted[t]) pthread_join(ath[t], NULL);
      else addr_worker(&aj[t]);
    }
    for (unsigned t = 0; t < P; t++) { /* (stretches in ascending order: a later stretch's last symbol of a block wins) */
      if (!aj[t].any) continue;
      any = true;
      if (aj[t].k_lo < k_lo) k_lo = aj[t].k_lo;
      if (aj[t].k_hi > k_hi) k_hi = aj[t].k_hi;
      for (unsigned sbn = 0; sbn < NRQ_Z_MAX; sbn++)
        if (aj[t].last_k[sbn] != 0xFFFFFFFFu) last_k[sbn] = aj[t].last_k[sbn];
    }
  }
  struct upstate *u = &rq->up[di];
  if (j->early_blob) {
    /* The packets have been travelling since the call began (add_symbols_impl): what is left is to tell the GPU where every
     * symbol goes and to sort each piece into rows when its copy has landed -- kernels on a stream of their own, the address
     * list brought down by a kernel too (a copy would queue behind the packets). */
    const uint32_t piece = j->early_piece;
    void *d_dst = NULL;
    ok = ok && nrq_dev_alloc(c, lbytes, &d_dst) == 0 && nrq_ctl_copy(c, 3, d_dst, dst, lbytes) == 0;
    const unsigned ev_first = u->nev; /* the event behind the sort of piece i: u->ev[ev_first + i] */
    unsigned pi = 0;
    for (uint32_t k0 = 0; k0 < n && ok; k0 += piece, pi++) {
      const uint32_t m = n - k0 < piece ? n - k0 : piece;
      void *ev = NULL;
      ok = nrq_stream_wait(c, 3, u->ev[j->early_ev0 + pi]) == 0 &&
           nrq_scatter_symbols_dev(c, 3, (const uint8_t *)j->early_blob + (size_t)k0 * T, m, (uint32_t)T, (const uint64_t *)d_dst + k0) == 0 &&
           nrq_event_new(c, &ev) == 0 && nrq_event_record(c, ev, 3) == 0;
      if (ev) u->ev[u->nev++] = ev;
    }
    for (unsigned sbn = 0; sbn < NRQ_Z_MAX && ok; sbn++)
      if (last_k[sbn] != 0xFFFFFFFFu && rq->blocks[sbn]) rq->blocks[sbn]->up_seq = ev_first + last_k[sbn] / piece + 1u;
    u->blob[u->nblob++] = j->early_blob;
    if (d_dst) u->blob[u->nblob++] = d_dst;
    if (dst && dst_pinned) {
      u->pin_cached[u->npin] = true;
      u->pin[u->npin++] = dst; /* (read by the kernel above: released by settle_uploads) */
    } else {
      host_free(dst, dst_pinned);
    }
    if (!ok || nold) { /* (replaced repair rows are freed below: nothing may still be copying out of them) */
      nrq_stream_sync(c, 1);
      nrq_stream_sync(c, 3);
      settle_uploads(rq, di);
    }
    for (unsigned i = 0; i < nold; i++) nrq_dev_free(c, olds[i]);
    gpu_unlock(di);
    j->ok = ok;
    return NULL;
  }
  /* deferred: nothing is waited for -- every piece leaves an event on the upload stream, every block remembers its last
   * piece, and nanorq_repair_all lets the solve of a chunk of blocks wait for that piece only (the pieces behind it travel
   * while the chunk is solved and comes down).  Falls back to the waiting form when a block's repair rows had to be
   * replaced (the old rows are freed below) or the event / staging lists are full. */
  const uint32_t dpiece = (uint32_t)((UP_PIECE_BYTES / T) ? (UP_PIECE_BYTES / T) : 1);
  const bool deferred = j->deferred && nold == 0 && any && u->nblob + 2u <= NRQ_UP_BLOB && u->npin < NRQ_UP_BLOB &&
                        u->nev + (k_hi - k_lo + dpiece - 1u) / dpiece <= NRQ_UP_EV;
  /* Several devices: a device takes up ITS packets only.  Runs of consecutive packets of its blocks (a transfer whose packets
   * arrive block after block has one per block) are copied one by one into a compact staging buffer; with the whole stretch
   * uploaded by every device the host-to-device traffic was N times the batch.  Packets interleaved finer than that (more
   * than 4096 runs) keep the whole-stretch form. */
  if (any && ok && g_ndev > 1) {
    uint32_t nmine = 0, nruns = 0;
    for (uint32_t k = k_lo; k < k_hi; k++)
      if (dst[k]) { nmine++; if (k == k_lo || !dst[k - 1u]) nruns++; }
    if (nmine < k_hi - k_lo && nruns <= 4096u) {
      const uint32_t cap0 = (uint32_t)((CHUNK_BYTES / T) ? (CHUNK_BYTES / T) : 1), cap = nmine < cap0 ? nmine : cap0;
      uint64_t *cdst = malloc((size_t)cap * sizeof(uint64_t));
      void *d_blob = NULL;
      ok = cdst && nrq_dev_alloc(c, (size_t)cap * T, &d_blob) == 0;
      uint32_t fill = 0;
      for (uint32_t k = k_lo; k < k_hi && ok;) {
        if (!dst[k]) { k++; continue; }
        uint32_t k1 = k;
        while (k1 < k_hi && dst[k1]) k1++;
        while (k < k1 && ok) {
          const uint32_t take = k1 - k < cap - fill ? k1 - k : cap - fill;
          ok = nrq_copy_on(c, 1, (uint8_t *)d_blob + (size_t)fill * T, j->pk + (size_t)k * T, (size_t)take * T) == 0;
          memcpy(cdst + fill, dst + k, (size_t)take * sizeof(uint64_t));
          fill += take;
          k += take;
          if (fill == cap && ok) { ok = nrq_scatter_symbols(c, 1, d_blob, fill, (uint32_t)T, cdst) == 0; fill = 0; }
        }
      }
      if (fill && ok) ok = nrq_scatter_symbols(c, 1, d_blob, fill, (uint32_t)T, cdst) == 0;
      ok = nrq_stream_sync(c, 1) == 0 && ok;
      if (d_blob) nrq_dev_free(c, d_blob);
      free(cdst);
      if (j->deferred) settle_uploads(rq, di);
      any = false; /* done */
      for (unsigned i = 0; i < nold; i++) nrq_dev_free(c, olds[i]);
      gpu_unlock(di);
      free(dst);
      j->ok = ok;
      return NULL;
    }
  }
  if (any && ok) {
    /* packets up in pieces, each sorted into its rows as soon as it has landed (same stream: the order is the stream's) */
    void *d_blob = NULL;
    const uint32_t piece = deferred ? dpiece : (uint32_t)((CHUNK_BYTES / T) ? (CHUNK_BYTES / T) : 1), span = k_hi - k_lo;
    /* (deferred: two staging buffers in turn would let piece i+1 travel while piece i is sorted; the sort takes a fraction
     * of the copy's time, so one buffer and the stream's order are kept) */
    ok = nrq_dev_alloc(c, (size_t)(span < piece ? span : piece) * T, &d_blob) == 0;
    void *d_dst = NULL, *h_dst = NULL;
    if (deferred && ok) {
      /* the destination addresses of the whole stretch go down ONCE, from a page-locked copy (nrq_scatter_symbols stages a
       * list per call through two buffers and waits for the call before last: with the pieces' copies queued in front of
       * them that wait is the upload itself) */
      ok = nrq_host_alloc_pinned((size_t)span * 8u, &h_dst) == 0 && nrq_dev_alloc(c, (size_t)span * 8u, &d_dst) == 0;
      if (ok) {
        memcpy(h_dst, dst + k_lo, (size_t)span * 8u);
        ok = nrq_copy_on(c, 1, d_dst, h_dst, (size_t)span * 8u) == 0;
      }
    }
    for (uint32_t k0 = k_lo; k0 < k_hi && ok; k0 += piece) {
      const uint32_t m = k_hi - k0 < piece ? k_hi - k0 : piece;
      ok = nrq_copy_on(c, 1, d_blob, j->pk + (size_t)k0 * T, (size_t)m * T) == 0 &&
           (deferred ? nrq_scatter_symbols_dev(c, 1, d_blob, m, (uint32_t)T, (const uint64_t *)d_dst + (k0 - k_lo))
                     : nrq_scatter_symbols(c, 1, d_blob, m, (uint32_t)T, dst + k0)) == 0;
      if (deferred && ok) {
        void *ev = NULL;
        ok = nrq_event_new(c, &ev) == 0 && nrq_event_record(c, ev, 1) == 0;
        if (ev) u->ev[u->nev++] = ev;
        for (uint32_t k = k0; k < k0 + m && ok; k++)
          if (j->rix[k] != RIX_NONE && rq->blocks[(uint8_t)(j->tags[k] >> 24)]->di == di) rq->blocks[(uint8_t)(j->tags[k] >> 24)]->up_seq = u->nev;
      }
    }
    if (deferred && ok) {
      u->blob[u->nblob++] = d_blob; /* released by settle_uploads */
      u->blob[u->nblob++] = d_dst;
      u->pin_cached[u->npin] = false;
      u->pin[u->npin++] = h_dst;
    } else {
      ok = nrq_stream_sync(c, 1) == 0 && ok;
      if (d_blob) nrq_dev_free(c, d_blob);
      if (d_dst) nrq_dev_free(c, d_dst);
      if (h_dst) nrq_host_free_pinned(h_dst);
      if (j->deferred) settle_uploads(rq, di);
    }
  } else if (!j->deferred) {
    ok = nrq_stream_sync(c, 1) == 0 && ok;
  }
  for (unsigned i = 0; i < nold; i++) nrq_dev_free(c, olds[i]);
  gpu_unlock(di);
  free(dst);
  j->ok = ok;
  return NULL;
}
/* bookkeeping of a packet batch (add_symbols_impl), the blocks sbn mod P == t */
struct book_job {
  nanorq *rq;
  const uint8_t *p;
  const uint32_t *tags;
  uint32_t n;
  int *results;
  struct ioctx *io;
  uint32_t *rix;
  size_t *nrep0;
  uint8_t *touched, *newdev;
  bool early;
  unsigned t, P;
  size_t added;
  uint8_t *obase; /* the output context's page-locked region (NULL: none): source symbols of device-resident blocks are written there now */
};
static uint32_t book_min(void) { /* symbols from which a batch is booked by several threads: "book_min", else NANORQ_HIP_BOOK_MIN, else 65536 */
  static int env = -1;
  if (env < 0) {
    const char *e = getenv("NANORQ_HIP_BOOK_MIN");
    env = e && *e && atol(e) > 0 ? (int)atol(e) : 0;
  }
  return g_book_min != 65536u ? g_book_min : env ? (uint32_t)env : 65536u;
}
static unsigned book_threads(void) {
  static int n = -1;
  if (g_book_threads > 0) return (unsigned)g_book_threads;
  if (n < 0) {
    const char *e = getenv("NANORQ_HIP_BOOK_THREADS");
    long v = e && *e ? atol(e) : 0;
    if (v <= 0) {
      cpu_set_t set;
      v = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) / 2 : 1; /* (the cores this process may use: a rank of N has its slice) */
    }
    /* (default: at most 16 -- the threads also write the batch's source symbols into a page-locked output, 1.2 GB for 128 blocks
     * of K=8192: receiver pipeline 331 / 394 / 377 Gbit/s with 8 / 16 / 32 threads; NANORQ_HIP_BOOK_THREADS takes up to 32) */
    const long cap = e && *e && atol(e) > 0 ? (long)NRQ_BOOK_THREADS : 16;
    n = v < 1 ? 1 : v > cap ? (int)cap : (int)v;
  }
  return (unsigned)n;
}
/* a symbol from the packet buffer to its place in the page-locked output.  Streaming stores (NANORQ_HIP_NT_ROWS=1) spare the read of
 * every destination line a plain copy makes before overwriting it; measured on one box, receiver pipeline of 128 blocks of K=8192,
 * streaming / plain: 375, 366, 357 / 409, 401, 333 Gbit/s (the ingest call itself is shorter, 13-14.6 against 16-18 ms, and the
 * waiting calls gain, 238-249 against 214-221 -- but beside the packets' own way up the pipeline as a whole loses): off by default */
#if defined(__x86_64__) && defined(__SSE2__)
#include <emmintrin.h>
static int g_nt_rows = -1; /* NANORQ_HIP_NT_ROWS=1: streaming stores */
static void copy_row_out(uint8_t *dst, const uint8_t *src, size_t n) {
  if (g_nt_rows < 0) { const char *e = getenv("NANORQ_HIP_NT_ROWS"); g_nt_rows = e && *e == '1'; }
  if (g_nt_rows && (((uintptr_t)dst | (uintptr_t)src) & 15u) == 0 && (n & 63u) == 0) {
    for (size_t o = 0; o < n; o += 64) {
      const __m128i a = _mm_load_si128((const __m128i *)(src + o)), b = _mm_load_si128((const __m128i *)(src + o + 16)),
                    c = _mm_load_si128((const __m128i *)(src + o + 32)), d = _mm_load_si128((const __m128i *)(src + o + 48));
      _mm_stream_si128((__m128i *)(dst + o), a); _mm_stream_si128((__m128i *)(dst + o + 16), b);
      _mm_stream_si128((__m128i *)(dst + o + 32), c); _mm_stream_si128((__m128i *)(dst + o + 48), d);
    }
    return;
  }
  memcpy(dst, src, n);
}
#define COPY_ROW_FENCE() _mm_sfence()
#else
static void copy_row_out(uint8_t *dst, const uint8_t *src, size_t n) { memcpy(dst, src, n); }
#define COPY_ROW_FENCE() ((void)0)
#endif
static void *book_worker(void *arg) {
  struct book_job *j = arg;
  nanorq *rq = j->rq;
  const size_t T = rq->T;
  uint8_t owner[NRQ_Z_MAX];
  for (unsigned s_ = 0; s_ < NRQ_Z_MAX; s_++) owner[s_] = (uint8_t)(j->P > 1 ? s_ % j->P : j->t);
  size_t added = 0; /* (a local: the jobs lie side by side, and a counter bumped per symbol in each made their cache lines travel) */
  size_t xoff[NRQ_Z_MAX], xlen[NRQ_Z_MAX];
  uint8_t xknown[NRQ_Z_MAX]; /* 0 = not looked at, 1 = extent known, 2 = no extent */
  memset(xknown, 0, sizeof(xknown));
  for (uint32_t k = 0; k < j->n; k++) {
    const uint8_t sbn = (uint8_t)(j->tags[k] >> 24);
    if (owner[sbn] != j->t) continue; /* (a table, not sbn % P: a division per symbol and thread was the whole gain) */
    const uint32_t esi = j->tags[k] & 0x00ffffffu;
    struct blockst *b = get_block(rq, sbn);
    int r = NANORQ_SYM_ADDED;
    j->rix[k] = RIX_NONE;
    if (!b || esi > rq->max_esi) r = NANORQ_SYM_ERR;
    else if (mask_gaps(b, b->K) == 0) r = NANORQ_SYM_IGN;
    else if (mask_get(b, esi)) r = NANORQ_SYM_DUP;
    else if (!b->dev && (b->have || b->nrep)) {
      /* the block already holds symbols on the host (per-symbol calls came first): it stays host-resident */
      r = j->P > 1 ? NANORQ_SYM_ERR /* (cannot happen: such an object is booked by one thread) */
                   : nanorq_decoder_add_symbol(rq, (void *)(uintptr_t)(j->p + (size_t)k * T), j->tags[k], j->io);
    } else {
      if (!b->dev) { /* first symbol of the block: it becomes device-resident.  Its device rows are allocated and zeroed by the
                      * calling thread once the books are done (add_symbols_impl): no runtime call from a booking thread -- a
                      * thread's first one costs it the runtime's per-thread set-up, a millisecond under the device lock */
        b->dev = true;
        j->newdev[sbn] = 1;
        b->io_base = j->obase; /* (from its first symbol on, or not at all) */
      } else if (b->io_base != j->obase && !xknown[sbn]) {
        b->io_base = NULL; /* another output, or none, this time: the block's rows come down whole after the decode, as before */
      }
      if (!xknown[sbn]) xknown[sbn] = b->io_base && block_extent(rq, sbn, b->K, &xoff[sbn], &xlen[sbn]) ? 1 : 2;
      if (xknown[sbn] == 2) b->io_base = NULL;
      if (r == NANORQ_SYM_ADDED && !j->touched[sbn]) { j->touched[sbn] = 1; j->nrep0[sbn] = b->nrep; }
      if (r == NANORQ_SYM_ADDED && esi < b->K) {
        j->rix[k] = RIX_SRC;
        if (b->io_base) { /* the symbol's place in the output, written by this thread while the packets travel to the device */
          const size_t o = (size_t)esi * T;
          if (o < xlen[sbn]) copy_row_out(b->io_base + xoff[sbn] + o, j->p + (size_t)k * T, xlen[sbn] - o < T ? xlen[sbn] - o : T);
        }
      } else if (r == NANORQ_SYM_ADDED) {
        if (!rep_reserve_host(rq, b)) r = NANORQ_SYM_ERR;
        else {
          j->rix[k] = (uint32_t)b->nrep;
          b->rep_esi[b->nrep++] = esi;
        }
      }
      if (r == NANORQ_SYM_ADDED) mask_set(b, esi);
    }
    if (j->results) j->results[k] = r;
    if (r == NANORQ_SYM_ADDED) added++;
  }
  COPY_ROW_FENCE(); /* (the streaming stores of this thread are visible before the thread is joined) */
  j->added = added;
  return NULL;
}

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}
static bool diag_on(void) {
  static int on = -1;
  if (on < 0) on = getenv("NANORQ_HIP_DIAG") ? 1 : 0;
  return on != 0;
}
static size_t add_symbols_impl(nanorq *rq, const void *data, const uint32_t *tags, uint32_t n, int *results, struct ioctx *io, bool deferred) {
  size_t added = 0;
  const double t_in = now_us();
  double t_pin = 0, t_early = 0, t_book = 0, t_rows = 0;
  const uint8_t *p = data;
  const size_t T = rq->T;
  uint8_t *obase;
  size_t olen;
  const bool dma = ndev() && n >= 16 && rq->N == 1 && nrq_host_range_is_pinned(data, (size_t)n * T) && (!io || (ioctx_dma_region(io, &obase, &olen) && olen >= rq->F));
  /* Bookkeeping first, addresses afterwards: a repair symbol is recorded as (block, index in the block's repair rows) while
   * the loop runs; only when the batch's final repair count of every block is known are the blocks' device rows grown and
   * the indices turned into addresses (add_all_worker). */
  t_pin = now_us();
  uint32_t *rix = dma ? malloc((size_t)n * sizeof(uint32_t)) : NULL; /* per symbol: index of its repair row, or RIX_* */
  if (!rix) {
    for (uint32_t k = 0; k < n; k++) {
      /* (add_symbol copies; the const is cast away only because the per-symbol signature of the reference is void *) */
      const int r = nanorq_decoder_add_symbol(rq, (void *)(uintptr_t)(p + (size_t)k * T), tags[k], io);
      if (results) results[k] = r;
      if (r == NANORQ_SYM_ADDED) added++;
    }
    return added;
  }
  /* Deferred ingestion on one device: the packet buffer starts its way up NOW, piece by piece into one device buffer -- the
   * bookkeeping below (a million symbols: ~9 ms) runs beside the copies instead of in front of them. */
  void *early_blob = NULL;
  uint32_t early_piece = 0;
  unsigned early_ev0 = 0;
  if (deferred && g_ndev == 1) {
    struct upstate *u = &rq->up[0];
    const uint32_t piece = (uint32_t)((UP_PIECE_BYTES / T) ? (UP_PIECE_BYTES / T) : 1), np = (n + piece - 1u) / piece;
    nrq_ctx *c = dctx(0);
    if (c && u->nblob + 2u <= NRQ_UP_BLOB && u->npin < NRQ_UP_BLOB && u->nev + 2u * np <= NRQ_UP_EV) {
      gpu_lock(0);
      bool ok = nrq_dev_alloc(c, (size_t)n * T, &early_blob) == 0;
      early_ev0 = u->nev;
      for (uint32_t k0 = 0; k0 < n && ok; k0 += piece) {
        const uint32_t m = n - k0 < piece ? n - k0 : piece;
        void *ev = NULL;
        ok = nrq_copy_on(c, 1, (uint8_t *)early_blob + (size_t)k0 * T, p + (size_t)k0 * T, (size_t)m * T) == 0 && nrq_event_new(c, &ev) == 0 &&
             nrq_event_record(c, ev, 1) == 0;
        if (ev) u->ev[u->nev++] = ev;
      }
      if (!ok) { /* back to the staged path: what was enqueued is waited for and dropped */
        nrq_stream_sync(c, 1);
        if (early_blob) nrq_dev_free(c, early_blob);
        early_blob = NULL;
      } else {
        early_piece = piece;
      }
      gpu_unlock(0);
    }
  }
  t_early = now_us();
  size_t nrep0[NRQ_Z_MAX];                 /* repair symbols a touched block held before this batch */
  uint8_t touched[NRQ_Z_MAX], newdev[NRQ_Z_MAX];
  memset(touched, 0, sizeof(touched));
  memset(newdev, 0, sizeof(newdev));
  /* The bookkeeping of a symbol depends on the earlier symbols of ITS BLOCK only (bitmap, repair list, "block complete"): a big
   * batch is booked by several threads, thread t the blocks sbn mod P == t, every thread walking all tags in order -- a million
   * symbols took one thread ~7 ms, which a receiver's downloads had to wait out (they start behind the planner run, which needs the
   * pattern).  Not when some block of the object is host-resident with symbols in it: those go through the per-symbol call and the
   * output context, which has one cursor. */
  struct book_job bj[NRQ_BOOK_THREADS];
  unsigned P = 1;
  /* (... or when the threads have bytes to move: with a page-locked output they write the batch's source symbols to their places,
   * a copy of the whole batch -- 64 blocks of K=1000, 82 MB: 7.2 ms on one thread) */
  if (n >= book_min() || (io && host_rows_on() && g_ndev == 1 && (size_t)n * T >= ((size_t)8 << 20))) {
    P = book_threads();
    for (unsigned sbn = 0; sbn < NRQ_Z_MAX && P > 1; sbn++) {
      const struct blockst *b = rq->blocks[sbn];
      if (b && !b->dev && (b->have || b->nrep)) P = 1;
    }
  }
  pthread_t bth[NRQ_BOOK_THREADS];
  bool bstarted[NRQ_BOOK_THREADS];
  for (unsigned t = 0; t < P; t++) {
    bj[t] = (struct book_job){.rq = rq, .p = p, .tags = tags, .n = n, .results = results, .io = io, .rix = rix, .nrep0 = nrep0, .touched = touched,
                              .newdev = newdev, .early = early_blob != NULL, .t = t, .P = P, .added = 0, .obase = io && host_rows_on() && g_ndev == 1 ? obase : NULL}; /* (one device: a kernel of device 0 writes the
                               * repaired rows into the page-locked output; with several devices every decoded block comes down whole, as it did) */
    bstarted[t] = false;
  }
  for (unsigned t = 1; t < P; t++) bstarted[t] = pthread_create(&bth[t], NULL, book_worker, &bj[t]) == 0;
  book_worker(&bj[0]);
  for (unsigned t = 1; t < P; t++) {
    if (bstarted[t]) pthread_join(bth[t], NULL);
    else book_worker(&bj[t]); /* (no thread to be had: one after the other) */
  }
  for (unsigned t = 0; t < P; t++) added += bj[t].added;
  t_book = now_us();
  /* the blocks that became device-resident in this batch: their rows, zeroed in front of the sort of the packets into them */
  bool dev_ok = true;
  for (unsigned sbn = 0; sbn < NRQ_Z_MAX && dev_ok; sbn++) {
    struct blockst *b = newdev[sbn] ? rq->blocks[sbn] : NULL;
    if (!b) continue;
    nrq_ctx *c = dctx(b->di);
    gpu_lock(b->di);
    dev_ok = c && (b->d_src || nrq_dev_alloc(c, (size_t)b->K * T, &b->d_src) == 0) &&
             nrq_memset_on(c, early_blob ? 3 : 1, b->d_src, 0, (size_t)b->K * T) == 0;
    gpu_unlock(b->di);
  }
  t_rows = now_us();
  struct all_job j;
  memset(&j, 0, sizeof(j));
  j.rq = rq; j.io = io; j.pk = p; j.tags = tags; j.rix = rix; j.n = n; j.nrep0 = nrep0; j.touched = touched; j.deferred = deferred;
  j.early_blob = early_blob; j.early_piece = early_piece; j.early_ev0 = early_ev0;
  if (dev_ok) for_devices(add_all_worker, &j, ndev()); /* (also with nothing to put: the memsets of new blocks are waited for) */
  else {
    /* (a block's device rows could not be had: the batch is taken back below like one whose bytes did not arrive; what is already
     * on its way -- the early copies, memsets -- is waited for first) */
    for (int d = 0; d < ndev(); d++) {
      nrq_ctx *c = dctx(d);
      gpu_lock(d);
      if (c) { nrq_stream_sync(c, 1); nrq_stream_sync(c, 3); }
      if (d == 0 && early_blob && c) nrq_dev_free(c, early_blob);
      gpu_unlock(d);
    }
    j.ok = false;
  }
  if (!j.ok) {
    /* The bytes did not reach the device rows: take the batch's bookkeeping back, so that the decoder does not believe in
     * symbols it does not hold (they can be sent again), and say so symbol by symbol. */
    for (uint32_t k = 0; k < n; k++) {
      if (rix[k] == RIX_NONE) {
        /* a symbol reported DUP / IGN because an EARLIER symbol of this batch set its bit or completed its block: that
         * earlier symbol is taken back below, so the decoder holds neither copy -- say so */
        if (results && touched[(uint8_t)(tags[k] >> 24)] && (results[k] == NANORQ_SYM_DUP || results[k] == NANORQ_SYM_IGN))
          results[k] = NANORQ_SYM_ERR;
        continue;
      }
      struct blockst *b = rq->blocks[(uint8_t)(tags[k] >> 24)];
      mask_clear(b, tags[k] & 0x00ffffffu);
      if (results) results[k] = NANORQ_SYM_ERR;
      added--;
    }
    for (unsigned sbn = 0; sbn < NRQ_Z_MAX; sbn++) {
      struct blockst *b = touched[sbn] ? rq->blocks[sbn] : NULL;
      if (!b) continue;
      b->nrep = nrep0[sbn];
      if (newdev[sbn]) { b->dev = false; b->dirty = false; b->io_base = NULL; } /* (d_src stays allocated for the next attempt) */
    }
  } else {
    for (uint32_t k = 0; k < n; k++)
      if (rix[k] == RIX_SRC && !rq->blocks[(uint8_t)(tags[k] >> 24)]->io_base) rq->blocks[(uint8_t)(tags[k] >> 24)]->dirty = true;
  }
  free(rix);
  if (diag_on())
    fprintf(stderr, "[NANORQ_HIP_DIAG] add_symbols %u symbols, %u booking threads: page-lock check %.2f ms, early copies %.2f, books %.2f, new blocks' rows %.2f, "
            "addresses + enqueue %.2f, total %.2f ms\n", n, P, (t_pin - t_in) * 1e-3, (t_early - t_pin) * 1e-3, (t_book - t_early) * 1e-3,
            (t_rows - t_book) * 1e-3, (now_us() - t_rows) * 1e-3, (now_us() - t_in) * 1e-3);
  return added;
}

size_t nanorq_decoder_add_symbols(nanorq *rq, const void *data, const uint32_t *tags, uint32_t n, int *results, struct ioctx *io) {
  settle_all_uploads(rq); /* (a deferred batch before this one: this call's contract is "the buffer is free on return") */
  return add_symbols_impl(rq, data, tags, n, results, io, false);
}
/* The same, enqueue only (page-locked packet buffer; otherwise identical to the call above): bookkeeping and result codes
 * are final on return, the BYTES travel afterwards -- `data` must stay untouched until nanorq_repair_all,
 * nanorq_decoder_flush, nanorq_repair_block of a block it feeds, or nanorq_free has returned.  nanorq_repair_all starts its
 * planner run at once (it needs the reception pattern, not the symbols) and lets the solve of every chunk of blocks wait
 * for the upload piece that completes the chunk: ingest, solve and the way back of the decoded blocks overlap. */
size_t nanorq_decoder_add_symbols_async(nanorq *rq, const void *data, const uint32_t *tags, uint32_t n, int *results, struct ioctx *io) {
  return add_symbols_impl(rq, data, tags, n, results, io, true);
}

/* Decoder, all blocks that can be repaired: per device and block size a pipeline of chunks -- host-resident blocks go up chunk
 * by chunk (upload stream), the chunk is decoded (the context's stream), the decoded blocks come down (download stream)
 * beside the next chunk's decode; device-resident blocks skip the upload.  What comes down: whole blocks into a
 * page-locked output context, else the blocks' host rows and the repaired symbols from there through the context. */
static void *repair_all_worker(void *arg) {
  struct all_job *j = arg;
  nanorq *rq = j->rq;
  struct ioctx *io = j->io;
  const int di = j->di;
  nrq_ctx *c = g_dev[di].c;
  const size_t Z = nanorq_blocks(rq), T = rq->T;
  const double t_call = now_us();
  gpu_lock(di);
  /* Nothing is believed before it is known to have happened: a block's bitmap (and the "output has not seen these rows" flag of
   * a device-resident block) changes only after the copies that carry its rows have been WAITED for successfully.  (The fault
   * sweep of tools/sanitize_exercise.c: with a copy or a wait failing, a block used to count as complete while its rows had
   * not reached the output context.) */
  unsigned char wrote[NRQ_Z_MAX]; /* device-resident blocks whose rows were enqueued for the output: 1 = all rows, 2 = the received ones */
  memset(wrote, 0, sizeof(wrote));
  for (uint32_t cls = 0; cls < 2; cls++) {
    unsigned todo[NRQ_Z_MAX], n = 0;
    uint32_t K = 0, Kp = 0;
    size_t lost_cap = 0, rep_cap = 0;
    for (unsigned sbn = 0; sbn < Z; sbn++) {
      if (class_of(rq, sbn) != cls) continue;
      struct blockst *b = rq->blocks[sbn];
      if (!b || b->di != di || b->K == 0) continue;
      const size_t gaps = mask_gaps(b, b->K);
      if (gaps == 0) { /* complete; a device-resident block may still owe the output its received symbols */
        if (b->dev && b->dirty && io) {
          bool fl = true;
          if (b->up_seq && b->up_seq <= rq->up[di].nev) fl = nrq_stream_wait(c, 2, rq->up[di].ev[b->up_seq - 1u]) == 0; /* (still on their way up) */
          if (fl && flush_dev_block(rq, (uint8_t)sbn, b, io, true)) wrote[sbn] = 1; /* (committed behind the sync at the end) */
        }
        continue;
      }
      if (b->nrep < gaps || b->nrep - gaps > b->spare) continue; /* as nanorq_repair_block */
      K = b->K; Kp = b->Kp;
      if (gaps > lost_cap) lost_cap = gaps;
      if (b->nrep > rep_cap) rep_cap = b->nrep;
      todo[n++] = sbn;
    }
    if (!n) continue;
    const size_t sbytes = (size_t)K * T;
    /* ONE planner run for all blocks of the class (its fixed cost -- a launch the host waits for, ~3 ms whatever the number of
     * blocks -- is paid once), then the solve in chunks (nrq_decode_blocks_vc): the decoded blocks of chunk i come down while
     * the chunks behind it are still being solved, and host-resident blocks of chunk i+1 go up meanwhile */
    unsigned C = (unsigned)(REPAIR_CHUNK_BYTES / sbytes);
    if (C < 1) C = 1;
    if (C > n) C = n;
    const unsigned nch = (n + C - 1) / C;
    uint32_t *lost = calloc((size_t)n * lost_cap, sizeof(uint32_t)), *nlost = calloc(n, sizeof(uint32_t));
    uint32_t *resi = calloc((size_t)n * rep_cap, sizeof(uint32_t)), *nuse = calloc(n, sizeof(uint32_t)), *navail = calloc(n, sizeof(uint32_t));
    int *status = calloc(n, sizeof(int));
    bool *dev_done = calloc(n, sizeof(bool)); /* device-resident blocks whose recovered rows are (enqueued to be) where they belong */
    uint64_t *sv = calloc(n, sizeof(uint64_t)), *rv = calloc(n, sizeof(uint64_t));
    void **ev_done = calloc(nch, sizeof(void *)), **ev_up = calloc(nch, sizeof(void *)), **ev_dl = calloc(nch, sizeof(void *));
    bool *ev_up_borrowed = calloc(nch, sizeof(bool)); /* (an event of the deferred ingestion: released by settle_uploads) */
    void *tmp_rep[NRQ_Z_MAX]; /* device copies of host-resident blocks' repair symbols (freed at the end) */
    unsigned ntmp = 0;
    bool ok = lost && nlost && resi && nuse && navail && status && dev_done && sv && rv && ev_done && ev_up && ev_dl && ev_up_borrowed;
    for (unsigned ci = 0; ci < nch && ok; ci++) ok = nrq_event_new(c, &ev_done[ci]) == 0 && nrq_event_new(c, &ev_dl[ci]) == 0;
    for (unsigned c0 = 0, ci = 0; c0 < n && ok; c0 += C, ci++) { /* lists, and what has to go up, chunk by chunk */
      const unsigned m = n - c0 < C ? n - c0 : C;
      bool any_up = false;
      for (unsigned k = c0; k < c0 + m && ok; k++) {
        struct blockst *b = rq->blocks[todo[k]];
        nlost[k] = list_lost(b, lost + (size_t)k * lost_cap);
        nuse[k] = rep_upfront(nlost[k], b->nrep);
        navail[k] = (uint32_t)b->nrep;
        memcpy(resi + (size_t)k * rep_cap, b->rep_esi, b->nrep * sizeof(uint32_t));
        if (!b->dev) { /* host-resident: rows and repair symbols go up now */
          void *d_rep = NULL;
          ok = ensure_src(rq, b) && (b->d_src || nrq_dev_alloc(c, sbytes, &b->d_src) == 0) && nrq_dev_alloc(c, b->nrep * T, &d_rep) == 0 &&
               nrq_copy_on(c, 1, b->d_src, b->src, sbytes) == 0 && nrq_copy_on(c, 1, d_rep, b->rep_data, b->nrep * T) == 0;
          if (d_rep) tmp_rep[ntmp++] = d_rep;
          rv[k] = (uint64_t)(uintptr_t)d_rep;
          any_up = true;
        } else {
          rv[k] = (uint64_t)(uintptr_t)b->d_rep;
        }
        sv[k] = (uint64_t)(uintptr_t)b->d_src;
      }
      /* device-resident blocks fed by a deferred batch: the chunk waits for the last piece any of them is in */
      uint32_t seq = 0;
      for (unsigned k = c0; k < c0 + m; k++)
        if (rq->blocks[todo[k]]->up_seq > seq) seq = rq->blocks[todo[k]]->up_seq;
      if (seq > rq->up[di].nev) seq = 0;
      if (any_up && ok) {
        /* host-resident blocks in the chunk too: ONE event for both -- recorded on the upload stream behind this chunk's
         * copies, after that stream has been made to wait for the deferred piece (whose sort runs on another stream: the
         * upload stream's own order does not cover it) */
        if (seq) ok = nrq_stream_wait(c, 1, rq->up[di].ev[seq - 1u]) == 0;
        ok = ok && nrq_event_new(c, &ev_up[ci]) == 0 && nrq_event_record(c, ev_up[ci], 1) == 0;
      } else if (ok && seq) {
        ev_up[ci] = rq->up[di].ev[seq - 1u];
        ev_up_borrowed[ci] = true;
      }
    }
    /* Device-resident blocks whose received source symbols the host wrote into the output at ingestion (blockst::io_base): only
     * their REPAIRED rows come down -- by a kernel that writes each from its device row to its place in the page-locked output
     * (address pairs, built here: the lost lists are known before the decode; one launch per chunk behind its solve). */
    uint64_t *pairs = NULL, *d_pairs = NULL; /* (src, dst) per lost row, blocks in todo order */
    uint32_t *pair0 = calloc((size_t)n + 1u, sizeof(uint32_t)); /* first pair of block k; pair0[k + 1] - pair0[k] = its pairs (0: whole block comes down) */
    bool pairs_pinned = false;
    size_t pairs_bytes = 0;
    uint8_t *obase = NULL;
    size_t olen = 0;
    ok = ok && pair0 != NULL;
    if (ok && io && host_rows_on() && ioctx_dma_region(io, &obase, &olen)) {
      const uint64_t odev = nrq_host_device_address(obase);
      uint32_t np = 0;
      for (unsigned k = 0; k < n; k++) {
        struct blockst *b = rq->blocks[todo[k]];
        size_t off, len;
        pair0[k] = np;
        if (odev && b->dev && b->io_base == obase && !b->dirty && block_extent(rq, (uint8_t)todo[k], b->K, &off, &len) && len == (size_t)b->K * T &&
            off + len <= olen)
          np += nlost[k];
      }
      pair0[n] = np;
      pairs_bytes = (((size_t)np * 16u) + 15u) & ~(size_t)15u;
      if (np) {
        pairs = host_alloc(pairs_bytes >= PIN_MIN ? pairs_bytes : PIN_MIN, &pairs_pinned);
        void *dp = NULL;
        if (pairs && pairs_pinned && nrq_dev_alloc(c, pairs_bytes, &dp) == 0) {
          d_pairs = dp;
          for (unsigned k = 0; k < n; k++) {
            if (pair0[k + 1] == pair0[k]) continue;
            struct blockst *b = rq->blocks[todo[k]];
            size_t off, len;
            block_extent(rq, (uint8_t)todo[k], b->K, &off, &len);
            for (uint32_t q = 0; q < nlost[k]; q++) {
              const uint32_t e = lost[(size_t)k * lost_cap + q];
              pairs[2u * (pair0[k] + q)] = (uint64_t)(uintptr_t)b->d_src + (uint64_t)e * T;
              pairs[2u * (pair0[k] + q) + 1u] = odev + off + (uint64_t)e * T;
            }
          }
          ok = nrq_ctl_copy(c, 2, d_pairs, pairs, pairs_bytes) == 0; /* (download stream: in front of the launches that read it) */
        } else { /* no list to be had: every block comes down whole */
          for (unsigned k = 0; k <= n; k++) pair0[k] = 0;
        }
      }
    }
    const double t_lists = now_us();
    ok = ok && nrq_decode_blocks_vc(c, K, Kp, (uint32_t)T, n, sv, lost, nlost, (uint32_t)lost_cap, resi, nuse, navail, (uint32_t)rep_cap, rv, status,
                                    NULL, C, ev_done, ev_up) == 0;
    const double t_planned = now_us();
    bool any_host = false;
    for (unsigned c0 = 0, ci = 0; c0 < n && ok; c0 += C, ci++) { /* the downloads, each chunk behind its solve */
      const unsigned m = n - c0 < C ? n - c0 : C;
      ok = nrq_stream_wait(c, 2, ev_done[ci]) == 0;
      for (unsigned k = c0; k < c0 + m && ok; k++) {
        struct blockst *b = rq->blocks[todo[k]];
        const uint8_t sbn = (uint8_t)todo[k];
        if (!status[k]) { /* rank deficient: retry after more symbols (nanorq.c:620-623); what was received is written */
          if (b->dev && b->dirty && io && flush_dev_block(rq, sbn, b, io, false)) wrote[sbn] = 2;
          continue;
        }
        if (b->dev) {
          if (io && d_pairs && pair0[k + 1] > pair0[k]) /* (the received rows are in the output since they arrived: the repaired ones follow) */
            ok = nrq_move_rows_dev(c, 2, d_pairs + 2u * (size_t)pair0[k], pair0[k + 1] - pair0[k], (uint32_t)T) == 0;
          else if (io) ok = flush_dev_block(rq, sbn, b, io, true);
          if (ok) wrote[sbn] = 1; /* (bitmap and flag: behind the sync at the end) */
          dev_done[k] = ok;
        } else {
          ok = nrq_copy_on(c, 2, b->src, b->d_src, sbytes) == 0;
          any_host = true;
        }
      }
      ok = ok && nrq_event_record(c, ev_dl[ci], 2) == 0;
    }
    /* host-resident blocks: their repaired symbols go through the context once their rows are down */
    for (unsigned c0 = 0, ci = 0; c0 < n && ok && any_host; c0 += C, ci++) {
      const unsigned m = n - c0 < C ? n - c0 : C;
      ok = nrq_event_sync(c, ev_dl[ci]) == 0;
      pthread_mutex_lock(&rq->io_lock);
      for (unsigned k = c0; k < c0 + m && ok; k++) {
        struct blockst *b = rq->blocks[todo[k]];
        if (!status[k] || b->dev) continue;
        for (uint32_t q = 0; q < nlost[k]; q++) {
          const uint32_t e = lost[(size_t)k * lost_cap + q];
          if (io) transfer_symbol(rq, (uint8_t)todo[k], e, b->K, b->src + (size_t)e * T, io, 1);
          mask_set(b, e);
        }
      }
      pthread_mutex_unlock(&rq->io_lock);
    }
    /* (each wait is tried twice: the temporaries below must not be freed under work in flight whatever the first attempt said) */
    const double t_enq = now_us();
    const bool s0 = nrq_ctx_sync(c) == 0 || nrq_ctx_sync(c) == 0, s1 = nrq_stream_sync(c, 1) == 0 || nrq_stream_sync(c, 1) == 0;
    const double t_solved = now_us();
    const bool s2 = nrq_stream_sync(c, 2) == 0;
    if (diag_on())
      fprintf(stderr, "[NANORQ_HIP_DIAG] repair_all device %d class %u: %u blocks in %u chunks: lists %.2f ms, planner run + solves enqueued %.2f, downloads enqueued %.2f, "
              "uploads + solves done %.2f, downloads done %.2f ms after the call began\n", di, cls, n, nch, (t_lists - t_call) * 1e-3, (t_planned - t_call) * 1e-3,
              (t_enq - t_call) * 1e-3, (t_solved - t_call) * 1e-3, (now_us() - t_call) * 1e-3);
    if (ok && s0 && s1 && s2 && dev_done)
      for (unsigned k = 0; k < n; k++) { /* device-resident blocks: recovered, and (with an output context) written */
        if (!dev_done[k]) continue;
        struct blockst *b = rq->blocks[todo[k]];
        for (uint32_t q = 0; q < nlost[k]; q++) mask_set(b, lost[(size_t)k * lost_cap + q]);
      }
    else
      for (unsigned k = 0; k < n; k++) wrote[todo[k]] = 0;
    for (unsigned i = 0; i < ntmp; i++) nrq_dev_free(c, tmp_rep[i]);
    if (d_pairs) nrq_dev_free(c, d_pairs);
    if (pairs) host_free(pairs, pairs_pinned);
    free(pair0);
    for (unsigned ci = 0; ci < nch; ci++) {
      if (ev_done) nrq_event_free(ev_done[ci]);
      if (ev_up && !(ev_up_borrowed && ev_up_borrowed[ci])) nrq_event_free(ev_up[ci]);
      if (ev_dl) nrq_event_free(ev_dl[ci]);
    }
    free(ev_done); free(ev_up); free(ev_dl); free(ev_up_borrowed);
    free(lost); free(nlost); free(resi); free(nuse); free(navail); free(status); free(dev_done); free(sv); free(rv);
  }
  if ((nrq_stream_sync(c, 2) == 0 || nrq_stream_sync(c, 2) == 0) && io)
    for (unsigned sbn = 0; sbn < Z; sbn++) /* what was written is down: the output context has seen these blocks' rows */
      if (wrote[sbn] && rq->blocks[sbn]) rq->blocks[sbn]->dirty = false;
  settle_uploads(rq, di); /* (whatever a deferred batch still had in flight: the blocks that waited for it are done) */
  gpu_unlock(di);
  return NULL;
}
size_t nanorq_repair_all(nanorq *rq, struct ioctx *io) {
  const size_t Z = nanorq_blocks(rq);
  for (unsigned sbn = 0; sbn < Z; sbn++) { /* blocks an earlier nanorq_repair_block call decoded ahead (repair_ahead): commit / forget */
    struct blockst *b = rq->blocks[sbn];
    if (!b || !b->pre_state) continue;
    if (b->pre_state == 1) commit_rows(rq, (uint8_t)sbn, b, b->pre_lost, b->pre_n, io);
    b->pre_state = 0;
  }
  if (ndev()) {
    struct all_job j;
    memset(&j, 0, sizeof(j));
    j.rq = rq; j.io = io;
    for_devices(repair_all_worker, &j, devices_for(rq));
  }
  size_t complete = 0;
  for (unsigned sbn = 0; sbn < Z; sbn++) {
    struct blockst *b = rq->blocks[sbn];
    if (b && b->mask && mask_gaps(b, b->K) == 0) complete++;
  }
  return complete;
}

size_t nanorq_decoder_flush(nanorq *rq, struct ioctx *io) {
  if (!ndev() || !io) return 0;
  settle_all_uploads(rq);
  size_t nflushed = 0;
  for (unsigned sbn = 0; sbn < NRQ_Z_MAX; sbn++) {
    struct blockst *b = rq->blocks[sbn];
    if (!b || !b->dev || !b->dirty) continue;
    gpu_lock(b->di);
    if (flush_dev_block(rq, (uint8_t)sbn, b, io, mask_gaps(b, b->K) == 0)) { b->dirty = false; nflushed++; }
    gpu_unlock(b->di);
  }
  for (int d = 0; d < g_ndev; d++) {
    gpu_lock(d);
    nrq_stream_sync(g_dev[d].c, 2);
    gpu_unlock(d);
  }
  return nflushed;
}

/* Give back what the layer caches between objects: the page-locked host rows of freed objects (up to 8 GiB) and the
 * device pools of every context (up to 24 GiB each).  For long-lived processes after a large object. */
void nanorq_trim(void) {
  pthread_mutex_lock(&g_pin_lock);
  for (int i = 0; i < PIN_CACHE_SLOTS; i++)
    if (g_pin_cache[i].p) {
      nrq_host_free_pinned((uint8_t *)g_pin_cache[i].p - 64);
      g_pin_cache[i].p = NULL;
      g_pin_cache[i].cap = 0;
    }
  g_pin_cached = 0;
  pthread_mutex_unlock(&g_pin_lock);
  for (int d = 0; d < ndev(); d++) {
    gpu_lock(d);
    nrq_dev_trim(g_dev[d].c);
    gpu_unlock(d);
  }
}
