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
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/nanorq.h"
#include "../../include/nanorq_hip.h"

#define NRQ_Z_MAX 256u
#define NRQ_K_MAX 56403u
#define ENC_WINDOW 512u /* repair symbols fetched from the device per miss of the encode cache */

struct part { /* RFC 6330 section 4.4.1.2 Partition[I, J] */
  size_t IL, IS, JL, JS;
};

struct blockst {
  uint32_t K;
  bool loaded, inverted;
  uint8_t *src;       /* K x T: source symbols (encoder) / received source symbols (decoder) */
  void *d_src;        /* device copy */
  void *d_inter;      /* device: L x T intermediate symbols once solved */
  /* decoder side */
  uint32_t *mask;     /* received-ESI bitmap */
  size_t mask_words;
  uint32_t *rep_esi;  /* repair symbols in arrival order */
  uint8_t *rep_data;
  size_t nrep, rep_cap;
  size_t spare;       /* max_esi - K: rows available beyond L (reference D sizing, nanorq.c:137-142) */
  /* encoder side: window of generated symbols */
  uint8_t *win;
  uint32_t win_isi0, win_n;
};

struct nanorq {
  size_t F, T, Al;         /* common OTI */
  size_t Z, N, Kt;         /* scheme specific */
  struct part src_part, sub_part;
  uint32_t Kp, S, H, L;    /* parameters of block 0, shared by every block (nanorq.c:289, :372) */
  uint32_t max_esi;
  bool precalc;
  struct blockst *blocks[NRQ_Z_MAX];
};

/* ---------------------------------------------------------------- GPU context (process-wide) ---- */
static nrq_ctx *g_ctx;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void ctx_init(void) {
  int dev = 0;
  const char *e = getenv("NANORQ_HIP_DEVICE");
  if (e && *e) dev = atoi(e);
  if (nrq_ctx_create(dev, NULL, &g_ctx) != 0) g_ctx = NULL;
}
static nrq_ctx *ctx(void) {
  pthread_once(&g_once, ctx_init);
  return g_ctx;
}

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
  b->mask[w] |= 1u << (id % 32);
}
static size_t mask_gaps(const struct blockst *b, size_t until) { /* zero bits below `until` */
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

uint64_t nanorq_oti_common(nanorq *rq) { /* nanorq.c:309-315: F<<24 | (T-1) */
  return ((uint64_t)rq->F << 24) | ((rq->T - 1) & 0xffff);
}
uint32_t nanorq_oti_scheme_specific(nanorq *rq) { /* nanorq.c:317-324: (Z-1)<<24 | (N-1)<<8 | Al */
  return (uint32_t)((rq->Z - 1) << 24) | (uint32_t)((rq->N - 1) << 8) | (uint32_t)rq->Al;
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

/* ---------------------------------------------------------------------- construction ---- */
nanorq *nanorq_encoder_new_ex(size_t len, uint16_t T16, uint16_t K, uint16_t Z16, uint8_t Al8) { /* nanorq.c:241-296 */
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
  nanorq *rq = calloc(1, sizeof(nanorq));
  if (!rq) return NULL;
  rq->F = len; rq->T = T; rq->Al = Al; rq->Z = Z; rq->N = 1; rq->Kt = Kt;
  rq->src_part = partition(Kt, Z);
  rq->sub_part = partition(T / Al, rq->N);
  if (!set_block_params(rq)) { free(rq); return NULL; }
  return rq;
}

nanorq *nanorq_encoder_new(size_t len, uint16_t T, uint8_t Al) { return nanorq_encoder_new_ex(len, T, 0, 0, Al); }

nanorq *nanorq_decoder_new(uint64_t common, uint32_t specific) { /* nanorq.c:336-377 */
  uint64_t F = common >> 24;
  size_t T = (size_t)((common & 0xffff) + 1) & 0xffff; /* 16-bit wrap as in the reference */
  if (F > NANORQ_MAX_TRANSFER) return NULL;
  size_t Z = ((specific >> 24) & 0xff) + 1, N = ((specific >> 8) & 0xffff) + 1, Al = specific & 0xff;
  if (T == 0 || Al == 0 || T < Al || T % Al != 0) return NULL;
  size_t Kt = ceil_div(F, T);
  if (ceil_div(Kt, Z) > NRQ_K_MAX) return NULL;
  nanorq *rq = calloc(1, sizeof(nanorq));
  if (!rq) return NULL;
  rq->F = F; rq->T = T; rq->Al = Al; rq->Z = Z; rq->N = N; rq->Kt = Kt;
  rq->src_part = partition(Kt, Z);
  rq->sub_part = partition(T / Al, N);
  if (!set_block_params(rq)) { free(rq); return NULL; }
  rq->max_esi = 2 * rq->Kp;
  return rq;
}

bool nanorq_set_max_esi(nanorq *rq, uint32_t max_esi) { /* nanorq.c:471-476 */
  if (!rq || max_esi >= (1u << 24) || max_esi < rq->Kp) return false;
  rq->max_esi = max_esi;
  return true;
}

/* -------------------------------------------------------------------- per-block state ---- */
static struct blockst *get_block(nanorq *rq, uint8_t sbn) { /* nanorq.c:130-146 */
  if (rq->blocks[sbn]) return rq->blocks[sbn];
  struct blockst *b = calloc(1, sizeof(*b));
  if (!b) return NULL;
  b->K = (uint32_t)nanorq_block_symbols(rq, sbn);
  if (rq->max_esi) {
    b->mask_words = rq->max_esi / 32 + 1;
    b->mask = calloc(b->mask_words, sizeof(uint32_t));
    b->spare = rq->max_esi - b->K;
  }
  b->src = calloc((size_t)(b->K ? b->K : 1) * rq->T, 1);
  if (!b->src || (rq->max_esi && !b->mask)) { free(b->src); free(b->mask); free(b); return NULL; }
  rq->blocks[sbn] = b;
  return b;
}

static void drop_device(struct blockst *b) {
  nrq_ctx *c = g_ctx;
  if (c) {
    if (b->d_src) nrq_dev_free(c, b->d_src);
    if (b->d_inter) nrq_dev_free(c, b->d_inter);
  }
  b->d_src = b->d_inter = NULL;
}

void nanorq_encoder_cleanup(nanorq *rq, uint8_t sbn) { /* nanorq.c:437-451 */
  struct blockst *b = rq->blocks[sbn];
  if (!b) return;
  drop_device(b);
  free(b->src); free(b->mask); free(b->rep_esi); free(b->rep_data); free(b->win); free(b);
  rq->blocks[sbn] = NULL;
}

void nanorq_encoder_reset(nanorq *rq, uint8_t sbn) { /* nanorq.c:453-469 */
  struct blockst *b = rq->blocks[sbn];
  if (!b) return;
  b->loaded = b->inverted = false;
  memset(b->src, 0, (size_t)(b->K ? b->K : 1) * rq->T);
  b->nrep = 0;
  b->win_n = 0;
  if (b->mask) memset(b->mask, 0, b->mask_words * sizeof(uint32_t));
}

void nanorq_free(nanorq *rq) { /* nanorq.c:298-307 */
  if (!rq) return;
  for (unsigned sbn = 0; sbn < NRQ_Z_MAX; sbn++) nanorq_encoder_cleanup(rq, (uint8_t)sbn);
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
    if (offset >= rq->F) continue;
    if (io->seek(io, offset)) {
      if (offset + stride >= rq->F) stride = rq->F - offset;
      moved += out ? io->write(io, ptr + col, stride) : io->read(io, ptr + col, stride);
      col += stride;
    }
  }
  return moved;
}

static bool load_block(nanorq *rq, uint8_t sbn, struct blockst *b, struct ioctx *io) { /* nanorq.c:175-182 */
  if (!io) return false;
  memset(b->src, 0, (size_t)(b->K ? b->K : 1) * rq->T);
  if (rq->N == 1) {
    /* no sub-blocking (the only case nanorq creates, nanorq.c:78): the block's symbols are one contiguous stretch
     * of the object -- one seek and one read instead of K of each; bytes beyond F stay zero */
    const size_t off = symbol_offset(rq, sbn, 0, b->K, 0) * rq->Al, want = (size_t)b->K * rq->T;
    if (off < rq->F && io->seek(io, off)) {
      const size_t len = off + want > rq->F ? rq->F - off : want;
      size_t got = 0;
      while (got < len) {
        const size_t n = io->read(io, b->src + got, len - got);
        if (n == 0) break;
        got += n;
      }
    }
    return true;
  }
  for (uint32_t esi = 0; esi < b->K; esi++) transfer_symbol(rq, sbn, esi, b->K, b->src + (size_t)esi * rq->T, io, 0);
  return true;
}

/* ------------------------------------------------------------------------- encoding ---- */
bool nanorq_precalculate(nanorq *rq) { /* nanorq.c:393-401 */
  nrq_ctx *c = ctx();
  size_t k0 = nanorq_block_symbols(rq, 0);
  if (!c || k0 == 0) return false;
  if (nrq_precalculate(c, (uint32_t)k0, rq->Kp) != 0) return false;
  rq->precalc = true;
  return true;
}

bool nanorq_generate_symbols(nanorq *rq, uint8_t sbn, struct ioctx *io) { /* nanorq.c:206-232 */
  struct blockst *b = get_block(rq, sbn);
  if (!b) return false;
  if (b->inverted) return true;
  if (!b->loaded) b->loaded = load_block(rq, sbn, b, io);
  if (!b->loaded || b->K == 0) return false;
  nrq_ctx *c = ctx();
  if (!c) return false; /* no GPU: no solve */
  const size_t T = rq->T, bytes = (size_t)b->K * T;
  /* every block of an object is coded with block 0's K' (nanorq.c:289); a short block just has more padding */
  if (!b->d_src && nrq_dev_alloc(c, bytes, &b->d_src) != 0) return false;
  if (!b->d_inter && nrq_dev_alloc(c, (size_t)rq->L * T, &b->d_inter) != 0) return false;
  if (nrq_dev_upload(c, b->d_src, b->src, bytes) != 0) return false;
  if (nrq_encode_blocks(c, b->K, rq->Kp, (uint32_t)T, 1, b->d_src, bytes, b->d_inter, (size_t)rq->L * T, 0, NULL, NULL, 0) != 0)
    return false;
  if (nrq_ctx_sync(c) != 0) return false;
  b->win_n = 0;
  b->inverted = true;
  return true;
}

/* repair symbol with internal id `isi` through a window of device-generated symbols */
static bool fetch_symbol(nanorq *rq, struct blockst *b, uint32_t isi, uint8_t *out) {
  const size_t T = rq->T;
  if (!(b->win_n && isi >= b->win_isi0 && isi < b->win_isi0 + b->win_n)) {
    nrq_ctx *c = ctx();
    if (!c || !b->d_inter) return false;
    if (!b->win && !(b->win = malloc((size_t)ENC_WINDOW * T))) return false;
    uint32_t isis[ENC_WINDOW], n = ENC_WINDOW;
    for (uint32_t k = 0; k < n; k++) isis[k] = isi + k;
    void *d_out = NULL;
    if (nrq_dev_alloc(c, (size_t)n * T, &d_out) != 0) return false;
    bool ok = nrq_gen_symbols(c, b->K, rq->Kp, (uint32_t)T, 1, b->d_inter, (size_t)rq->L * T, n, isis, d_out, (size_t)n * T) == 0 &&
              nrq_dev_download(c, b->win, d_out, (size_t)n * T) == 0;
    nrq_dev_free(c, d_out);
    if (!ok) return false;
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
    /* before the solve: the source symbol itself; after it the reference regenerates the same bytes from
     * the intermediate symbols (RFC 6330 is systematic) -- either way the loaded source row */
    if (!b->inverted && !b->loaded) b->loaded = load_block(rq, sbn, b, io);
    if (!b->inverted && !b->loaded) return 0;
    memcpy(data, b->src + (size_t)esi * T, T);
    return T;
  }
  if (esi > (1u << 24) - 1) return 0;
  if (!b->inverted) b->inverted = nanorq_generate_symbols(rq, sbn, io);
  if (!b->inverted) return 0;
  return fetch_symbol(rq, b, esi + (rq->Kp - b->K), data) ? T : 0;
}

/* ------------------------------------------------------------------------- decoding ---- */
int nanorq_decoder_add_symbol(nanorq *rq, void *data, uint32_t tag, struct ioctx *io) { /* nanorq.c:478-509 */
  uint8_t sbn = (uint8_t)(tag >> 24);
  uint32_t esi = tag & 0x00ffffffu;
  struct blockst *b = get_block(rq, sbn);
  if (!b || esi > rq->max_esi) return NANORQ_SYM_ERR;
  if (mask_gaps(b, b->K) == 0) return NANORQ_SYM_IGN;
  if (mask_get(b, esi)) return NANORQ_SYM_DUP;
  const size_t T = rq->T;
  if (esi < b->K) {
    memcpy(b->src + (size_t)esi * T, data, T);
    if (io) transfer_symbol(rq, sbn, esi, b->K, data, io, 1);
  } else {
    if (b->nrep == b->rep_cap) {
      size_t nc = b->rep_cap ? b->rep_cap * 2 : 64;
      uint32_t *e = realloc(b->rep_esi, nc * sizeof(uint32_t));
      if (!e) return NANORQ_SYM_ERR;
      b->rep_esi = e;
      uint8_t *d = realloc(b->rep_data, nc * T);
      if (!d) return NANORQ_SYM_ERR;
      b->rep_data = d;
      b->rep_cap = nc;
    }
    b->rep_esi[b->nrep] = esi;
    memcpy(b->rep_data + b->nrep * T, data, T);
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

bool nanorq_repair_block(nanorq *rq, struct ioctx *io, uint8_t sbn) { /* nanorq.c:591-631 */
  struct blockst *b = get_block(rq, sbn);
  if (!b) return false;
  const size_t gaps = mask_gaps(b, b->K);
  if (gaps == 0) return true;
  if (b->nrep < gaps) return false;
  const size_t overhead = b->nrep - gaps;
  if (overhead > b->spare) return false; /* D.rows < L + overhead in the reference */
  nrq_ctx *c = ctx();
  if (!c) return false;
  const size_t T = rq->T, bytes = (size_t)b->K * T;
  uint32_t *lost = malloc(gaps * sizeof(uint32_t));
  if (!lost) return false;
  size_t g = 0;
  for (uint32_t e = 0; e < b->K; e++)
    if (!mask_get(b, e)) lost[g++] = e;
  bool ok = false;
  void *d_rep = NULL;
  uint32_t nlost = (uint32_t)gaps, nrep = (uint32_t)b->nrep;
  int status = 0;
  if (!b->d_src && nrq_dev_alloc(c, bytes, &b->d_src) != 0) goto out;
  if (nrq_dev_alloc(c, b->nrep * T, &d_rep) != 0) goto out;
  if (nrq_dev_upload(c, b->d_src, b->src, bytes) != 0) goto out;
  if (nrq_dev_upload(c, d_rep, b->rep_data, b->nrep * T) != 0) goto out;
  if (nrq_decode_blocks(c, b->K, rq->Kp, (uint32_t)T, 1, b->d_src, bytes, lost, &nlost, nlost, b->rep_esi, &nrep, nrep, d_rep,
                        b->nrep * T, NULL, 0, &status) != 0)
    goto out;
  if (!status) goto out; /* rank deficient: retry after more symbols (nanorq.c:620-623) */
  if (nrq_dev_download(c, b->src, b->d_src, bytes) != 0) goto out;
  for (size_t k = 0; k < gaps; k++) { /* write_repair_rows, nanorq.c:579-589 */
    if (io) transfer_symbol(rq, sbn, lost[k], b->K, b->src + (size_t)lost[k] * T, io, 1);
    mask_set(b, lost[k]);
  }
  ok = mask_gaps(b, b->K) == 0;
out:
  if (d_rep) nrq_dev_free(c, d_rep);
  free(lost);
  return ok;
}

/* =============================================================== batched variants (nanorq_batch.h) ==== */
/* blocks come in at most two sizes (RFC 6330 section 4.4.1.2 partition: JL blocks of IL symbols, JS of IS) */
static uint32_t class_of(nanorq *rq, unsigned sbn) { return sbn < rq->src_part.JL ? 0u : 1u; }

size_t nanorq_generate_symbols_all(nanorq *rq, struct ioctx *io) {
  nrq_ctx *c = ctx();
  const size_t Z = nanorq_blocks(rq), T = rq->T;
  if (!c || !io) return 0;
  for (uint32_t cls = 0; cls < 2; cls++) {
    /* the blocks of this size that still need the solve */
    unsigned todo[NRQ_Z_MAX], n = 0;
    uint32_t K = 0;
    for (unsigned sbn = 0; sbn < Z; sbn++) {
      if (class_of(rq, sbn) != cls) continue;
      struct blockst *b = get_block(rq, (uint8_t)sbn);
      if (!b || b->K == 0 || b->inverted) continue;
      if (!b->loaded) b->loaded = load_block(rq, (uint8_t)sbn, b, io);
      if (!b->loaded) continue;
      K = b->K;
      todo[n++] = sbn;
    }
    if (!n) continue;
    const size_t sbytes = (size_t)K * T, ibytes = (size_t)rq->L * T;
    void *d_src = NULL, *d_inter = NULL;
    if (nrq_dev_alloc(c, sbytes * n, &d_src) != 0) continue;
    if (nrq_dev_alloc(c, ibytes * n, &d_inter) != 0) { nrq_dev_free(c, d_src); continue; }
    bool ok = true;
    for (unsigned k = 0; k < n && ok; k++)
      ok = nrq_dev_upload_async(c, (uint8_t *)d_src + sbytes * k, rq->blocks[todo[k]]->src, sbytes) == 0;
    ok = ok && nrq_encode_blocks(c, K, rq->Kp, (uint32_t)T, n, d_src, sbytes, d_inter, ibytes, 0, NULL, NULL, 0) == 0 &&
         nrq_ctx_sync(c) == 0;
    /* every block keeps its own intermediate symbols on the device, like after nanorq_generate_symbols */
    for (unsigned k = 0; k < n && ok; k++) {
      struct blockst *b = rq->blocks[todo[k]];
      if (!b->d_inter && nrq_dev_alloc(c, ibytes, &b->d_inter) != 0) { ok = false; break; }
      if (nrq_dev_copy(c, b->d_inter, (uint8_t *)d_inter + ibytes * k, ibytes) != 0) { ok = false; break; }
      b->win_n = 0;
      b->inverted = true;
    }
    if (ok) ok = nrq_ctx_sync(c) == 0;
    nrq_dev_free(c, d_src);
    nrq_dev_free(c, d_inter);
  }
  size_t done = 0;
  for (unsigned sbn = 0; sbn < Z; sbn++)
    if (rq->blocks[sbn] && rq->blocks[sbn]->inverted) done++;
  return done;
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
  nrq_ctx *c = ctx();
  if (!b->inverted || !c) return 0;
  /* repair symbols: generated on the device in one go, one download */
  uint32_t *isis = malloc((size_t)left * sizeof(uint32_t));
  void *d_out = NULL;
  bool ok = isis && nrq_dev_alloc(c, (size_t)left * T, &d_out) == 0;
  if (ok) {
    for (uint32_t k = 0; k < left; k++) isis[k] = esi + k + (rq->Kp - b->K);
    ok = nrq_gen_symbols(c, b->K, rq->Kp, (uint32_t)T, 1, b->d_inter, (size_t)rq->L * T, left, isis, d_out, (size_t)left * T) == 0 &&
         nrq_dev_download(c, out, d_out, (size_t)left * T) == 0;
  }
  if (d_out) nrq_dev_free(c, d_out);
  free(isis);
  return ok ? (size_t)n * T : 0;
}

size_t nanorq_decoder_add_symbols(nanorq *rq, const void *data, const uint32_t *tags, uint32_t n, int *results, struct ioctx *io) {
  size_t added = 0;
  const uint8_t *p = data;
  for (uint32_t k = 0; k < n; k++) {
    /* (add_symbol copies; the const is cast away only because the per-symbol signature of the reference is void *) */
    const int r = nanorq_decoder_add_symbol(rq, (void *)(uintptr_t)(p + (size_t)k * rq->T), tags[k], io);
    if (results) results[k] = r;
    if (r == NANORQ_SYM_ADDED) added++;
  }
  return added;
}

size_t nanorq_repair_all(nanorq *rq, struct ioctx *io) {
  nrq_ctx *c = ctx();
  const size_t Z = nanorq_blocks(rq), T = rq->T;
  if (c) {
    for (uint32_t cls = 0; cls < 2; cls++) {
      unsigned todo[NRQ_Z_MAX], n = 0;
      uint32_t K = 0;
      size_t lost_cap = 0, rep_cap = 0;
      for (unsigned sbn = 0; sbn < Z; sbn++) {
        if (class_of(rq, sbn) != cls) continue;
        struct blockst *b = rq->blocks[sbn];
        if (!b || b->K == 0) continue;
        const size_t gaps = mask_gaps(b, b->K);
        if (gaps == 0 || b->nrep < gaps || b->nrep - gaps > b->spare) continue; /* as nanorq_repair_block */
        K = b->K;
        if (gaps > lost_cap) lost_cap = gaps;
        if (b->nrep > rep_cap) rep_cap = b->nrep;
        todo[n++] = sbn;
      }
      if (!n) continue;
      const size_t sbytes = (size_t)K * T, rbytes = rep_cap * T;
      uint32_t *lost = calloc((size_t)n * lost_cap, sizeof(uint32_t)), *nlost = calloc(n, sizeof(uint32_t));
      uint32_t *resi = calloc((size_t)n * rep_cap, sizeof(uint32_t)), *nrep = calloc(n, sizeof(uint32_t));
      int *status = calloc(n, sizeof(int));
      void *d_src = NULL, *d_rep = NULL;
      bool ok = lost && nlost && resi && nrep && status && nrq_dev_alloc(c, sbytes * n, &d_src) == 0 &&
                nrq_dev_alloc(c, rbytes * n, &d_rep) == 0;
      for (unsigned k = 0; k < n && ok; k++) {
        struct blockst *b = rq->blocks[todo[k]];
        uint32_t g = 0;
        for (uint32_t e = 0; e < b->K; e++)
          if (!mask_get(b, e)) lost[(size_t)k * lost_cap + g++] = e;
        nlost[k] = g;
        nrep[k] = (uint32_t)b->nrep;
        memcpy(resi + (size_t)k * rep_cap, b->rep_esi, b->nrep * sizeof(uint32_t));
        ok = nrq_dev_upload_async(c, (uint8_t *)d_src + sbytes * k, b->src, sbytes) == 0 &&
             nrq_dev_upload_async(c, (uint8_t *)d_rep + rbytes * k, b->rep_data, b->nrep * T) == 0;
      }
      ok = ok && nrq_decode_blocks(c, K, rq->Kp, (uint32_t)T, n, d_src, sbytes, lost, nlost, (uint32_t)lost_cap, resi, nrep,
                                   (uint32_t)rep_cap, d_rep, rbytes, NULL, 0, status) == 0;
      for (unsigned k = 0; k < n && ok; k++) /* the decoded blocks come back with one wait */
        if (status[k]) ok = nrq_dev_download_async(c, rq->blocks[todo[k]]->src, (uint8_t *)d_src + sbytes * k, sbytes) == 0;
      ok = ok && nrq_ctx_sync(c) == 0;
      for (unsigned k = 0; k < n && ok; k++) {
        if (!status[k]) continue; /* rank deficient: retry after more symbols (nanorq.c:620-623) */
        struct blockst *b = rq->blocks[todo[k]];
        for (uint32_t j = 0; j < nlost[k]; j++) {
          const uint32_t e = lost[(size_t)k * lost_cap + j];
          if (io) transfer_symbol(rq, (uint8_t)todo[k], e, b->K, b->src + (size_t)e * T, io, 1);
          mask_set(b, e);
        }
      }
      if (d_src) nrq_dev_free(c, d_src);
      if (d_rep) nrq_dev_free(c, d_rep);
      free(lost); free(nlost); free(resi); free(nrep); free(status);
    }
  }
  size_t complete = 0;
  for (unsigned sbn = 0; sbn < Z; sbn++) {
    struct blockst *b = rq->blocks[sbn];
    if (b && b->mask && mask_gaps(b, b->K) == 0) complete++;
  }
  return complete;
}
