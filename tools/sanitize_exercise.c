/*
 * sanitize_exercise -- the host layer of the library driven from plain C, for the sanitizer builds of tools/sanitize.sh (a
 * Python process with a preloaded sanitizer runtime cannot bring up torch's copy of the HIP runtime).
 *   sanitize_exercise threads   two host threads, an object each, through the drop-in API at the same time (ThreadSanitizer)
 *   sanitize_exercise faults    the error paths: a packet batch, nanorq_repair_all and the per-block calls with the n-th
 *                               runtime call failing (nanorq_hip_option "fail_after"), n swept (AddressSanitizer / UBSan)
 *
 * The reference has no globals, so distinct nanorq objects are independent ("Threading"); this
 * library keeps process-wide state behind them -- the GPU contexts, their locks, the page-locked buffer cache -- and runs
 * a host thread per device inside the batched calls.  Each thread here encodes an object of several blocks with the
 * per-block calls AND the batched ones, drops 10 % of the source symbols, decodes through both families, and compares.
 * Exit status 0 = both objects came back intact (ThreadSanitizer makes it non-zero on a report: halt_on_error).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "io.h"
#include "nanorq.h"
#include "nanorq_batch.h"

struct job {
  unsigned seed;
  size_t K, T, Z;
  int ok;
};

static uint32_t rnd(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}

static void *work(void *arg) {
  struct job *j = arg;
  const size_t F = j->Z * j->K * j->T - 7, T = j->T;
  uint64_t st = j->seed;
  uint8_t *in = malloc(F), *out = calloc(1, F), *sym = malloc(T);
  j->ok = 0;
  if (!in || !out || !sym) return NULL;
  for (size_t i = 0; i < F; i++) in[i] = (uint8_t)rnd(&st);
  for (int round = 0; round < 3; round++) {
    struct ioctx *iin = ioctx_from_mem(in, F), *iout = ioctx_from_mem(out, F);
    nanorq *enc = nanorq_encoder_new_ex(F, (uint16_t)T, (uint16_t)j->K, 0, 8);
    if (!enc || !iin || !iout) return NULL;
    nanorq *dec = nanorq_decoder_new(nanorq_oti_common(enc), nanorq_oti_scheme_specific(enc));
    if (!dec) return NULL;
    memset(out, 0, F);
    const size_t Z = nanorq_blocks(enc);
    if (round & 1) {
      if (nanorq_generate_symbols_all(enc, iin) != Z) return NULL;
    }
    for (size_t sbn = 0; sbn < Z; sbn++) {
      const size_t k = nanorq_block_symbols(enc, (uint8_t)sbn);
      if (!(round & 1) && !nanorq_generate_symbols(enc, (uint8_t)sbn, iin)) return NULL;
      uint32_t dropped = 0;
      for (uint32_t esi = 0; esi < k; esi++) {
        if (rnd(&st) % 10u == 0u) { dropped++; continue; }
        if (nanorq_encode(enc, sym, esi, (uint8_t)sbn, iin) != T) return NULL;
        if (nanorq_decoder_add_symbol(dec, sym, nanorq_tag((uint8_t)sbn, esi), iout) == NANORQ_SYM_ERR) return NULL;
      }
      for (uint32_t esi = (uint32_t)k; esi < k + dropped + 3; esi++) {
        if (nanorq_encode(enc, sym, esi, (uint8_t)sbn, iin) != T) return NULL;
        if (nanorq_decoder_add_symbol(dec, sym, nanorq_tag((uint8_t)sbn, esi), iout) == NANORQ_SYM_ERR) return NULL;
      }
    }
    if (round == 2) {
      if (nanorq_repair_all(dec, iout) != Z) return NULL;
    } else {
      for (size_t sbn = 0; sbn < Z; sbn++)
        if (!nanorq_repair_block(dec, iout, (uint8_t)sbn)) return NULL;
    }
    if (memcmp(in, out, F) != 0) return NULL;
    nanorq_free(enc);
    nanorq_free(dec);
    iin->destroy(iin);
    iout->destroy(iout);
  }
  free(in); free(out); free(sym);
  j->ok = 1;
  return NULL;
}

/* ---- error paths: every entry point that moves bytes, with the n-th checked runtime call failing ---- */
static int faults(void) {
  const size_t K = 150, T = 256, Z = 4, F = Z * K * T - 11;
  uint64_t st = 77;
  uint8_t *in = malloc(F), *sym = malloc(T), *sym2 = malloc(T);
  if (!in || !sym || !sym2) return 1;
  for (size_t i = 0; i < F; i++) in[i] = (uint8_t)rnd(&st);
  struct ioctx *iin = ioctx_from_mem(in, F);
  nanorq *enc = nanorq_encoder_new_ex(F, (uint16_t)T, (uint16_t)K, 0, 8);
  if (!enc || !iin || nanorq_generate_symbols_all(enc, iin) != Z) return 1;
  /* the packets of a lossy transmission, in a page-locked buffer */
  const size_t cap = Z * (K + 40);
  uint8_t *pk = nanorq_pinned_alloc(cap * T);
  uint32_t *tags = malloc(cap * sizeof(uint32_t));
  int *res = malloc(cap * sizeof(int));
  if (!pk || !tags || !res) return 1;
  uint32_t n = 0;
  for (size_t sbn = 0; sbn < Z; sbn++) {
    const size_t k = nanorq_block_symbols(enc, (uint8_t)sbn);
    uint32_t dropped = 0;
    for (uint32_t esi = 0; esi < k; esi++) {
      if (rnd(&st) % 10u == 0u) { dropped++; continue; }
      if (nanorq_encode(enc, pk + (size_t)n * T, esi, (uint8_t)sbn, iin) != T) return 1;
      tags[n++] = nanorq_tag((uint8_t)sbn, esi);
    }
    for (uint32_t esi = (uint32_t)k; esi < k + dropped + 2; esi++) {
      if (nanorq_encode(enc, pk + (size_t)n * T, esi, (uint8_t)sbn, iin) != T) return 1;
      tags[n++] = nanorq_tag((uint8_t)sbn, esi);
    }
  }
  const uint64_t oc = nanorq_oti_common(enc);
  const uint32_t os = nanorq_oti_scheme_specific(enc);
  int injected = 0;
  for (long long fa = 1; fa <= 70; fa += (fa < 40 ? 1 : 6)) {
    struct ioctx *iout = ioctx_from_pinned_mem(F);
    nanorq *dec = nanorq_decoder_new(oc, os);
    if (!dec || !iout) return 1;
    memset(ioctx_mem_base(iout), 0, F);
    const int before = nanorq_hip_option(0, "faults_injected", 0);
    nanorq_hip_option(0, "fail_after", fa);
    size_t added = nanorq_decoder_add_symbols(dec, pk, tags, n, res, iout);
    nanorq_hip_option(0, "fail_after", 0);
    if (added != n) { /* what was reported NANORQ_SYM_ERR is sent again, one call per symbol */
      for (uint32_t k = 0; k < n; k++)
        if (res[k] == NANORQ_SYM_ERR && nanorq_decoder_add_symbol(dec, pk + (size_t)k * T, tags[k], iout) != NANORQ_SYM_ADDED) return 2;
    }
    nanorq_hip_option(0, "fail_after", fa);
    size_t done = nanorq_repair_all(dec, iout);
    nanorq_hip_option(0, "fail_after", 0);
    size_t done2 = done;
    if (done != Z && (done2 = nanorq_repair_all(dec, iout)) != Z) { fprintf(stderr, "fail_after %lld: repair_all %zu then %zu of %zu\n", fa, done, done2, Z); return 3; }
    if (memcmp(in, ioctx_mem_base(iout), F) != 0) {
      const uint8_t *o = ioctx_mem_base(iout);
      size_t bad = 0, first = F;
      for (size_t i = 0; i < F; i++) if (in[i] != o[i]) { if (first == F) first = i; bad++; }
      fprintf(stderr, "fail_after %lld: added %zu of %u, repair_all %zu then %zu; %zu bytes differ, first at %zu (block %zu, row %zu)\n", fa, added, n, done, done2,
              bad, first, first / (K * T), first % (K * T) / T);
      return 4;
    }
    injected += nanorq_hip_option(0, "faults_injected", 0) - before;
    nanorq_free(dec);
    iout->destroy(iout);
    /* per-block calls on host-resident blocks */
    uint8_t *out = calloc(1, F);
    struct ioctx *io2 = ioctx_from_mem(out, F);
    dec = nanorq_decoder_new(oc, os);
    if (!dec || !io2 || !out) return 1;
    for (uint32_t k = 0; k < n; k++)
      if (nanorq_decoder_add_symbol(dec, pk + (size_t)k * T, tags[k], io2) == NANORQ_SYM_ERR) return 5;
    nanorq_hip_option(0, "fail_after", fa);
    (void)nanorq_repair_block(dec, io2, 1);
    nanorq_hip_option(0, "fail_after", 0);
    for (size_t sbn = 0; sbn < Z; sbn++)
      if (!nanorq_repair_block(dec, io2, (uint8_t)sbn)) return 6;
    if (memcmp(in, out, F) != 0) return 7;
    nanorq_free(dec);
    io2->destroy(io2);
    free(out);
    /* encoder: a failed solve, then the same call again */
    nanorq *e2 = nanorq_encoder_new_ex(F, (uint16_t)T, (uint16_t)K, 0, 8);
    if (!e2) return 1;
    nanorq_hip_option(0, "fail_after", fa);
    (void)nanorq_generate_symbols(e2, 2, iin);
    nanorq_hip_option(0, "fail_after", 0);
    if (nanorq_encode(e2, sym, (uint32_t)K + 3, 2, iin) != T) return 8;
    if (nanorq_encode(enc, sym2, (uint32_t)K + 3, 2, iin) != T) return 8;
    if (memcmp(sym, sym2, T) != 0) return 9;
    nanorq_free(e2);
  }
  nanorq_free(enc);
  iin->destroy(iin);
  nanorq_pinned_free(pk);
  free(tags); free(res); free(in); free(sym); free(sym2);
  nanorq_trim();
  printf("sanitize_exercise faults: %d failures injected, every retry gave the object back\n", injected);
  return injected >= 30 ? 0 : 10;
}

int main(int argc, char **argv) {
  if (argc > 1 && strcmp(argv[1], "faults") == 0) return faults();
  struct job jobs[2] = {{11u, 200, 256, 5, 0}, {23u, 333, 64, 4, 0}};
  pthread_t th[2];
  for (int i = 0; i < 2; i++)
    if (pthread_create(&th[i], NULL, work, &jobs[i]) != 0) return 3;
  for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
  nanorq_trim();
  printf("sanitize_exercise threads: thread 0 %s, thread 1 %s\n", jobs[0].ok ? "ok" : "FAILED", jobs[1].ok ? "ok" : "FAILED");
  return jobs[0].ok && jobs[1].ok ? 0 : 1;
}
