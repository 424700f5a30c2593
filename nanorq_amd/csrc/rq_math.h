/*
 * rq_math.h -- RFC 6330 integer machinery shared by host (C/C++) and gfx950 device code:
 * parameter derivation (section 5.3.1.2 / Table 2), Rand (5.3.5.1), Deg (5.3.5.2),
 * Tuple (5.3.5.4) and the LT column expansion used for both constraint rows and symbol generation.
 *
 * Replaces, on this path, the reference's params.c:21-65, tuple.c:13-43, rand.c:183-190
 * (behaviour identical; verified against the oracle in tests/test_planner_emu.py and, on the GPU,
 * by every parity test of tests/test_gpu_parity.py).
 */
#ifndef NRQ_RQ_MATH_H
#define NRQ_RQ_MATH_H

#include <stdint.h>

#include "rfc6330_tables.h"

#if defined(__HIPCC__)
#define RQ_HD __host__ __device__ inline
#else
#define RQ_HD static inline
#endif

typedef struct rq_params {
  uint32_t K, Kp, J, S, H, W, L, P, P1, U, B;
} rq_params;

typedef struct rq_tuple {
  uint32_t d, a, b, d1, a1, b1;
} rq_tuple;

#define RQ_MAX_LT_COLS 40 /* capacity of column-list arrays */
#define RQ_LT_COLS_MAX_REAL 33 /* longest list RFC 6330 5.3.5.2-4 can give: d <= 30 and d1 <= 3 (d1 = 3 only with d < 4, so really 32) */

/* ---- tables: host copy always; device copy (constant memory) only in HIP translation units ---- */
static const uint32_t rq_host_V[4 * 256] = {RQ_V_WORDS};
static const uint32_t rq_host_degF[31] = {
    0,       5243,    529531,  704294,  791675,  844104,  879057,  904023,  922747,  937311, 948962,
    958494,  966438,  973160,  978921,  983914,  988283,  992138,  995565,  998631,  1001391,
    1003887, 1006157, 1008229, 1010129, 1011876, 1013490, 1014983, 1016370, 1017662, 1048576};
#if defined(__HIPCC__)
static __device__ __constant__ const uint32_t rq_dev_V[4 * 256] = {RQ_V_WORDS};
static __device__ __constant__ const uint32_t rq_dev_degF[31] = {
    0,       5243,    529531,  704294,  791675,  844104,  879057,  904023,  922747,  937311, 948962,
    958494,  966438,  973160,  978921,  983914,  988283,  992138,  995565,  998631,  1001391,
    1003887, 1006157, 1008229, 1010129, 1011876, 1013490, 1014983, 1016370, 1017662, 1048576};
#endif

RQ_HD uint32_t rq_V(uint32_t t, uint32_t idx) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rq_dev_V[t * 256 + idx];
#else
  return rq_host_V[t * 256 + idx];
#endif
}
RQ_HD uint32_t rq_degF(uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rq_dev_degF[k];
#else
  return rq_host_degF[k];
#endif
}

RQ_HD uint32_t rq_rand(uint32_t y, uint32_t i, uint32_t m) {
  uint32_t v = rq_V(0, (y + i) & 255u) ^ rq_V(1, ((y >> 8) + i) & 255u) ^ rq_V(2, ((y >> 16) + i) & 255u) ^
               rq_V(3, ((y >> 24) + i) & 255u);
  return v % m;
}

RQ_HD rq_tuple rq_make_tuple(const rq_params *p, uint32_t X) {
  rq_tuple t;
  uint64_t A = 53591ull + (uint64_t)p->J * 997ull;
  if ((A & 1ull) == 0) A++;
  uint64_t B1 = 10267ull * ((uint64_t)p->J + 1ull);
  uint32_t y = (uint32_t)(B1 + (uint64_t)X * A);
  uint32_t v = rq_rand(y, 0, 1u << 20);
  uint32_t dg = 0;
  for (uint32_t k = 0; k < 31; k++)
    if (v < rq_degF(k)) { dg = k; break; }
  if (dg > p->W - 2) dg = p->W - 2;
  t.d = dg;
  t.a = 1 + rq_rand(y, 1, p->W - 1);
  t.b = rq_rand(y, 2, p->W);
  t.d1 = (t.d < 4) ? 2 + rq_rand(X, 3, 2) : 2;
  t.a1 = 1 + rq_rand(X, 4, p->P1 - 1);
  t.b1 = rq_rand(X, 5, p->P1);
  return t;
}

/* columns (intermediate-symbol indices) of the LT row / encoding symbol with ISI X; returns count */
RQ_HD uint32_t rq_lt_columns(const rq_params *p, uint32_t X, uint32_t *out) {
  rq_tuple t = rq_make_tuple(p, X);
  uint32_t n = 0, b = t.b, b1 = t.b1;
  out[n++] = b;
  for (uint32_t k = 1; k < t.d; k++) {
    b += t.a;
    if (b >= p->W) b -= p->W; /* a < W, b < W */
    out[n++] = b;
  }
  while (b1 >= p->P) b1 = (b1 + t.a1) % p->P1;
  out[n++] = p->W + b1;
  for (uint32_t k = 1; k < t.d1; k++) {
    b1 = (b1 + t.a1) % p->P1;
    while (b1 >= p->P) b1 = (b1 + t.a1) % p->P1;
    out[n++] = p->W + b1;
  }
  return n;
}

/* ---- host-only: Table 2 lookup ---- */
static const struct { uint16_t kp, j, s, h, w; } rq_table2[RQ_TABLE2_COUNT] = {RQ_TABLE2_ROWS};

static inline int rq_is_prime(uint32_t n) {
  if (n < 2) return 0;
  for (uint32_t f = 2; f * f <= n; f++)
    if (n % f == 0) return 0;
  return 1;
}

/* returns 1 on success; K in [1, 56403] */
static inline int rq_params_init(uint32_t K, rq_params *p) {
  if (K == 0 || K > RQ_K_MAX) return 0;
  int lo = 0, hi = RQ_TABLE2_COUNT - 1;
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (rq_table2[mid].kp >= K) hi = mid; else lo = mid + 1;
  }
  p->K = K;
  p->Kp = rq_table2[lo].kp; p->J = rq_table2[lo].j; p->S = rq_table2[lo].s;
  p->H = rq_table2[lo].h; p->W = rq_table2[lo].w;
  p->L = p->Kp + p->S + p->H;
  p->P = p->L - p->W;
  p->U = p->P - p->H;
  p->B = p->W - p->S;
  p->P1 = p->P;
  while (!rq_is_prime(p->P1)) p->P1++;
  return 1;
}

#endif /* NRQ_RQ_MATH_H */
