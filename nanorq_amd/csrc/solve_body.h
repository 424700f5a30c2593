/*
 * solve_body.h -- the data stage of the precode solve for ONE column strip of ONE source block,
 * written as per-thread phase functions.  nrq_solve_kernel (nrq_device.hip) instantiates them inside the
 * gfx950 kernel (persistent 768- or 256-thread workgroups, `__syncthreads()` between phases); tests/emu
 * compiles the same functions with g++ and runs the threads of a phase in a loop, so the indexing, the
 * GF(256) bit tricks and the strip handling are exercised on the CPU build machine as well (the row pipeline
 * fwd_rows / fwd_rows_half exists on the device only: the emulator has its own row loop).
 *
 * What it replaces in the reference: precode_matrix_intermediate (precode.c:379-389 =
 * apply_sched :23-32 + permute :3-13), decode_row / APPLYROW (nanorq.c:8-13,:184-204) and the
 * oblas row kernels underneath (oaxpy/oscal), for every byte column of a block at once.
 *
 * Data layout: the strip's WB bytes of every constraint row ("slot") live in
 * LDS for the whole solve: M slots x WB bytes (8416 x 16 B = 131.5 KiB at K=8192), plus small
 * scratch regions.  All row operations of the solve are LDS<->register traffic; HBM is touched
 * once to load the received symbols and once to store results.
 *
 * GF(256) = RFC 6330 section 5.7 (x^8+x^4+x^3+x^2+1).  Arithmetic is carried out on 4 packed
 * bytes per dword: multiply-by-alpha is a shift/mask/xor ("xtime"), a general multiply is 8
 * conditional xtime-accumulates, and the HDPC block (reference precode.c:60-83, :232-252) is
 * applied through its MT*GAMMA factorisation as a Horner recurrence, not by table lookups.
 */
#ifndef NRQ_SOLVE_BODY_H
#define NRQ_SOLVE_BODY_H

#include <stdint.h>
#include <string.h>

#include "plan.h"

#if defined(__HIPCC__)
#define SB_HD __host__ __device__ __forceinline__
#define SB_MEM __host__ __device__ __forceinline__
#else
#define SB_HD static inline
#define SB_MEM inline
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
#endif

/* Pointers that come out of the job descriptor are generic as far as the compiler can tell, which
 * would turn every access into flat_load/flat_store (counted on BOTH vmcnt and lgkmcnt, so each LDS
 * wait would also drain them).  They all point to HBM: say so. */
#if defined(__HIP_DEVICE_COMPILE__)
#define NRQ_GAS __attribute__((address_space(1)))
#else
#define NRQ_GAS
#endif
template <class X> SB_HD const NRQ_GAS X *gptr(const void *p) { return (const NRQ_GAS X *)p; }
template <class X> SB_HD const NRQ_GAS X *gptr(uint64_t p) { return (const NRQ_GAS X *)(uintptr_t)p; }
template <class X> SB_HD NRQ_GAS X *gptr_w(uint64_t p) { return (NRQ_GAS X *)(uintptr_t)p; }

/* device-visible description of one block's work (filled by the host API) */
typedef struct nrq_job {
  uint64_t plan;     /* plan arena (plan.h) */
  uint64_t rowsrc;   /* u32[M]: slot -> NRQ_ROW_ZERO | j (row of src) | 0x80000000|q (row of rep) */
  uint64_t src;      /* source-symbol rows, pitch T */
  uint64_t rep;      /* repair-symbol rows, pitch T */
  uint64_t inter;    /* nullable: L x T intermediate symbols out */
  uint64_t out;      /* generated symbols out, pitch T */
  uint64_t out_cptr; /* u32[nout+1] */
  uint64_t out_slots;/* u16[]: slots (plan colslot[] of the LT neighbours) XORed into each generated symbol */
  uint64_t out_row;  /* u32[nout]: destination row in `out` */
  uint32_t nout;
  uint32_t pad;
} nrq_job;

/* true in every lane of the wave if the condition holds in one (the CPU emulation runs a thread at a time: the thread's own) */
#if defined(__HIP_DEVICE_COMPILE__)
#define NRQ_WAVE_ANY(x) (__ballot(x) != 0ull)
#else
#define NRQ_WAVE_ANY(x) (x)
#endif
/* A value that is the same in every lane of the wave, said so: it then lives in a scalar register (the descriptors of the
 * group being gathered / scattered are such values, live across all phases of a strip). */
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef NRQ_NO_UNIFORM
__device__ __forceinline__ uint32_t nrq_uniform(uint32_t x) { return x; }
#else
__device__ __forceinline__ uint32_t nrq_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
#endif
template <class P> __device__ __forceinline__ P nrq_uniform_ptr(P p) {
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint64_t u = (uint64_t)nrq_uniform((uint32_t)v) | ((uint64_t)nrq_uniform((uint32_t)(v >> 32)) << 32);
  return (P)(uintptr_t)u;
}
#else
SB_HD uint32_t nrq_uniform(uint32_t x) { return x; }
template <class P> SB_HD P nrq_uniform_ptr(P p) { return p; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define NRQ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define NRQ_SCHED_FENCE() do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define NRQ_MARK(c, i) do { if ((c).dbg && (c).dbg_t0) (c).dbg[i] = (unsigned long long)clock64(); } while (0)
#define NRQ_MARK_MAX(c, i) do { if ((c).dbg) atomicMax(&(c).dbg[i], (unsigned long long)clock64()); } while (0) /* (costs the marked waves: 768 atomics on one address are ~20 k clocks) */
#else
#define NRQ_MARK(c, i) do { } while (0)
#define NRQ_MARK_MAX(c, i) do { } while (0)
#endif
#define NRQ_ROW_ZERO 0xFFFFFFFFu
#define NRQ_ROW_REP 0x80000000u

/* ---- strip vector: WB bytes as packed dwords ---- */
template <int WB> struct SV {
  static constexpr int ND = (WB + 3) / 4;
  uint32_t w[ND];
};

/* a strip element's natural alignment: its width, 4 bytes for the 12-byte strip (three dwords; round 6: K ~ 8500-12000, whose
 * 16-byte image does not fit the LDS) */
template <int WB> struct SVAlign { static constexpr uintptr_t mask = (uintptr_t)(WB == 12 ? 3 : WB - 1); };
struct nrq_u3 { uint32_t x, y, z; };
/* strips that make a line group (staging buffers per set; the largest work slot): the strips that share a 128-byte line of a
 * row -- 8 for the 12-byte strip (a power of two: 96 bytes of every row, so a line is shared by the groups either side) */
#ifndef NRQ_W12_GROUP
#define NRQ_W12_GROUP 8u
#endif
SB_HD constexpr uint32_t nrq_group_strips(uint32_t wbe) { return wbe >= 128u ? 1u : wbe == 12u ? NRQ_W12_GROUP : 128u / wbe; }
template <int WB> SB_HD SV<WB> sv_zero() {
  SV<WB> r;
#pragma unroll
  for (int i = 0; i < SV<WB>::ND; i++) r.w[i] = 0;
  return r;
}
template <int WB> SB_HD void sv_xor(SV<WB> &a, const SV<WB> &b) {
#pragma unroll
  for (int i = 0; i < SV<WB>::ND; i++) a.w[i] ^= b.w[i];
}
template <int WB> SB_HD void sv_xor_masked(SV<WB> &a, const SV<WB> &b, uint32_t mask) {
#pragma unroll
  for (int i = 0; i < SV<WB>::ND; i++) a.w[i] ^= b.w[i] & mask;
}
/* multiply every byte by alpha (= 2) */
SB_HD uint32_t xtime32(uint32_t x) {
  /* bytes with the top bit set get the reduction 0x1d: (0x80 - 0x01) & 0x1d per such byte -- shifts, a subtract and
   * masks only (a 32-bit multiply by 0x1d is a quarter-rate instruction on CDNA) */
  const uint32_t hi = x & 0x80808080u;
  return ((x ^ hi) << 1) ^ ((hi - (hi >> 7)) & 0x1d1d1d1du);
}
/* three-input logic in one instruction (gfx950: v_bitop3_b32, truth table over a = 0xF0, b = 0xCC, c = 0xAA) */
SB_HD uint32_t nrq_xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}
SB_HD uint32_t nrq_xor_and(uint32_t a, uint32_t b, uint32_t c) { /* a ^ (b & c) */
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78);
#else
  return a ^ (b & c);
#endif
}
template <int WB> SB_HD SV<WB> sv_xtime(const SV<WB> &a) {
  SV<WB> r;
#pragma unroll
  for (int i = 0; i < SV<WB>::ND; i++) r.w[i] = xtime32(a.w[i]);
  return r;
}
/* every byte times coef (coef may differ per thread) */
template <int WB> SB_HD SV<WB> sv_mul(SV<WB> v, uint32_t coef) {
  SV<WB> acc = sv_zero<WB>();
#pragma unroll
  for (int b = 0; b < 8; b++) {
    uint32_t m = 0u - ((coef >> b) & 1u);
    sv_xor_masked<WB>(acc, v, m);
    v = sv_xtime<WB>(v);
  }
  return acc;
}

/* ---- LDS access (plain memory in the emulator) ----
 * G > 1 ("wide strips", 16-byte lanes only): an element is G adjacent 16-byte columns, each handled by its own lane;
 * element idx of lane `sub` lies at base + idx * (16 * G) + sub * 16 -- the lane's column offset is folded into the
 * base pointer (StripCtx), the accessors only scale the index. */
template <int WB, int G = 1> SB_HD SV<WB> lds_get(const uint8_t *lds, uint32_t idx) {
  static_assert(G == 1 || WB == 16, "wide strips are made of 16-byte lanes");
  SV<WB> r;
  if constexpr (WB == 16) {
    const uint4 *p = reinterpret_cast<const uint4 *>(lds) + (size_t)idx * G;
    uint4 v = *p;
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  } else if constexpr (WB == 12) {
    const nrq_u3 v = reinterpret_cast<const nrq_u3 *>(lds)[idx];
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z;
  } else if constexpr (WB == 8) {
    const uint2 *p = reinterpret_cast<const uint2 *>(lds) + idx;
    uint2 v = *p;
    r.w[0] = v.x; r.w[1] = v.y;
  } else if constexpr (WB == 4) {
    r.w[0] = reinterpret_cast<const uint32_t *>(lds)[idx];
  } else {
    r.w[0] = reinterpret_cast<const uint16_t *>(lds)[idx];
  }
  return r;
}
template <int WB, int G = 1> SB_HD void lds_put(uint8_t *lds, uint32_t idx, const SV<WB> &v) {
  static_assert(G == 1 || WB == 16, "wide strips are made of 16-byte lanes");
  if constexpr (WB == 16) {
    uint4 t; t.x = v.w[0]; t.y = v.w[1]; t.z = v.w[2]; t.w = v.w[3];
    reinterpret_cast<uint4 *>(lds)[(size_t)idx * G] = t;
  } else if constexpr (WB == 12) {
    nrq_u3 t; t.x = v.w[0]; t.y = v.w[1]; t.z = v.w[2];
    reinterpret_cast<nrq_u3 *>(lds)[idx] = t;
  } else if constexpr (WB == 8) {
    uint2 t; t.x = v.w[0]; t.y = v.w[1];
    reinterpret_cast<uint2 *>(lds)[idx] = t;
  } else if constexpr (WB == 4) {
    reinterpret_cast<uint32_t *>(lds)[idx] = v.w[0];
  } else {
    reinterpret_cast<uint16_t *>(lds)[idx] = (uint16_t)v.w[0];
  }
}
/* slot ^= v, safe against other threads of the workgroup doing the same to the same slot */
template <int WB, int G = 1> SB_HD void lds_xor(uint8_t *lds, uint32_t idx, const SV<WB> &v) {
  static_assert(G == 1 || WB == 16, "wide strips are made of 16-byte lanes");
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (WB == 16) {
    unsigned long long *p = reinterpret_cast<unsigned long long *>(lds) + 2 * (size_t)idx * G;
    atomicXor(p, (unsigned long long)v.w[0] | ((unsigned long long)v.w[1] << 32));
    atomicXor(p + 1, (unsigned long long)v.w[2] | ((unsigned long long)v.w[3] << 32));
  } else if constexpr (WB == 12) { /* (slots on 4-byte boundaries: three 32-bit atomics) */
    unsigned int *p = reinterpret_cast<unsigned int *>(lds) + 3 * (size_t)idx;
    atomicXor(p, v.w[0]); atomicXor(p + 1, v.w[1]); atomicXor(p + 2, v.w[2]);
  } else if constexpr (WB == 8) {
    unsigned long long *p = reinterpret_cast<unsigned long long *>(lds) + idx;
    atomicXor(p, (unsigned long long)v.w[0] | ((unsigned long long)v.w[1] << 32));
  } else if constexpr (WB == 4) {
    atomicXor(reinterpret_cast<unsigned int *>(lds) + idx, v.w[0]);
  } else {
    atomicXor(reinterpret_cast<unsigned int *>(lds) + (idx >> 1), v.w[0] << ((idx & 1u) * 16u));
  }
#else
  SV<WB> t = lds_get<WB, G>(lds, idx);
  sv_xor<WB>(t, v);
  lds_put<WB, G>(lds, idx, t);
#endif
}

/* ---- global strip access: `valid` bytes (<= WB) at p ---- */
template <int WB> SB_HD SV<WB> g_get(const NRQ_GAS uint8_t *p, uint32_t valid) {
  SV<WB> r = sv_zero<WB>();
  if (valid == (uint32_t)WB && (reinterpret_cast<uintptr_t>(p) & SVAlign<WB>::mask) == 0) {
    if constexpr (WB == 16) {
      uint4 v = *reinterpret_cast<const NRQ_GAS uint4 *>(p);
      r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else if constexpr (WB == 12) {
      const NRQ_GAS uint32_t *q = reinterpret_cast<const NRQ_GAS uint32_t *>(p);
      r.w[0] = q[0]; r.w[1] = q[1]; r.w[2] = q[2];
    } else if constexpr (WB == 8) {
      uint2 v = *reinterpret_cast<const NRQ_GAS uint2 *>(p);
      r.w[0] = v.x; r.w[1] = v.y;
    } else if constexpr (WB == 4) {
      r.w[0] = *reinterpret_cast<const NRQ_GAS uint32_t *>(p);
    } else {
      r.w[0] = *reinterpret_cast<const NRQ_GAS uint16_t *>(p);
    }
  } else {
    /* (constant indices: a run-time index into r.w would move every SV the caller holds into scratch memory) */
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)WB; k++)
      if (k < valid) r.w[k >> 2] |= (uint32_t)p[k] << ((k & 3u) * 8u);
  }
  return r;
}
/* an aligned, full-width strip element straight from L2 (staging data written earlier by this workgroup: never an
 * older copy out of the CU's vector L1) */
template <int WB> SB_HD SV<WB> g_get_l2(const NRQ_GAS uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  SV<WB> r = sv_zero<WB>();
  if constexpr (WB == 16) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u4 *>(p));
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  } else if constexpr (WB == 12) {
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    typedef u3 u3a __attribute__((aligned(4)));
    const u3 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u3a *>(p));
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z;
  } else if constexpr (WB == 8) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u2 *>(p));
    r.w[0] = v.x; r.w[1] = v.y;
  } else if constexpr (WB == 4) {
    r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS uint32_t *>(p));
  } else {
    r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS uint16_t *>(p));
  }
  return r;
#else
  return g_get<WB>(p, WB);
#endif
}
/* CB bytes of a symbol row from an address aligned to the strip width only (declared 2-byte aligned; the memory path takes
 * unaligned vector loads) */
template <int CB, int AL> SB_HD SV<CB> g_get_chunk(const NRQ_GAS uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  SV<CB> r = sv_zero<CB>();
  if constexpr (CB == 16) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef u4 u4a __attribute__((aligned(2)));
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u4a *>(p));
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  } else if constexpr (CB == 8) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    typedef u2 u2a __attribute__((aligned(2)));
    const u2 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u2a *>(p));
    r.w[0] = v.x; r.w[1] = v.y;
  } else {
    typedef uint32_t u1a __attribute__((aligned(2)));
    r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u1a *>(p));
  }
  return r;
#else
  return g_get<CB>(p, CB);
#endif
}
template <int WB> SB_HD void g_put(NRQ_GAS uint8_t *p, uint32_t valid, const SV<WB> &v) {
  if (valid == (uint32_t)WB && (reinterpret_cast<uintptr_t>(p) & SVAlign<WB>::mask) == 0) {
    if constexpr (WB == 16) {
      uint4 t; t.x = v.w[0]; t.y = v.w[1]; t.z = v.w[2]; t.w = v.w[3];
      *reinterpret_cast<NRQ_GAS uint4 *>(p) = t;
    } else if constexpr (WB == 12) {
      NRQ_GAS uint32_t *q = reinterpret_cast<NRQ_GAS uint32_t *>(p);
      q[0] = v.w[0]; q[1] = v.w[1]; q[2] = v.w[2];
    } else if constexpr (WB == 8) {
      uint2 t; t.x = v.w[0]; t.y = v.w[1];
      *reinterpret_cast<NRQ_GAS uint2 *>(p) = t;
    } else if constexpr (WB == 4) {
      *reinterpret_cast<NRQ_GAS uint32_t *>(p) = v.w[0];
    } else {
      *reinterpret_cast<NRQ_GAS uint16_t *>(p) = (uint16_t)v.w[0];
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)WB; k++)
      if (k < valid) p[k] = (uint8_t)(v.w[k >> 2] >> ((k & 3u) * 8u));
  }
}

/* 12-byte strips, the movers' aligned form: T a multiple of 4 and rows on 4-byte boundaries, so a strip element is 1-3 whole
 * dwords (T = 1280: the 107th strip holds 8 bytes).  Branch-free: a dword beyond `valid` re-reads the first and is dropped. */
SB_HD SV<12> g_get_al12(const NRQ_GAS uint8_t *p, uint32_t valid) {
  const NRQ_GAS uint32_t *q = reinterpret_cast<const NRQ_GAS uint32_t *>(p);
  SV<12> r = sv_zero<12>();
  const bool h1 = valid > 4u, h2 = valid > 8u;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + (h1 ? 1 : 0)), c = __builtin_nontemporal_load(q + (h2 ? 2 : 0));
#else
  const uint32_t a = q[0], b = q[h1 ? 1 : 0], c = q[h2 ? 2 : 0];
#endif
  r.w[0] = a; r.w[1] = h1 ? b : 0u; r.w[2] = h2 ? c : 0u;
  return r;
}
SB_HD void g_put_al12(NRQ_GAS uint8_t *p, uint32_t valid, const SV<12> &v) {
  NRQ_GAS uint32_t *q = reinterpret_cast<NRQ_GAS uint32_t *>(p);
  q[0] = v.w[0];
  if (valid > 4u) q[1] = v.w[1];
  if (valid > 8u) q[2] = v.w[2];
}
template <int WB> SB_HD void g_put_al(NRQ_GAS uint8_t *p, const SV<WB> &v) { /* a whole, aligned strip element */
  if constexpr (WB == 16) {
    uint4 t; t.x = v.w[0]; t.y = v.w[1]; t.z = v.w[2]; t.w = v.w[3];
    *reinterpret_cast<NRQ_GAS uint4 *>(p) = t;
  } else if constexpr (WB == 12) {
    NRQ_GAS uint32_t *q = reinterpret_cast<NRQ_GAS uint32_t *>(p);
    q[0] = v.w[0]; q[1] = v.w[1]; q[2] = v.w[2];
  } else if constexpr (WB == 8) {
    uint2 t; t.x = v.w[0]; t.y = v.w[1];
    *reinterpret_cast<NRQ_GAS uint2 *>(p) = t;
  } else if constexpr (WB == 4) {
    *reinterpret_cast<NRQ_GAS uint32_t *>(p) = v.w[0];
  } else {
    *reinterpret_cast<NRQ_GAS uint16_t *>(p) = (uint16_t)v.w[0];
  }
}

/* Streaming accesses (symbol rows read once, staging buffers, results written once): marked non-temporal so that
 * they do not push the plans -- read over and over by every strip of a block -- out of L2. */
template <int WB> SB_HD SV<WB> g_get_stream(const NRQ_GAS uint8_t *p, uint32_t valid) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (valid == (uint32_t)WB && (reinterpret_cast<uintptr_t>(p) & SVAlign<WB>::mask) == 0) {
    SV<WB> r = sv_zero<WB>();
    if constexpr (WB == 16) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const u4 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u4 *>(p));
      r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else if constexpr (WB == 12) {
      typedef uint32_t u3 __attribute__((ext_vector_type(3)));
      typedef u3 u3a __attribute__((aligned(4)));
      const u3 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u3a *>(p));
      r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z;
    } else if constexpr (WB == 8) {
      typedef uint32_t u2 __attribute__((ext_vector_type(2)));
      const u2 v = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS u2 *>(p));
      r.w[0] = v.x; r.w[1] = v.y;
    } else if constexpr (WB == 4) {
      r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS uint32_t *>(p));
    } else {
      r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const NRQ_GAS uint16_t *>(p));
    }
    return r;
  }
#endif
  return g_get<WB>(p, valid);
}
template <int WB> SB_HD void g_put_stream(NRQ_GAS uint8_t *p, uint32_t valid, const SV<WB> &v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (valid == (uint32_t)WB && (reinterpret_cast<uintptr_t>(p) & SVAlign<WB>::mask) == 0) {
    if constexpr (WB == 16) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const u4 t = {v.w[0], v.w[1], v.w[2], v.w[3]};
      __builtin_nontemporal_store(t, reinterpret_cast<NRQ_GAS u4 *>(p));
    } else if constexpr (WB == 12) {
      typedef uint32_t u3 __attribute__((ext_vector_type(3)));
      typedef u3 u3a __attribute__((aligned(4)));
      const u3 t = {v.w[0], v.w[1], v.w[2]};
      __builtin_nontemporal_store(t, reinterpret_cast<NRQ_GAS u3a *>(p));
    } else if constexpr (WB == 8) {
      typedef uint32_t u2 __attribute__((ext_vector_type(2)));
      const u2 t = {v.w[0], v.w[1]};
      __builtin_nontemporal_store(t, reinterpret_cast<NRQ_GAS u2 *>(p));
    } else if constexpr (WB == 4) {
      __builtin_nontemporal_store(v.w[0], reinterpret_cast<NRQ_GAS uint32_t *>(p));
    } else {
      __builtin_nontemporal_store((uint16_t)v.w[0], reinterpret_cast<NRQ_GAS uint16_t *>(p));
    }
    return;
  }
#endif
  g_put<WB>(p, valid, v);
}

/* ---- LDS carve-up, identical on host (launch sizing) and device ---- */
typedef struct nrq_lds_layout {
  uint32_t off_slots, off_cu, off_x, total;
} nrq_lds_layout;

SB_HD uint32_t nrq_r16(uint32_t x) { return (x + 15u) & ~15u; }

SB_HD nrq_lds_layout nrq_lds_plan(const nrq_plan_hdr *h, uint32_t WB) {
  nrq_lds_layout l;
  uint32_t o = 0;
  o = NRQ_SCRATCH * WB;                                  /* per-lane scratch slots of the padding ops */
  l.off_slots = o; o = nrq_r16(o + (h->M + h->r2) * WB); /* M slots, then the r2 scratch rows E_p */
  l.off_cu = o;    o = nrq_r16(o + (h->u ? h->u : 1u) * WB);
  /* region X: the free-column accumulators Cf during the dense stage, then the 4-bit XOR tables */
  uint32_t cf = NRQ_MAX_FREE * WB;
  uint32_t t4 = h->wpr * 8u * 16u * (WB == 12u ? 16u : WB); /* (12-byte strips: entries on 16-byte boundaries, one LDS read each) */
  l.off_x = o;
  o = nrq_r16(o + (cf > t4 ? cf : t4));
  l.total = o;
  return l;
}

/* ---- per-workgroup context (uniform across the threads of a strip) ---- */
/* (G > 1: `lds` already carries the lane's column offset sub * 16, `lay` is the layout of the 16*G-byte wide image, `valid`
 * the bytes of the lane's own column inside T, and the staging pointers the kernel passes to the phases are offset alike) */
template <int WB, int G = 1> struct StripCtx {
  const uint8_t *plan;
  const nrq_plan_hdr *h;
  const uint8_t *kc; /* nrq_kconst_hdr arena of this K' */
  const nrq_job *job; /* (a pointer, not a copy: the workgroup context stays in few registers) */
  uint8_t *lds;
  nrq_lds_layout lay;
  uint32_t T, strip, valid; /* valid = bytes of this strip inside T */
  unsigned long long *dbg = nullptr; /* NRQ_PROF: 7 clock marks (thread 0's, or the latest thread's) */
  bool dbg_t0 = false;
  SB_MEM uint8_t *slots() const { return lds + lay.off_slots; }
  SB_MEM uint8_t *cu() const { return lds + lay.off_cu; }
  SB_MEM uint8_t *cf() const { return lds + lay.off_x; }
  SB_MEM uint8_t *t4() const { return lds + lay.off_x; }
  template <class X> SB_MEM const NRQ_GAS X *arr(uint32_t off) const { return gptr<X>(plan + off); }
};

/* phase 0: bring the strip of every slot into LDS; clear the scratch rows and region X.
 * A strip is WB bytes out of every symbol row: fetched on its own, every 16-byte piece costs a whole 128-byte
 * line of HBM/L2 traffic (measured: ~26 such requests per clock for the whole chip, i.e. HBM-bound at one line
 * per piece).  So the persistent solve kernel works on LINE GROUPS -- the 128/WB strips that share a line of
 * every row: while it solves the strips of one group one after the other, the waves that are idle during the
 * forward passes gather the next group, whole lines at a time, into per-strip staging buffers (global memory,
 * strip-major, contiguous), from where a strip image is filled by coalesced loads.
 *   pf_gather    units [u0, u1) of the gather, unit = (row, piece of the line); thread p of np;
 *   pf_commit    staging buffer -> LDS image;
 *   ph_clear     the scratch rows E_p, the per-lane scratch slots and region X. */
template <int WB> struct GroupSrc { /* where the rows of one line group of one block come from */
  const NRQ_GAS uint32_t *rowsrc;
  const NRQ_GAS uint8_t *src, *rep;
  uint32_t M, T, strip0, nstrips; /* first strip of the group; strips of the block */
  uint32_t lsub;                  /* log2 of the strips the group has (a whole line: 128/WB; fewer when work is scarce) */
};
/* (G > 1: a unit is moved by the G lanes of a virtual thread p of np; `sub` is the lane's 16-byte column of the wide strip) */
/* AL: every piece is a whole, aligned strip element (T a multiple of the strip width, rows aligned): the loop then holds no
 * byte-wise path at all -- with one in it, the compiler waits for ALL outstanding loads wherever the paths join, and the
 * pieces of a trip, meant to be in flight together, are fetched one memory latency after the other */
/* Narrow strips (2, 4, 8 bytes): a request per strip piece moves 2-8 bytes, and the gather is bound by its requests in flight
 * against the memory latency -- at K=27000 (4-byte strips) and K'=56403 (2-byte strips) the movers, not the forward waves or
 * the HDPC phase, bounded both of their windows.  The strips of a line group are neighbours in the symbol row, so one load
 * of CB = 4, 8 or 16 bytes brings NP = CB / WB of them; the pieces then go to their strips' staging buffers.  Needs
 * every chunk of a row whole (T a multiple of CB) and rows that start on a strip boundary.  A range [u0, u1) of pieces owns
 * the chunks whose first piece it holds: consecutive ranges tile the chunks as they tile the pieces. */
template <int WB, int CB> SB_HD SV<WB> chunk_piece(const SV<CB> &v, int j) {
  SV<WB> r = sv_zero<WB>();
  if constexpr (WB == 8) { r.w[0] = v.w[2 * j]; r.w[1] = v.w[2 * j + 1]; }
  else if constexpr (WB == 4) r.w[0] = v.w[j];
  else r.w[0] = (v.w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
  return r;
}
template <int WB, int CB, bool PIPELINED> SB_HD void pf_gather_chunks(const GroupSrc<WB> &g, NRQ_GAS uint8_t *stage, size_t stage_stride, uint32_t u0,
                                                                      uint32_t u1, uint32_t p, uint32_t np) {
#ifndef NRQ_GATHER_CHUNK_PB
#define NRQ_GATHER_CHUNK_PB 4 /* chunk loads a mover thread has in flight per trip */
#endif
  constexpr int NP = CB / WB, LNP = NP == 8 ? 3 : NP == 4 ? 2 : 1, PB = NRQ_GATHER_CHUNK_PB;
  const uint32_t lc = g.lsub - (uint32_t)LNP, cmask = (1u << lc) - 1u; /* chunk unit c = (row c >> lc, chunk c & cmask) */
  const uint32_t c0 = (u0 + NP - 1u) >> LNP, c1 = (u1 + NP - 1u) >> LNP;
  auto fetch = [&](uint32_t cu, uint32_t src) {
    const uint32_t strip = g.strip0 + ((cu & cmask) << LNP);
    SV<CB> v = sv_zero<CB>();
    if (cu < c1 && src != NRQ_ROW_ZERO && strip < g.nstrips) {
      const NRQ_GAS uint8_t *b = (src & NRQ_ROW_REP) ? g.rep + (size_t)(src & 0x7FFFFFFFu) * g.T : g.src + (size_t)src * g.T;
      v = g_get_chunk<CB, WB>(b + (size_t)strip * WB);
    }
    return v;
  };
  auto put = [&](uint32_t cu, const SV<CB> &v) {
    if (cu >= c1) return;
    NRQ_GAS uint8_t *d = stage + (size_t)((cu & cmask) << LNP) * stage_stride + (size_t)(cu >> lc) * WB;
#pragma unroll
    for (int j = 0; j < NP; j++) g_put_stream<WB>(d + (size_t)j * stage_stride, WB, chunk_piece<WB, CB>(v, j));
  };
  const uint32_t first = c0 + p, step = (uint32_t)PB * np;
  if (first >= c1) return;
  if constexpr (!PIPELINED) {
    for (uint32_t base = first; base < c1; base += step) {
      uint32_t sr[PB];
      SV<CB> v[PB];
#pragma unroll
      for (int q = 0; q < PB; q++) { const uint32_t cu = base + (uint32_t)q * np; sr[q] = cu < c1 ? g.rowsrc[cu >> lc] : NRQ_ROW_ZERO; }
#pragma unroll
      for (int q = 0; q < PB; q++) v[q] = fetch(base + (uint32_t)q * np, sr[q]);
#pragma unroll
      for (int q = 0; q < PB; q++) put(base + (uint32_t)q * np, v[q]);
    }
    return;
  }
  uint32_t s_cur[PB], s_nxt[PB];
  SV<CB> v_prev[PB], v_cur[PB];
#pragma unroll
  for (int q = 0; q < PB; q++) { const uint32_t cu = first + (uint32_t)q * np; s_cur[q] = cu < c1 ? g.rowsrc[cu >> lc] : NRQ_ROW_ZERO; }
  uint32_t prev = 0;
  bool have_prev = false;
  for (uint32_t base = first; base < c1; base += step) {
#pragma unroll
    for (int q = 0; q < PB; q++) { const uint32_t cu = base + step + (uint32_t)q * np; s_nxt[q] = cu < c1 ? g.rowsrc[cu >> lc] : NRQ_ROW_ZERO; }
#pragma unroll
    for (int q = 0; q < PB; q++) v_cur[q] = fetch(base + (uint32_t)q * np, s_cur[q]);
    if (have_prev) {
#pragma unroll
      for (int q = 0; q < PB; q++) put(prev + (uint32_t)q * np, v_prev[q]);
    }
#pragma unroll
    for (int q = 0; q < PB; q++) { v_prev[q] = v_cur[q]; s_cur[q] = s_nxt[q]; }
    prev = base;
    have_prev = true;
  }
#pragma unroll
  for (int q = 0; q < PB; q++) put(prev + (uint32_t)q * np, v_prev[q]);
}
#ifndef NRQ_GATHER_CHUNKS
#define NRQ_GATHER_CHUNKS 1
#endif
/* AL: 0 byte-wise, 1 whole aligned elements, 2 (12-byte strips) whole dwords of an element that may end early -- the form of the
 * one line group that holds a row's last, partial strip (T = 1280: 106 strips and 8 bytes); a request per dword there, one per
 * element everywhere else (every request of a wave touches 64 lines: the CU's cache looks up a line per clock) */
template <int WB, int G, bool PIPELINED, int AL> SB_HD void pf_gather_impl(const GroupSrc<WB> &g, NRQ_GAS uint8_t *stage, size_t stage_stride, uint32_t u0, uint32_t u1,
                                       uint32_t p, uint32_t np, uint32_t sub) {
  const uint32_t lsub = g.lsub, pmask = (1u << lsub) - 1u; /* unit u = (row u >> lsub, piece u & pmask) */
  if constexpr (WB == 12 && AL == 1) {
    if ((g.strip0 + (1u << lsub)) * 12u > g.T) { pf_gather_impl<WB, G, PIPELINED, 2>(g, stage, stage_stride, u0, u1, p, np, sub); return; }
  }
#if NRQ_GATHER_CHUNKS
  if constexpr (G == 1 && WB < 16 && WB != 12) {
    const uint32_t span = (uint32_t)WB << lsub, cb = span >= 16u ? 16u : span; /* bytes of a row that a line group holds; of a chunk */
    const bool whole = cb > (uint32_t)WB && g.T % cb == 0u &&
                       ((reinterpret_cast<uintptr_t>(g.src) | reinterpret_cast<uintptr_t>(g.rep)) & (uintptr_t)(WB - 1)) == 0;
    if (whole) {
      if (cb == 16u) pf_gather_chunks<WB, 16, PIPELINED>(g, stage, stage_stride, u0, u1, p, np);
      else if constexpr (WB < 8) {
        if (cb == 8u) pf_gather_chunks<WB, 8, PIPELINED>(g, stage, stage_stride, u0, u1, p, np);
        else if constexpr (WB < 4) pf_gather_chunks<WB, 4, PIPELINED>(g, stage, stage_stride, u0, u1, p, np);
      }
      return;
    }
  }
#endif
#ifndef NRQ_GATHER_PB
#define NRQ_GATHER_PB 4
#endif
#ifndef NRQ_W12_GATHER_PB
#define NRQ_W12_GATHER_PB NRQ_GATHER_PB
#endif
  constexpr int PB = WB == 12 ? NRQ_W12_GATHER_PB : NRQ_GATHER_PB; /* requests in flight per thread and stage: few -- a deep queue of gather requests in the CU's
                                     * memory pipeline delays the op words wave 0 is waiting for */
  if constexpr (!PIPELINED) { /* the register-lean form (the 256- and 64-thread workgroups: 96-128 registers per thread) */
  for (uint32_t base = u0 + p; base < u1; base += PB * np) {
      uint32_t s[PB];
      SV<WB> v[PB];
  #pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t u = base + (uint32_t)q * np;
        s[q] = u < u1 ? g.rowsrc[u >> lsub] : NRQ_ROW_ZERO;
      }
  #pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t u = base + (uint32_t)q * np, strip = g.strip0 + (u & pmask);
        v[q] = sv_zero<WB>();
        if (s[q] != NRQ_ROW_ZERO && strip < g.nstrips) {
          const NRQ_GAS uint8_t *b = (s[q] & NRQ_ROW_REP) ? g.rep + (size_t)(s[q] & 0x7FFFFFFFu) * g.T : g.src + (size_t)s[q] * g.T;
          if constexpr (AL == 2) {
            const uint32_t rem = g.T - strip * WB;
            v[q] = g_get_al12(b + (size_t)strip * WB, rem < 12u ? rem : 12u);
          } else if constexpr (AL == 1) {
            v[q] = g_get_l2<WB>(b + (size_t)strip * WB);
          } else if constexpr (G == 1) {
            const uint32_t rem = g.T - strip * WB;
            v[q] = g_get_stream<WB>(b + (size_t)strip * WB, rem < (uint32_t)WB ? rem : (uint32_t)WB);
          } else {
            const uint32_t at = strip * (WB * G) + sub * WB, rem = at < g.T ? g.T - at : 0u;
            if (rem) v[q] = g_get_stream<WB>(b + at, rem < (uint32_t)WB ? rem : (uint32_t)WB);
          }
        }
      }
  #pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t u = base + (uint32_t)q * np;
        if (u < u1) g_put_stream<WB>(stage + (size_t)(u & pmask) * stage_stride + (size_t)(u >> lsub) * (WB * G) + sub * WB, WB, v[q]);
      }
    }
    return;
  }
  /* Three stages, each a trip to memory (row map -> symbol piece -> staging buffer), run as a software pipeline: the row
   * map entries of trip t + 1 and the pieces of trip t are requested before the pieces of trip t - 1 are stored, so a trip
   * costs one memory latency, not three in a row (vector memory operations of a wave complete in order: waiting for a
   * trip's loads right after its stores waits for the stores too). */
  auto unit_of = [&](uint32_t base, int q) { return base + (uint32_t)q * np; };
  auto fetch = [&](uint32_t u, uint32_t src) {
    const uint32_t strip = g.strip0 + (u & pmask);
    SV<WB> v = sv_zero<WB>();
    if (u < u1 && src != NRQ_ROW_ZERO && strip < g.nstrips) {
      const NRQ_GAS uint8_t *b = (src & NRQ_ROW_REP) ? g.rep + (size_t)(src & 0x7FFFFFFFu) * g.T : g.src + (size_t)src * g.T;
      if constexpr (AL == 2) {
        const uint32_t rem = g.T - strip * WB;
        v = g_get_al12(b + (size_t)strip * WB, rem < 12u ? rem : 12u);
      } else if constexpr (AL == 1) {
        v = g_get_l2<WB>(b + (size_t)strip * WB);
      } else if constexpr (G == 1) {
        const uint32_t rem = g.T - strip * WB;
        v = g_get_stream<WB>(b + (size_t)strip * WB, rem < (uint32_t)WB ? rem : (uint32_t)WB);
      } else {
        const uint32_t at = strip * (WB * G) + sub * WB, rem = at < g.T ? g.T - at : 0u;
        if (rem) v = g_get_stream<WB>(b + at, rem < (uint32_t)WB ? rem : (uint32_t)WB);
      }
    }
    return v;
  };
  const uint32_t first = u0 + p, step = (uint32_t)PB * np;
  if (first >= u1) return;
  uint32_t s_cur[PB], s_nxt[PB];
  SV<WB> v_prev[PB], v_cur[PB];
#pragma unroll
  for (int q = 0; q < PB; q++) { const uint32_t u = unit_of(first, q); s_cur[q] = u < u1 ? g.rowsrc[u >> lsub] : NRQ_ROW_ZERO; }
  uint32_t prev = 0; /* base of the trip whose pieces are in v_prev (none yet) */
  bool have_prev = false;
  for (uint32_t base = first; base < u1; base += step) {
#pragma unroll
    for (int q = 0; q < PB; q++) { const uint32_t u = unit_of(base + step, q); s_nxt[q] = u < u1 ? g.rowsrc[u >> lsub] : NRQ_ROW_ZERO; }
#pragma unroll
    for (int q = 0; q < PB; q++) v_cur[q] = fetch(unit_of(base, q), s_cur[q]);
    if (have_prev) {
#pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t u = unit_of(prev, q);
        if (u < u1) g_put_stream<WB>(stage + (size_t)(u & pmask) * stage_stride + (size_t)(u >> lsub) * (WB * G) + sub * WB, WB, v_prev[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < PB; q++) { v_prev[q] = v_cur[q]; s_cur[q] = s_nxt[q]; }
    prev = base;
    have_prev = true;
  }
#pragma unroll
  for (int q = 0; q < PB; q++) {
    const uint32_t u = unit_of(prev, q);
    if (u < u1) g_put_stream<WB>(stage + (size_t)(u & pmask) * stage_stride + (size_t)(u >> lsub) * (WB * G) + sub * WB, WB, v_prev[q]);
  }
}
template <int WB, int G = 1, bool PIPELINED = false> SB_HD void pf_gather(const GroupSrc<WB> &g, NRQ_GAS uint8_t *stage, size_t stage_stride, uint32_t u0, uint32_t u1,
                                       uint32_t p, uint32_t np, uint32_t sub = 0) {
#ifndef NRQ_NO_AL
  if constexpr (G == 1 && WB >= 4) {
    const bool al = g.T % (uint32_t)(WB == 12 ? 4 : WB) == 0u && ((reinterpret_cast<uintptr_t>(g.src) | reinterpret_cast<uintptr_t>(g.rep)) & SVAlign<WB>::mask) == 0;
    if (al) { pf_gather_impl<WB, G, PIPELINED, true>(g, stage, stage_stride, u0, u1, p, np, sub); return; }
  }
#endif
  pf_gather_impl<WB, G, PIPELINED, false>(g, stage, stage_stride, u0, u1, p, np, sub);
}
#ifndef NRQ_COMMIT_PB
#define NRQ_COMMIT_PB 4
#endif
/* PB: loads in flight per thread (8 in the multi-wave workgroups: the 8.4 k rows of a K=8192 image are two trips of a 768-thread
 * workgroup instead of three -- solve 6.38-6.54 / 5.99-6.02 -> 6.36-6.37 / 5.93 ms, K=1000 8.43 / 7.85 -> 8.17 / 7.76; the single-wave
 * variant, 18 workgroups per CU, is 2 % slower with 8 and keeps 4) */
template <int WB, int G = 1, int PB = NRQ_COMMIT_PB> SB_HD void pf_commit(const StripCtx<WB, G> &c, const NRQ_GAS uint8_t *stage, uint32_t r0, uint32_t tid, uint32_t nt) {
  const uint32_t M = c.h->M;
  for (uint32_t base = r0 + tid; base < M; base += PB * nt) {
    SV<WB> v[PB];
#pragma unroll
    for (int q = 0; q < PB; q++) {
      const uint32_t r = base + (uint32_t)q * nt;
      v[q] = r < M ? g_get_l2<WB>(stage + (size_t)r * (WB * G)) : sv_zero<WB>();
    }
#pragma unroll
    for (int q = 0; q < PB; q++) {
      const uint32_t r = base + (uint32_t)q * nt;
      if (r < M) lds_put<WB, G>(c.slots(), r, v[q]);
    }
  }
}
template <int WB, int G = 1> SB_HD void ph_clear(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const uint32_t M = c.h->M;
  for (uint32_t p = tid; p < NRQ_SCRATCH; p += nt) lds_put<WB, G>(c.lds, p, sv_zero<WB>());
  for (uint32_t p = tid; p < c.h->r2; p += nt) lds_put<WB, G>(c.slots(), M + p, sv_zero<WB>());
  /* region X starts as the private HDPC accumulators (ph_hdpc), which ph_hdpc_reduce leaves zeroed for Cf */
  const uint32_t nx = (c.lay.total - c.lay.off_x) / (WB * G);
  for (uint32_t f = tid; f < nx; f += nt) lds_put<WB, G>(c.cf(), f, sv_zero<WB>());
}

/* phases 1+2: the forward passes (X^-1 on the peeled rows, the leftover rows, the GF(2) combinations of the
 * dense stage) are rows of 64 XOR ops that one wave runs as a software pipeline (plan.h): ph_row_read fetches
 * the source strips of a row, ph_row_apply XORs them into the targets NRQ_PIPE rows later.  Op fields are
 * slot + NRQ_SCRATCH, i.e. indices from the start of the LDS image; padding ops touch the lane's scratch slot.
 * On the GPU the strip travels as one native vector register group and the LDS byte address is formed by a
 * single sub-dword shift per field; the LDS image starts at LDS address 0 (the kernel has no static LDS). */
#if defined(__HIP_DEVICE_COMPILE__)
template <int WB> struct RowVal;
template <> struct RowVal<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct RowVal<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct RowVal<4> { typedef uint32_t type; };
template <> struct RowVal<2> { typedef uint32_t type; };
#define NRQ_LDSP(T, addr) (reinterpret_cast<__attribute__((address_space(3))) T *>((uintptr_t)(addr)))
template <int WB> __device__ __forceinline__ uint32_t row_addr_hi(uint32_t op) { /* (op >> 16) * WB */
  constexpr uint32_t sh = WB == 16 ? 4 : WB == 8 ? 3 : WB == 4 ? 2 : 1;
  uint32_t r, s = sh;
  if constexpr (WB == 12) { /* (no shift does it: the 24-bit multiply has the same sub-dword operand select) */
    s = 12u;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(s), "v"(op));
    return r;
  }
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(s), "v"(op));
  return r;
}
template <int WB> __device__ __forceinline__ uint32_t row_addr_lo(uint32_t op) { /* (op & 0xFFFF) * WB */
  constexpr uint32_t sh = WB == 16 ? 4 : WB == 8 ? 3 : WB == 4 ? 2 : 1;
  uint32_t r, s = sh;
  if constexpr (WB == 12) {
    s = 12u;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(s), "v"(op));
    return r;
  }
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(s), "v"(op));
  return r;
}
__device__ __forceinline__ uint32_t t4_addr(uint32_t base, uint32_t x, uint32_t byte) { /* base + byte `byte` of x */
  uint32_t r;
  if (byte == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(base), "v"(x));
  else if (byte == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(base), "v"(x));
  else if (byte == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(base), "v"(x));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(base), "v"(x));
  return r;
}
template <int WB> __device__ __forceinline__ typename RowVal<WB>::type ph_row_read(const StripCtx<WB> &, uint32_t op) {
  const uint32_t a = row_addr_hi<WB>(op);
  if constexpr (WB == 2) return *NRQ_LDSP(uint16_t, a);
  else return *NRQ_LDSP(typename RowVal<WB>::type, a);
}
template <int WB, class V> __device__ __forceinline__ void row_apply_at(uint32_t a, V v) { /* LDS bytes at a ^= v (V: the whole strip, or half of it) */
  if constexpr (sizeof(V) == 16) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const u2 lo = {v.x, v.y}, hi = {v.z, v.w};
    __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, a), __builtin_bit_cast(unsigned long long, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, a + 8u), __builtin_bit_cast(unsigned long long, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if constexpr (sizeof(V) == 8 && WB == 12) { /* (two dwords of a slot that starts on any dword: no 64-bit atomic there) */
    __hip_atomic_fetch_xor(NRQ_LDSP(unsigned int, a), v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_xor(NRQ_LDSP(unsigned int, a + 4u), v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if constexpr (sizeof(V) == 8) {
    __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, a), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if constexpr (WB >= 4) {
    __hip_atomic_fetch_xor(NRQ_LDSP(unsigned int, a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else { /* 2-byte slots: the halfword inside its dword.  (The shift count is a << 3 as it stands: a is even, so its low five bits --
            * all the shifter looks at -- are 16 for the upper halfword and 0 for the lower: one instruction, not an AND and a
            * multiply; tools/microbench/narrow_rows.hip: 57.4 -> clocks per row in profiles/r6_microbench_rows.txt) */
    __hip_atomic_fetch_xor(NRQ_LDSP(unsigned int, a & ~3u), v << ((a << 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
template <int WB> __device__ __forceinline__ void ph_row_apply(const StripCtx<WB> &, uint32_t op, typename RowVal<WB>::type v) {
  row_apply_at<WB, typename RowVal<WB>::type>(row_addr_lo<WB>(op), v);
}
template <int WB> __device__ __forceinline__ typename RowVal<WB>::type row_zero() { typename RowVal<WB>::type z = {}; return z; }
/* The row pipeline itself, run by ONE wave (lane = 0..63) on the LDS image that starts at LDS address 0: step q
 * applies row q-NRQ_PIPE and then reads the sources of row q.  Op words live in a ring of U fixed registers, a quad of
 * four rows per 16-byte load (plan.h: the stream is quad-interleaved), each quad reloaded (U rows ahead) right after its
 * last row has been applied -- no hand-over between registers, so the loads stay outstanding across the LDS work.  The
 * ring is primed with the stream's first U-4 rows; its last quad starts as padding (what the steady state has there at
 * the top of a trip: rows already read, about to be applied and replaced).  The stream is padded (NRQ_PAD_ROWS) so that
 * every fetch is in bounds.  Used by the solve kernel on symbol strips and by the planner kernels on strips of the W bit rows. */
typedef uint32_t OpQuad __attribute__((ext_vector_type(4)));
template <int WB, int OFF, class V, uint32_t U> __device__ __forceinline__ void fwd_rows_impl(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  constexpr uint32_t P = NRQ_PIPE, NS = NRQ_PIPE + 1u, NQ = U / 4u;
  static_assert(U % NS == 0 && U % 4u == 0 && P <= 4u && U <= NRQ_RING_MAX, "ring: whole quads, whole value sets");
  const NRQ_GAS OpQuad *nxt = reinterpret_cast<const NRQ_GAS OpQuad *>(ops) + lane; /* quad g of this lane: nxt[g * NRQ_ROW] */
  uint32_t o[U];
  V v[NS];
#pragma unroll
  for (uint32_t g = 0; g + 1u < NQ; g++) {
    const OpQuad w = nxt[g * NRQ_ROW];
    o[4u * g] = w.x; o[4u * g + 1u] = w.y; o[4u * g + 2u] = w.z; o[4u * g + 3u] = w.w;
  }
#pragma unroll
  for (uint32_t k = U - 4u; k < U; k++) o[k] = NRQ_NOP_AT(lane);
#pragma unroll
  for (uint32_t k = 0; k < NS; k++) { V z = {}; v[k] = z; }
  for (uint32_t base = 0; base < nrows + P; base += U, nxt += NQ * NRQ_ROW) {
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t j = (k + U - P) % U; /* ring slot of row q - P */
      /* both LDS addresses of the step are formed BEFORE the step waits for the LDS data it applies: a lone wave issues
       * in order, and an address instruction between the wait and the LDS instruction that needs it puts its latency on
       * the critical path of every row (tools/microbench/fwd_loop.hip: 51 -> 44 clocks per row) */
      const uint32_t a_dst = row_addr_lo<WB>(o[j]) + (uint32_t)OFF, a_src = row_addr_hi<WB>(o[k]) + (uint32_t)OFF;
      NRQ_SCHED_FENCE();
      row_apply_at<WB, V>(a_dst, v[(k + NS - P) % NS]);
      if constexpr (sizeof(V) == 4 && WB == 2 && OFF == 0) v[k % NS] = *NRQ_LDSP(uint16_t, a_src);
      else if constexpr (sizeof(V) == 8 && WB == 12) { typedef V Va __attribute__((aligned(4))); v[k % NS] = *NRQ_LDSP(Va, a_src); } /* (ds_read2_b32) */
      else v[k % NS] = *NRQ_LDSP(V, a_src);
#ifndef NRQ_EXPERIMENT_NO_OPLOAD /* (measurement only: the loop on the first ring's words for ever -- no vector memory load in it) */
      if (j % 4u == 3u) { /* the quad's last row has been applied: the rows it holds next (this trip's if still ahead, k < P) */
        const OpQuad w = nxt[(j / 4u + (j < k ? NQ : 0u)) * NRQ_ROW];
        o[j - 3u] = w.x; o[j - 2u] = w.y; o[j - 1u] = w.z; o[j] = w.w;
      }
#endif
      NRQ_SCHED_FENCE();
    }
  }
}
template <int WB, uint32_t U = NRQ_RING> __device__ __forceinline__ void fwd_rows(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  fwd_rows_impl<WB, 0, typename RowVal<WB>::type, U>(ops, nrows, lane);
}
/* The same pipeline on HALF of the strip width (bytes [OFF, OFF + WB/2) of every slot): the solve kernel runs two of
 * them on two waves -- byte columns are independent, so the two never need to meet -- because a row costs a single
 * wave ~45 clocks of instruction issue and LDS round trip against ~25 of LDS time. */
template <int WB> struct HalfVal;
template <> struct HalfVal<16> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct HalfVal<8> { typedef uint32_t type; };
template <int WB, int OFF, uint32_t U = NRQ_RING> __device__ __forceinline__ void fwd_rows_half(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  static_assert(WB == 16 || WB == 8, "half-width pipeline: 16- and 8-byte strips only");
  fwd_rows_impl<WB, OFF, typename HalfVal<WB>::type, U>(ops, nrows, lane);
}
/* ... and on a THIRD of the 12-byte strip (bytes [OFF, OFF + 4) of every slot): three waves, a dword each.  A 12-byte slot starts on
 * any dword, so the 64 lanes' dwords spread over all 64 LDS banks (the 8-byte halves of 16-byte slots reach 32 of them). */
template <int OFF, uint32_t U = NRQ_RING> __device__ __forceinline__ void fwd_rows_third(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  fwd_rows_impl<12, OFF, uint32_t, U>(ops, nrows, lane);
}
template <uint32_t U = NRQ_RING> __device__ __forceinline__ void fwd_rows_two_thirds(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  fwd_rows_impl<12, 0, u2, U>(ops, nrows, lane);
}
/* The op stream TRANSPOSED and in reverse: row by row from the last to the first, every op dst ^= src becomes
 * slot(src) ^= slot(dst).  With z = one 16-byte value per slot this computes z <- z * X^-1 for the row vector z (the forward
 * passes compute X^-1 * D for the columns of D): the planner folds the pivot columns out of the HDPC rows with it (planner_body.h
 * pl_mhrev_*).  The stream's hazard rule carries over: a row's ops read the targets of the forward ops and write their sources,
 * never a slot the row itself reads, and whatever wrote a slot it reads lies NRQ_PIPE or more rows later in the stream (the
 * forward rule applied to the later op).  Same depth-2 pipeline, one wave, no barriers; op words 16 rows (four quads) at a time,
 * the next 16 requested while these run.  The stream's padding rows behind nrows are all NOPs, so the first chunk may start in them. */
template <int WB> __device__ __forceinline__ void rev_rows(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  typedef typename RowVal<WB>::type V;
  constexpr uint32_t P = NRQ_PIPE, CH = 16u;
  static_assert(WB == 16 || WB == 8 || WB == 4, "transposed pass: 16-, 8- and 4-byte slots");
  static_assert(CH % P == 0u && P <= 4u, "chunk: whole turns of the pending ring");
  if (nrows == 0u) return;
  const NRQ_GAS OpQuad *q4 = reinterpret_cast<const NRQ_GAS OpQuad *>(ops) + lane; /* quad g of this lane: q4[g * NRQ_ROW] */
  const uint32_t top = (nrows + CH - 1u) / CH * CH; /* rows [nrows, top) are padding */
  uint32_t o[CH], on[CH];
  auto fetch = [&](uint32_t base, uint32_t (&dst)[CH]) {
#pragma unroll
    for (uint32_t g = 0; g < CH / 4u; g++) {
      const OpQuad w = q4[(size_t)(base / 4u + g) * NRQ_ROW];
      dst[4u * g] = w.x; dst[4u * g + 1u] = w.y; dst[4u * g + 2u] = w.z; dst[4u * g + 3u] = w.w;
    }
  };
  V pv[P];
  uint32_t pa[P]; /* pending: values read (slot(dst)) and the LDS address they go to (slot(src)), oldest at index row % P */
#pragma unroll
  for (uint32_t k = 0; k < P; k++) { V z = {}; pv[k] = z; pa[k] = row_addr_hi<WB>(NRQ_NOP_AT(lane)); }
  fetch(top - CH, o);
  for (uint32_t base = top - CH;; base -= CH) {
    if (base >= CH) fetch(base - CH, on);
#pragma unroll
    for (uint32_t kk = 0; kk < CH; kk++) {
      const uint32_t k = CH - 1u - kk; /* row base + k, descending */
      const uint32_t a_rd = row_addr_lo<WB>(o[k]), a_wr = row_addr_hi<WB>(o[k]);
      NRQ_SCHED_FENCE();
      row_apply_at<WB, V>(pa[k % P], pv[k % P]); /* the row read P steps ago */
      pv[k % P] = *NRQ_LDSP(V, a_rd);
      pa[k % P] = a_wr;
      NRQ_SCHED_FENCE();
    }
    if (base < CH) break;
#pragma unroll
    for (uint32_t k = 0; k < CH; k++) o[k] = on[k];
  }
#pragma unroll
  for (uint32_t kk = 0; kk < P; kk++) { const uint32_t k = P - 1u - kk; row_apply_at<WB, V>(pa[k % P], pv[k % P]); } /* (oldest first: rows P-1 ... 0) */
}
/* The forward passes on a WIDE strip (G lanes per op, 16 bytes each; lane = op * G + sub): a row of 64 op slots is G
 * wave instructions of 64 / G ops.  Ops of one row never read what the row writes (plan.h), so a row is: fetch the G
 * source pieces, then the G XORs; instructions whose 64 / G ops are all padding are skipped (the planners fill a row from
 * the front, so a thin level costs one or two instructions, not a row).  For small
 * blocks: their levels hold a dozen ops, which leaves most of a 64-op row's lanes empty on a 16-byte strip. */
template <int G> __device__ __forceinline__ void fwd_rows_wide(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  constexpr uint32_t NV = NRQ_ROW / G, SH = G == 8 ? 7u : G == 4 ? 6u : G == 2 ? 5u : 4u; /* log2(16 * G) */
  constexpr uint32_t R = 32u; /* op words held in registers: instructions (of NV ops) fetched that far ahead -- like
                               * fwd_rows' ring: every register is reloaded right after its instruction has run, so the loads
                               * stay outstanding across the LDS work (a load per row that is waited for in the next row is a trip
                               * to L2 per row: measured 390 clocks per row) */
  static_assert(R % G == 0, "whole rows in the ring");
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const uint32_t vl = lane / G, subofs = (lane % G) * 16u;
  /* instruction s = sub-instruction s % G of row s / G: the ops of lanes [(s % G) * NV, +NV) of the (quad-interleaved) stream */
  auto word = [&](uint32_t sidx) { return ops[NRQ_OP_INDEX(sidx / G, (sidx % G) * NV + vl)]; };
  const uint32_t s0 = 0u, S = nrows * G;
  uint32_t o[R];
#pragma unroll
  for (uint32_t k = 0; k < R; k++) o[k] = word(s0 + k);
  /* software pipeline, one row deep: the sources of row r + 1 are fetched before row r is applied (rows that follow
   * each other never depend on each other, plan.h), so an LDS round trip is not exposed per row */
  bool live[2][G];
  u4 v[2][G];
#pragma unroll
  for (uint32_t s = 0; s < (uint32_t)G; s++) { live[1][s] = false; v[1][s] = u4{0u, 0u, 0u, 0u}; }
  uint32_t oa[G]; /* op words of the row being applied (its ring entries are reloaded meanwhile) */
#pragma unroll
  for (uint32_t s = 0; s < (uint32_t)G; s++) oa[s] = 0u;
  for (uint32_t base = s0; base < S + G; base += R) { /* (runs up to R - 1 instructions into the padding rows: all skipped) */
#pragma unroll
    for (uint32_t g0 = 0; g0 < R; g0 += G) { /* read row (base + g0) / G, apply the row before it */
      constexpr uint32_t dummy = 0;
      (void)dummy;
      const uint32_t cur = (g0 / G) & 1u, prv = cur ^ 1u;
#pragma unroll
      for (uint32_t s = 0; s < (uint32_t)G; s++) {
        live[cur][s] = __ballot(!NRQ_OP_IS_NOP(o[g0 + s])) != 0ull;
        if (live[cur][s]) v[cur][s] = *NRQ_LDSP(u4, ((o[g0 + s] >> 16) << SH) + subofs);
      }
#pragma unroll
      for (uint32_t s = 0; s < (uint32_t)G; s++) {
        if (live[prv][s]) {
          const uint32_t a = ((oa[s] & 0xFFFFu) << SH) + subofs;
          const u2 lo = {v[prv][s].x, v[prv][s].y}, hi = {v[prv][s].z, v[prv][s].w};
          __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, a), __builtin_bit_cast(unsigned long long, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, a + 8u), __builtin_bit_cast(unsigned long long, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        oa[s] = o[g0 + s];
        o[g0 + s] = word(base + g0 + s + R); /* (the stream is padded by NRQ_PAD_ROWS: in bounds) */
      }
    }
  }
}
#else
template <int G> SB_HD void fwd_rows_wide(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <int WB> struct RowVal { typedef SV<WB> type; };
template <int WB> SB_HD SV<WB> ph_row_read(const StripCtx<WB> &c, uint32_t op) { return lds_get<WB>(c.lds, op >> 16); }
template <int WB> SB_HD void ph_row_apply(const StripCtx<WB> &c, uint32_t op, const SV<WB> &v) {
  lds_xor<WB>(c.lds, op & 0xFFFFu, v);
}
template <int WB> SB_HD SV<WB> row_zero() { return sv_zero<WB>(); }
/* host compilation pass of the kernels only: the row pipeline exists on the device (the CPU emulators have
 * their own row loops over ph_row_read / ph_row_apply) */
template <int WB, uint32_t U = NRQ_RING> SB_HD void fwd_rows(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <int WB, int OFF, uint32_t U = NRQ_RING> SB_HD void fwd_rows_half(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <int OFF, uint32_t U = NRQ_RING> SB_HD void fwd_rows_third(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <uint32_t U = NRQ_RING> SB_HD void fwd_rows_two_thirds(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <int WB> SB_HD void rev_rows(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
#endif

/* phase 3: HDPC right-hand sides R_h = SUM_c HDPC[h][c] * Y(c) over the peeled columns, through
 * HDPC = MT*GAMMA (RFC 6330 section 5.3.3.3): thread t owns columns [a,b); g follows the GAMMA
 * recurrence g = alpha*g + Y(c); the two unit entries of MT's column c add g to two of the H
 * accumulators; whatever the chunk owes to the columns beyond b is G[.][b] * alpha*g_end.  The
 * accumulators are private per lane (nsets copies of the H sums in region X, free until the dense
 * stage): all threads XORing into the same H slots made this phase one long LDS atomic conflict.
 * ph_hdpc_reduce folds the copies into the HDPC slots and zeroes them again. */
template <int WB, int G = 1> SB_HD uint32_t hdpc_nsets(const StripCtx<WB, G> &c) {
  const uint32_t cap = (c.lay.total - c.lay.off_x) / (c.h->H * WB * G);
  uint32_t n = 1;
  while (n * 2u <= cap && n < 64u) n *= 2u;
  return n;
}
template <int WB, int G = 1> SB_HD void ph_hdpc_reduce(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const uint32_t H = c.h->H, nsets = hdpc_nsets<WB, G>(c);
  const uint32_t hq = tid & 15u, part = tid >> 4, nparts = nt >> 4;
  if (hq >= H) return;
  SV<WB> acc = sv_zero<WB>();
  bool any = false;
  for (uint32_t s = part; s < nsets; s += nparts) {
    sv_xor<WB>(acc, lds_get<WB, G>(c.cf(), s * H + hq));
    lds_put<WB, G>(c.cf(), s * H + hq, sv_zero<WB>());
    any = true;
  }
  if (any) lds_xor<WB, G>(c.slots(), c.h->S + hq, acc);
}
#ifndef NRQ_HDPC_COEF_ALL_NT
#define NRQ_HDPC_COEF_ALL_NT 256u /* from this workgroup size on the H coefficients of the closing fold are loaded together (the 256-thread
                                     * variant built for four per CU has the registers: K=1000 1062 -> 1102 Gbit/s, K=2000 1210 -> 1242;
                                     * the single-wave variant gains nothing) */
#endif
/* all ones if bit h of x is set (one v_bfe_i32) */
SB_HD uint32_t nrq_bit_mask(uint32_t x, uint32_t h) { return (uint32_t)((int32_t)(x << (31u - h)) >> 31); }
/* REGS form of a thread's chunk [a, b) (whole groups of 8 columns): the H sums live in REGISTERS while the thread walks its
 * columns -- a column adds g to two of them, chosen by a two-bit mask over the H rows (one bit-field extract and one
 * a ^ (b & m) per row and dword) -- and reach the thread's LDS copy once, at the end.  With LDS atomics per column (four
 * ds_xor_b64 on 16-byte strips, their 64 lanes on 16 bank pairs) the recurrence was bound by the LDS pipeline: 17.5 k clocks for
 * the 24 columns of the longest thread at K=8192, and the same count of atomics per column whatever the strip width (72 k
 * clocks on 4-byte strips at K=27000, 211 k on 2-byte strips at K'=56403).  Needs 4 * H registers on 16-byte strips: the big
 * workgroup's variant only. */
#define NRQ_MIN_H 10u /* RFC 6330 table 2: 10 <= H <= 16 */
template <int WB, int G> SB_HD void hdpc_chunk_regs(const StripCtx<WB, G> &c, uint32_t a, uint32_t b, uint32_t n, uint32_t H, uint32_t mine,
                                                    uint32_t zero_at, const NRQ_GAS uint16_t *pivof, const NRQ_GAS uint8_t *b12,
                                                    const NRQ_GAS uint8_t *Gm) {
  constexpr int ND = SV<WB>::ND;
  SV<WB> acc[16]; /* (H <= 16) */
#pragma unroll
  for (uint32_t h = 0; h < 16; h++) acc[h] = sv_zero<WB>();
  SV<WB> g = sv_zero<WB>();
  /* (the slot numbers and MT rows of the NEXT group are asked for before this group's strips: a trip to L2 per group of eight
   * columns otherwise stands in front of ~350 instructions -- 14 groups per thread at K'=56403) */
  uint4 sl_n = *reinterpret_cast<const NRQ_GAS uint4 *>(pivof + a);
  uint2 bb_n = *reinterpret_cast<const NRQ_GAS uint2 *>(b12 + a);
  for (uint32_t c0 = a; c0 < b; c0 += 8) {
    const uint4 sl = sl_n;
    const uint2 bb = bb_n;
    if (c0 + 8u < b) {
      sl_n = *reinterpret_cast<const NRQ_GAS uint4 *>(pivof + c0 + 8u);
      bb_n = *reinterpret_cast<const NRQ_GAS uint2 *>(b12 + c0 + 8u);
    }
    const uint32_t slw[4] = {sl.x, sl.y, sl.z, sl.w};
    const uint32_t bbw[2] = {bb.x, bb.y};
    /* the eight strips first, unconditionally (a column without a slot, or beyond the chunk, reads the lane's zero scratch
     * slot): nothing between them that the LDS would have to keep in order */
    SV<WB> y[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
      const uint32_t sq = (slw[q >> 1] >> ((q & 1u) * 16u)) & 0xFFFFu;
      y[q] = lds_get<WB, G>(c.lds, (c0 + q < b && sq != NRQ_NOSLOT) ? sq + NRQ_SCRATCH : zero_at);
    }
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
      const uint32_t col = c0 + q;
      if (col >= b) break;
      g = sv_xtime<WB>(g);
      sv_xor<WB>(g, y[q]);
      if (col + 1 < n) {
        const uint32_t e = (bbw[q >> 2] >> ((q & 3u) * 8u)) & 0xFFu;
        const uint32_t bits = (1u << (e & 15u)) ^ (1u << (e >> 4));
#pragma unroll
        for (uint32_t h = 0; h < 16; h++) {
          if (h < NRQ_MIN_H || h < H) { /* (no `break`: the loop must unroll completely, or acc[] is an array in scratch memory) */
            const uint32_t m = nrq_bit_mask(bits, h);
#pragma unroll
            for (int i = 0; i < ND; i++) acc[h].w[i] = nrq_xor_and(acc[h].w[i], g.w[i], m);
          }
        }
      } else { /* last column of MT is alpha^h */
        SV<WB> v = g;
#pragma unroll
        for (uint32_t h = 0; h < 16; h++) {
          if (h < H) {
            sv_xor<WB>(acc[h], v);
            v = sv_xtime<WB>(v);
          }
        }
      }
    }
  }
  NRQ_MARK(c, 0);
  if (b < n) { /* what the chunk owes to the columns beyond b: G[.][b] * alpha * g */
    SV<WB> pw[8];
    pw[0] = sv_xtime<WB>(g);
#pragma unroll
    for (int k = 1; k < 8; k++) pw[k] = sv_xtime<WB>(pw[k - 1]);
    uint32_t coef[16];
#pragma unroll
    for (uint32_t h = 0; h < 16; h++) coef[h] = h < H ? Gm[(size_t)h * n + b] : 0u;
#pragma unroll
    for (uint32_t h = 0; h < 16; h++) {
      if (h < NRQ_MIN_H || h < H) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint32_t m = nrq_bit_mask(coef[h], (uint32_t)k);
#pragma unroll
          for (int i = 0; i < ND; i++) acc[h].w[i] = nrq_xor_and(acc[h].w[i], pw[k].w[i], m);
        }
      }
    }
  }
#pragma unroll
  for (uint32_t h = 0; h < 16; h++) {
    if (h < H) lds_xor<WB, G>(c.cf(), mine + h, acc[h]);
  }
}
template <int WB, int G = 1, bool REGS = false> SB_HD void ph_hdpc(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const nrq_kconst_hdr *kh = reinterpret_cast<const nrq_kconst_hdr *>(c.kc);
  const NRQ_GAS uint8_t *Gm = gptr<uint8_t>(c.kc + kh->off_g); /* the HDPC block */
  const NRQ_GAS uint8_t *b12 = gptr<uint8_t>(c.kc + kh->off_b12);
  const NRQ_GAS uint16_t *pivof = c.template arr<uint16_t>(c.h->off_pivof);
  const uint32_t n = kh->n, H = c.h->H;
  const uint32_t mine = (tid & (hdpc_nsets<WB, G>(c) - 1u)) * H; /* this lane's copy of the H accumulators */
  /* chunks of whole 8-column groups, as equal as possible; the threads that take one group more are the FIRST
   * ones, so that a single wave (not one lane of every wave) runs the longer loop */
  /* (small blocks: groups of 4 or 2 columns while there are fewer groups than threads -- the closing fold below costs a wave the
   * same instructions whether its lanes close 8 columns or 2, the recurrence costs it per column of its longest lane: K=100 on
   * a single wave is 15 lanes x 8 columns as groups of 8, 59 lanes x 2 as groups of 2) */
  const uint32_t gsz = (n + 1u) / 2u <= nt ? 2u : (n + 3u) / 4u <= nt ? 4u : 8u; /* (smaller than 8: at most one group per thread) */
  const uint32_t groups = (n + gsz - 1u) / gsz, base = groups / nt, extra = groups - base * nt;
  const uint32_t a = gsz * (tid * base + (tid < extra ? tid : extra));
  if (a >= n || (base == 0u && tid >= extra)) return;
  const uint32_t len = gsz * (base + (tid < extra ? 1u : 0u));
  const uint32_t b = (a + len < n) ? a + len : n;
  if constexpr (REGS && G == 1) {
    if (gsz == 8u) {
      hdpc_chunk_regs<WB, G>(c, a, b, n, H, mine, tid & (NRQ_SCRATCH - 1u), pivof, b12, Gm);
      return;
    }
  }
  SV<WB> g = sv_zero<WB>();
  if (gsz < 8u) { /* one group of at most 4 columns per thread (groups < nt): slot numbers and MT rows column by column, in flight together */
    uint32_t sl4[4], e4[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
      const uint32_t col = a + q < b ? a + q : a;
      sl4[q] = pivof[col];
      e4[q] = b12[col];
    }
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
      const uint32_t col = a + q;
      if (col >= b) break;
      g = sv_xtime<WB>(g);
      if (sl4[q] != NRQ_NOSLOT) {
        SV<WB> y = lds_get<WB, G>(c.slots(), sl4[q]);
        sv_xor<WB>(g, y);
      }
      if (col + 1 < n) {
        lds_xor<WB, G>(c.cf(), mine + (e4[q] & 15u), g);
        lds_xor<WB, G>(c.cf(), mine + (e4[q] >> 4), g);
      } else { /* last column of MT is alpha^h */
        SV<WB> v = g;
        for (uint32_t h = 0; h < H; h++) {
          lds_xor<WB, G>(c.cf(), mine + h, v);
          v = sv_xtime<WB>(v);
        }
      }
    }
  } else
  /* 8 columns per step: their slot indices and MT rows come in two vector loads (the arrays are
   * 16-byte aligned and padded, a is a multiple of 8) */
  for (uint32_t c0 = a; c0 < b; c0 += 8) {
    const uint4 sl = *reinterpret_cast<const NRQ_GAS uint4 *>(pivof + c0);
    const uint2 bb = *reinterpret_cast<const NRQ_GAS uint2 *>(b12 + c0);
    const uint32_t slw[4] = {sl.x, sl.y, sl.z, sl.w};
    const uint32_t bbw[2] = {bb.x, bb.y};
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
      const uint32_t col = c0 + q;
      if (col >= b) break;
      const uint32_t s = (slw[q >> 1] >> ((q & 1u) * 16u)) & 0xFFFFu;
      g = sv_xtime<WB>(g);
      if (s != NRQ_NOSLOT) {
        SV<WB> y = lds_get<WB, G>(c.slots(), s);
        sv_xor<WB>(g, y);
      }
      if (col + 1 < n) {
        const uint32_t e = (bbw[q >> 2] >> ((q & 3u) * 8u)) & 0xFFu;
        lds_xor<WB, G>(c.cf(), mine + (e & 15u), g);
        lds_xor<WB, G>(c.cf(), mine + (e >> 4), g);
      } else { /* last column of MT is alpha^h */
        SV<WB> v = g;
        for (uint32_t h = 0; h < H; h++) {
          lds_xor<WB, G>(c.cf(), mine + h, v);
          v = sv_xtime<WB>(v);
        }
      }
    }
  }
  NRQ_MARK(c, 0);
  if (b < n) {
    /* H products of the same vector: its 8 multiples by alpha^k once, then one masked XOR per set coefficient bit */
    SV<WB> pw[8];
    pw[0] = sv_xtime<WB>(g);
#pragma unroll
    for (int k = 1; k < 8; k++) pw[k] = sv_xtime<WB>(pw[k - 1]);
    if (nt >= NRQ_HDPC_COEF_ALL_NT) { /* the H coefficients loaded together: one trip to L2 instead of one per HDPC row (the compiler may not
                                       * move the next load above the LDS atomic of the term before) */
      uint32_t coef[16]; /* (H <= 16) */
#pragma unroll
      for (uint32_t h = 0; h < 16; h++) coef[h] = h < H ? Gm[(size_t)h * n + b] : 0u;
#pragma unroll
      for (uint32_t h = 0; h < 16; h++) {
        if (h >= H) break;
        SV<WB> t = sv_zero<WB>();
#pragma unroll
        for (int k = 0; k < 8; k++) sv_xor_masked<WB>(t, pw[k], 0u - ((coef[h] >> k) & 1u));
        lds_xor<WB, G>(c.cf(), mine + h, t);
      }
    } else {
      for (uint32_t h = 0; h < H; h++) {
        const uint32_t coef = Gm[(size_t)h * n + b];
        SV<WB> t = sv_zero<WB>();
#pragma unroll
        for (int k = 0; k < 8; k++) sv_xor_masked<WB>(t, pw[k], 0u - ((coef >> k) & 1u));
        lds_xor<WB, G>(c.cf(), mine + h, t);
      }
    }
  }
}

/* phase 4a: the GF(2) combinations of the dense stage, E_p (slot M+p, zeroed by ph_clear) = XOR of the leftover rows that
 * the bit matrix of the plan names (plan.h off_augt).  Method of four Russians, like the back-substitution: region X --
 * idle and zero between ph_hdpc_reduce and the dense fold -- takes 16-entry XOR tables over groups of four leftover rows
 * (words [w0, w0 + nw) of the bit rows at a time, as many as region X holds), then every reduced row looks its nibbles up.
 * Threads share a row word by word and add their parts with LDS atomics.  ph_clear_x hands region X back zeroed.
 * (As ops of the stream these terms -- r2 * nlow / 2, a quarter of all row operations at K=8192 -- cost the forward wave
 * 300 of its 1400 rows: 16 k clocks per strip against 3 k here.) */
template <int WB, int G = 1> SB_HD uint32_t low_table_words(const StripCtx<WB, G> &c) { /* words of the bit rows region X has tables for */
  const uint32_t groups = (c.lay.total - c.lay.off_x) / (16u * WB * G);
  return groups / 8u;
}
template <int WB, int G = 1> SB_HD void ph_low_tables(const StripCtx<WB, G> &c, uint32_t w0, uint32_t tid, uint32_t nt) {
  const NRQ_GAS uint16_t *lowslot = c.template arr<uint16_t>(c.h->off_lowslot);
  const uint32_t nlow = c.h->nlow, lpr = c.h->lpr, cap = low_table_words<WB, G>(c);
  const uint32_t nw = lpr - w0 < cap ? lpr - w0 : cap;
  for (uint32_t e = tid; e < nw * 8u * 16u; e += nt) {
    const uint32_t grp = e >> 4, nib = e & 15u, j0 = w0 * 32u + grp * 4u;
    uint32_t sl[4];
#pragma unroll
    for (uint32_t bq = 0; bq < 4; bq++) sl[bq] = (((nib >> bq) & 1u) && j0 + bq < nlow) ? (uint32_t)lowslot[j0 + bq] : NRQ_NOSLOT;
    SV<WB> v = sv_zero<WB>();
#pragma unroll
    for (uint32_t bq = 0; bq < 4; bq++)
      if (sl[bq] != NRQ_NOSLOT) sv_xor<WB>(v, lds_get<WB, G>(c.slots(), sl[bq]));
    lds_put<WB, G>(c.t4(), e, v);
  }
}
/* (the words of the bit rows a thread will use are fetched BEFORE the barrier that ends the table build -- ph_combine_fetch --
 * so that the trip to L2 runs beside it; NRQ_COMBINE_WU words per thread at most, the rest is fetched in ph_combine itself) */
#define NRQ_COMBINE_WU 4u
template <int WB, int G = 1> SB_HD void ph_combine_fetch(const StripCtx<WB, G> &c, uint32_t w0, uint32_t tid, uint32_t nt, uint32_t (&bits)[NRQ_COMBINE_WU]) {
  const NRQ_GAS uint32_t *augt = c.template arr<uint32_t>(c.h->off_augt);
  const uint32_t r2 = c.h->r2, lpr = c.h->lpr, stride = c.h->aug_stride, cap = low_table_words<WB, G>(c);
  const uint32_t nw = lpr - w0 < cap ? lpr - w0 : cap;
#pragma unroll
  for (uint32_t k = 0; k < NRQ_COMBINE_WU; k++) bits[k] = 0u;
  if (!r2) return;
  uint32_t nparts = nt / r2;
  if (nparts < 1u) nparts = 1u;
  if (nparts > nw) nparts = nw;
  if (tid >= r2 * nparts) return;
  const uint32_t p = tid % r2, part = tid / r2;
#pragma unroll
  for (uint32_t k = 0; k < NRQ_COMBINE_WU; k++) bits[k] = part + k * nparts < nw ? augt[(size_t)(w0 + part + k * nparts) * stride + p] : 0u;
}
template <int WB, int G = 1> SB_HD void ph_combine(const StripCtx<WB, G> &c, uint32_t w0, uint32_t tid, uint32_t nt, const uint32_t (&first)[NRQ_COMBINE_WU]) {
  const NRQ_GAS uint32_t *augt = c.template arr<uint32_t>(c.h->off_augt);
  const uint32_t r2 = c.h->r2, lpr = c.h->lpr, stride = c.h->aug_stride, cap = low_table_words<WB, G>(c);
  const uint32_t nw = lpr - w0 < cap ? lpr - w0 : cap;
  if (!r2) return;
  uint32_t nparts = nt / r2; /* threads per reduced row: they take its words in turn */
  if (nparts < 1u) nparts = 1u;
  if (nparts > nw) nparts = nw;
  for (uint32_t i = tid; i < r2 * nparts; i += nt) {
    const uint32_t p = i % r2, part = i / r2;
    constexpr uint32_t WU = NRQ_COMBINE_WU; /* words of the row in flight */
    SV<WB> acc = sv_zero<WB>();
    for (uint32_t wa = part; wa < nw; wa += WU * nparts) {
      uint32_t bits[WU];
#pragma unroll
      for (uint32_t k = 0; k < WU; k++)
        bits[k] = (i == tid && wa == part) ? first[k] : (wa + k * nparts < nw ? augt[(size_t)(w0 + wa + k * nparts) * stride + p] : 0u);
#pragma unroll
      for (uint32_t k = 0; k < WU; k++) {
        if (!bits[k]) continue;
        const uint32_t w = wa + k * nparts;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) sv_xor<WB>(acc, lds_get<WB, G>(c.t4(), (w * 8u + q) * 16u + ((bits[k] >> (4u * q)) & 15u)));
      }
    }
    lds_xor<WB, G>(c.slots(), c.h->M + p, acc);
  }
}
template <int WB, int G = 1> SB_HD void ph_clear_x(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const uint32_t nx = (c.lay.total - c.lay.off_x) / (WB * G);
  for (uint32_t f = tid; f < nx; f += nt) lds_put<WB, G>(c.cf(), f, sv_zero<WB>());
}

/* phase 4b: fold the columns resolved by binary rows out of the HDPC rows: R_h ^= mh[h][p]*E_p.
 * Big workgroups (dense_fold_shared): one thread per E_p -- its 8 multiples by alpha^k once, then every one of the H
 * products is a masked XOR per set coefficient bit (a general GF(256) multiply per (h, p) pair is 3.4x the
 * instructions).  The products go to the private accumulator copies of region X like those of ph_hdpc, and
 * ph_hdpc_reduce has to run once more to fold them into the HDPC slots.  Small workgroups (small blocks, few E_p):
 * one general multiply per (h, p) pair, spread over all threads, straight into the slots -- the extra phase costs
 * more than the instructions saved (measured: 6 % at K <= 1024). */
#ifndef NRQ_DENSE_SHARED_MIN_NT
#define NRQ_DENSE_SHARED_MIN_NT 512u
#endif
/* (a single-wave workgroup, nt == 64, also shares the multiples: with one wave the few (h, p) pairs of a small block
 * leave most lanes idle either way, and the general multiply is 3.4x the instructions of the shared form) */
SB_HD bool dense_fold_shared(uint32_t nt) { return nt >= NRQ_DENSE_SHARED_MIN_NT || nt == 64u; }
/* BATCH (the single-wave variant): the coefficients of a thread's terms are asked for eight at a time before the first
 * is used -- one after the other each was a trip to L2 in front of ~230 instructions (K=1000: 5 terms per thread, fold 12 k clocks
 * of a 110 k strip; K=100: free columns 14 k of 140 k).  K=100: dense stage 45 k -> 38 k clocks, solve 7.58 / 7.16 -> 7.24 / 6.84 ms; the 256-thread
 * variant LOSES 2-4 % with it (K=500, 1000, 2000: its fold is bound by the multiplications of four waves per SIMD, and the unrolled
 * form costs it 12 more spills), so it keeps the loop. */
template <int WB, int G = 1, int BATCH = 0> SB_HD void ph_dense_fold(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const NRQ_GAS uint8_t *mh = c.template arr<uint8_t>(c.h->off_mh);
  const uint32_t H = c.h->H, r2 = c.h->r2, M = c.h->M;
  if (!dense_fold_shared(nt) && G == 1) {
    const uint32_t hq = tid & 15u, part = tid >> 4, nparts = nt >> 4;
    if (hq >= H) return;
    SV<WB> acc = sv_zero<WB>();
    if constexpr (BATCH > 0) { /* (BATCH coefficients of the thread's terms in flight together: 8 on a single wave, see above) */
      for (uint32_t p0 = part; p0 < r2; p0 += (uint32_t)BATCH * nparts) {
        uint32_t coef[BATCH];
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)BATCH; i++) coef[i] = mh[(size_t)hq * r2 + (p0 + i * nparts < r2 ? p0 + i * nparts : p0)];
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)BATCH; i++) {
          const uint32_t p = p0 + i * nparts;
          if (p >= r2 || !coef[i]) continue;
          SV<WB> t = sv_mul<WB>(lds_get<WB, G>(c.slots(), M + p), coef[i]);
          sv_xor<WB>(acc, t);
        }
      }
    } else {
    for (uint32_t p = part; p < r2; p += nparts) {
      uint32_t coef = mh[(size_t)hq * r2 + p];
      if (!coef) continue;
      SV<WB> t = sv_mul<WB>(lds_get<WB, G>(c.slots(), M + p), coef);
      sv_xor<WB>(acc, t);
    }
    }
    lds_xor<WB, G>(c.slots(), c.h->S + hq, acc);
    return;
  }
  const uint32_t mine = (tid & (hdpc_nsets<WB, G>(c) - 1u)) * H;
  for (uint32_t p = tid; p < r2; p += nt) {
    SV<WB> pw[8];
    pw[0] = lds_get<WB, G>(c.slots(), M + p);
#pragma unroll
    for (int k = 1; k < 8; k++) pw[k] = sv_xtime<WB>(pw[k - 1]);
    uint32_t coef[16]; /* (H <= 16) all of them in flight together: one trip to L2, not one per HDPC row */
#pragma unroll
    for (uint32_t h = 0; h < 16; h++) coef[h] = h < H ? mh[(size_t)h * r2 + p] : 0u;
#pragma unroll
    for (uint32_t h = 0; h < 16; h++) {
      if (!coef[h]) continue;
      SV<WB> t = sv_zero<WB>();
#pragma unroll
      for (int k = 0; k < 8; k++) sv_xor_masked<WB>(t, pw[k], 0u - ((coef[h] >> k) & 1u));
      lds_xor<WB, G>(c.cf(), mine + h, t);
    }
  }
}

/* phase 4c: free columns C_f = SUM_h hinv[f][h] * R_h */
template <int WB, int G = 1, bool BATCH = false> SB_HD void ph_dense_free(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const NRQ_GAS uint8_t *hinv = c.template arr<uint8_t>(c.h->off_hinv);
  const uint32_t H = c.h->H, nfree = c.h->nfree;
  const uint32_t hq = tid & 15u;
  if (hq >= H) return;
  if constexpr (BATCH) {
    const uint32_t nparts = nt >> 4;
    for (uint32_t f0 = tid >> 4; f0 < nfree; f0 += 8u * nparts) {
      uint32_t coef[8];
#pragma unroll
      for (uint32_t i = 0; i < 8; i++) coef[i] = hinv[(size_t)(f0 + i * nparts < nfree ? f0 + i * nparts : f0) * H + hq];
      const SV<WB> r = lds_get<WB, G>(c.slots(), c.h->S + hq);
#pragma unroll
      for (uint32_t i = 0; i < 8; i++) {
        const uint32_t f = f0 + i * nparts;
        if (f >= nfree || !coef[i]) continue;
        SV<WB> t = sv_mul<WB>(r, coef[i]);
        lds_xor<WB, G>(c.cf(), f, t);
      }
    }
    return;
  }
  for (uint32_t f = tid >> 4; f < nfree; f += nt >> 4) {
    uint32_t coef = hinv[(size_t)f * H + hq];
    if (!coef) continue;
    SV<WB> t = sv_mul<WB>(lds_get<WB, G>(c.slots(), c.h->S + hq), coef);
    lds_xor<WB, G>(c.cf(), f, t);
  }
}

/* phase 4d: values of all u inactive columns */
template <int WB, int G = 1, bool BATCH = false> SB_HD void ph_dense_cu(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const NRQ_GAS uint16_t *pivx = c.template arr<uint16_t>(c.h->off_pivx);
  const NRQ_GAS uint32_t *fbits = c.template arr<uint32_t>(c.h->off_fbits);
  const NRQ_GAS uint16_t *freex = c.template arr<uint16_t>(c.h->off_freex);
  uint32_t fx0 = 0, px0 = 0;
  if constexpr (BATCH) { /* (the map entries of both loops' first trips in flight together) */
    if (tid < c.h->nfree) fx0 = freex[tid];
    if (tid < c.h->r2) px0 = pivx[tid];
  }
  for (uint32_t p = tid; p < c.h->r2; p += nt) {
    SV<WB> v = lds_get<WB, G>(c.slots(), c.h->M + p);
    uint32_t fb = fbits[p];
    while (fb) {
      uint32_t f = (uint32_t)__builtin_ctz(fb);
      fb &= fb - 1u;
      SV<WB> t = lds_get<WB, G>(c.cf(), f);
      sv_xor<WB>(v, t);
    }
    lds_put<WB, G>(c.cu(), (BATCH && p == tid) ? px0 : (uint32_t)pivx[p], v);
  }
  for (uint32_t f = tid; f < c.h->nfree; f += nt) lds_put<WB, G>(c.cu(), (BATCH && f == tid) ? fx0 : (uint32_t)freex[f], lds_get<WB, G>(c.cf(), f));
}

/* table entry e of the back-substitution tables.  12-byte strips keep them 16 bytes apart: an entry is then ONE 16-byte LDS read at
 * an aligned address (as three dwords, two read instructions, a lookup cost 7.7 clocks against 5.6 on the 16-byte strip) */
template <int WB, int G = 1> SB_HD SV<WB> t4_get(const uint8_t *t4, uint32_t e) {
  if constexpr (WB == 12) {
    SV<12> r = sv_zero<12>();
    const uint4 v = reinterpret_cast<const uint4 *>(t4)[e];
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z;
    return r;
  } else return lds_get<WB, G>(t4, e);
}
template <int WB, int G = 1> SB_HD void t4_put(uint8_t *t4, uint32_t e, const SV<WB> &v) {
  if constexpr (WB == 12) {
    uint4 t; t.x = v.w[0]; t.y = v.w[1]; t.z = v.w[2]; t.w = 0u;
    reinterpret_cast<uint4 *>(t4)[e] = t;
  } else lds_put<WB, G>(t4, e, v);
}
/* phase 5a: 16-entry XOR tables over groups of 4 inactive columns (region X is reused) */
template <int WB, int G = 1> SB_HD void ph_tables(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const uint32_t ngroups = c.h->wpr * 8u, u = c.h->u;
  for (uint32_t e = tid; e < ngroups * 16u; e += nt) {
    uint32_t grp = e >> 4, nib = e & 15u;
    SV<WB> v = sv_zero<WB>();
#pragma unroll
    for (uint32_t bq = 0; bq < 4; bq++) {
      uint32_t x = grp * 4u + bq;
      if (((nib >> bq) & 1u) && x < u) {
        SV<WB> t = lds_get<WB, G>(c.cu(), x);
        sv_xor<WB>(v, t);
      }
    }
    t4_put<WB, G>(c.t4(), e, v);
  }
}

/* phase 5b: back substitution C(pivot k) = Y_k ^ W_k * C_u.  Two pivots per trip with separate
 * register sets: the W words of the second are in flight while the first does its table lookups
 * (no register hand-over between trips, so the compiler can leave the loads outstanding). */
template <int WB, int NW, int G = 1, bool FAST = true>
SB_HD void backsub_one(const StripCtx<WB, G> &c, const uint8_t *t4, uint32_t slot, const uint32_t (&bitsw)[NW],
                       uint32_t wpr) {
  SV<WB> acc = lds_get<WB, G>(c.slots(), slot);
#pragma unroll
  for (uint32_t w = 0; w < (uint32_t)NW; w++) {
    if (w >= wpr) break; /* tables exist for wpr*8 groups only */
    const uint32_t bits = bitsw[w];
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (WB == 16 && G == 1) {
      /* the phase is bound by VALU issue and LDS reads about equally (the 4 XORs per lookup are a given): both
       * nibbles of every byte are brought to "16 * nibble" in place with three operations per word, and a table
       * address is ONE byte-select add (the compiler's own sequence is shift, mask, add per lookup).
       * 8-bit tables in passes were measured too: half the lookups, but 64 lanes then read 64 different entries
       * instead of at most 16, and the phase becomes LDS-bandwidth bound at 1.6x the time. */
      uint32_t odd = bits & 0xF0F0F0F0u, even = (bits << 4) & 0xF0F0F0F0u;
      asm volatile("" : "+v"(odd), "+v"(even)); /* keep the packed form */
      /* all eight lookups of the word in flight together, each address one byte-select add with the table's offset in the
       * instruction's immediate field, then the 32 dwords folded three at a time (v_bitop3_b32: a ^ b ^ c).  As it was -- four
       * lookups in flight, then four one by one, each a full LDS round trip the wave waited for, and three VALU operations per
       * address -- the phase was bound by those round trips (47 k clocks per strip at K=8192 against 33 k of LDS time). */
      if constexpr (FAST) {
      const uint32_t tbw = (uint32_t)(uintptr_t)t4 + w * 2048u; /* (the word's tables; the lookup's own offset q * 256 is an immediate) */
      uint4 d[8];
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) d[q] = *NRQ_LDSP(uint4, t4_addr(tbw, (q & 1u) ? odd : even, q >> 1) + q * 256u);
#pragma unroll
      for (uint32_t q = 0; q < 8; q += 2) {
        acc.w[0] = nrq_xor3(acc.w[0], d[q].x, d[q + 1].x); acc.w[1] = nrq_xor3(acc.w[1], d[q].y, d[q + 1].y);
        acc.w[2] = nrq_xor3(acc.w[2], d[q].z, d[q + 1].z); acc.w[3] = nrq_xor3(acc.w[3], d[q].w, d[q + 1].w);
      }
      } else { /* (the register-lean form of the small workgroups) */
        const uint32_t tb = (uint32_t)(uintptr_t)t4;
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) {
        const uint32_t a = t4_addr(tb, (q & 1u) ? odd : even, q >> 1) + (w * 8u + q) * 256u;
        const uint4 v = *NRQ_LDSP(uint4, a);
        acc.w[0] ^= v.x; acc.w[1] ^= v.y; acc.w[2] ^= v.z; acc.w[3] ^= v.w;
        if (q == 3) NRQ_SCHED_FENCE();
      }
      }
      NRQ_SCHED_FENCE();
      continue;
    }
    if constexpr (WB == 8 && G == 1) {
      /* the same for 8-byte strips (blocks whose 16-byte image does not fit the LDS: K from ~8500 on): table entries of 8
       * bytes, so a nibble is brought to "8 * nibble" in place and a word's tables are 1 KB */
      uint32_t odd = (bits >> 1) & 0x78787878u, even = (bits << 3) & 0x78787878u;
      asm volatile("" : "+v"(odd), "+v"(even));
      const uint32_t tbw = (uint32_t)(uintptr_t)t4 + w * 1024u;
      uint2 d[8];
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) d[q] = *NRQ_LDSP(uint2, t4_addr(tbw, (q & 1u) ? odd : even, q >> 1) + q * 128u);
#pragma unroll
      for (uint32_t q = 0; q < 8; q += 2) {
        acc.w[0] = nrq_xor3(acc.w[0], d[q].x, d[q + 1].x); acc.w[1] = nrq_xor3(acc.w[1], d[q].y, d[q + 1].y);
      }
      NRQ_SCHED_FENCE();
      continue;
    }
    if constexpr (WB == 12 && G == 1) {
      /* 12-byte strips: entries 16 bytes apart (t4_get), looked up like the 16-byte strip's */
      uint32_t odd = bits & 0xF0F0F0F0u, even = (bits << 4) & 0xF0F0F0F0u;
      asm volatile("" : "+v"(odd), "+v"(even));
      const uint32_t tbw = (uint32_t)(uintptr_t)t4 + w * 2048u;
      uint4 d[8];
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) d[q] = *NRQ_LDSP(uint4, t4_addr(tbw, (q & 1u) ? odd : even, q >> 1) + q * 256u);
      uint32_t pad = 0; /* (the entries' fourth dword, zero: used, so that the read stays a ds_read_b128 -- left to itself the compiler
                         * narrows it to ds_read_b96, which this LDS serves at half the rate: the phase took 116 k clocks against
                         * 68 k with three dword reads per entry) */
#pragma unroll
      for (uint32_t q = 0; q < 8; q += 2) {
        acc.w[0] = nrq_xor3(acc.w[0], d[q].x, d[q + 1].x); acc.w[1] = nrq_xor3(acc.w[1], d[q].y, d[q + 1].y);
        acc.w[2] = nrq_xor3(acc.w[2], d[q].z, d[q + 1].z); pad = nrq_xor3(pad, d[q].w, d[q + 1].w);
      }
      asm volatile("" : : "v"(pad));
      NRQ_SCHED_FENCE();
      continue;
    }
#endif
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
      const uint32_t nib = (bits >> (4u * q)) & 15u;
      SV<WB> t = t4_get<WB, G>(t4, (w * 8u + q) * 16u + nib);
      sv_xor<WB>(acc, t);
      if (q == 3) NRQ_SCHED_FENCE(); /* 4 lookups in flight are enough; hoisting all NW*8 of them costs ~100 more registers */
    }
    NRQ_SCHED_FENCE();
  }
  lds_put<WB, G>(c.slots(), slot, acc);
}

template <int WB, int NW, int G = 1, bool FAST = true> SB_HD void backsub_fixed(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  if constexpr (!FAST) { /* the small workgroups: both register sets loaded, then both used (fewer live registers) */

  const NRQ_GAS uint16_t *pivslot = c.template arr<uint16_t>(c.h->off_pivslot);
  const NRQ_GAS uint32_t *wt = c.template arr<uint32_t>(c.h->off_wt);
  const uint32_t wpr = c.h->wpr, stride = c.h->npiv_pad, npiv = c.h->npiv;
  const uint8_t *t4 = c.t4();
  for (uint32_t k = tid; k < npiv; k += 2 * nt) {
    const uint32_t k2 = k + nt;
    uint32_t a[NW], b[NW];
    const uint32_t sa = pivslot[k];
    const uint32_t sb = k2 < npiv ? pivslot[k2] : 0u;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)NW; w++) a[w] = w < wpr ? wt[(size_t)w * stride + k] : 0u;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)NW; w++) b[w] = (w < wpr && k2 < npiv) ? wt[(size_t)w * stride + k2] : 0u;
    backsub_one<WB, NW, G, false>(c, t4, sa, a, wpr);
    if (k2 < npiv) backsub_one<WB, NW, G, false>(c, t4, sb, b, wpr);
  }
    return;
  }
  const NRQ_GAS uint16_t *pivslot = c.template arr<uint16_t>(c.h->off_pivslot);
  const NRQ_GAS uint32_t *wt = c.template arr<uint32_t>(c.h->off_wt);
  const uint32_t wpr = c.h->wpr, stride = c.h->npiv_pad, npiv = c.h->npiv;
  const uint8_t *t4 = c.t4();
  /* two register sets, filled alternately: the W words of the NEXT pivot are requested before the current one starts its
   * table lookups, so a trip to L2 is never waited for with nothing else to do (it was: both sets loaded, then both used) */
  uint32_t a[NW], b[NW], sa = 0, sb = 0;
  uint32_t k = tid;
  if (k < npiv) {
    sa = pivslot[k];
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)NW; w++) a[w] = w < wpr ? wt[(size_t)w * stride + k] : 0u;
  }
  for (; k < npiv; k += 2 * nt) {
    const uint32_t k2 = k + nt, k3 = k + 2 * nt;
    if (k2 < npiv) {
      sb = pivslot[k2];
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)NW; w++) b[w] = w < wpr ? wt[(size_t)w * stride + k2] : 0u;
    }
    backsub_one<WB, NW, G, FAST>(c, t4, sa, a, wpr);
    if (k3 < npiv) {
      sa = pivslot[k3];
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)NW; w++) a[w] = w < wpr ? wt[(size_t)w * stride + k3] : 0u;
    }
    if (k2 < npiv) backsub_one<WB, NW, G, FAST>(c, t4, sb, b, wpr);
  }
}

template <int WB, int G = 1, bool FAST = true> SB_HD void ph_backsub(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const uint32_t wpr = c.h->wpr;
  if (wpr <= 4) backsub_fixed<WB, 4, G, FAST>(c, tid, nt);
  else if (wpr <= 8) backsub_fixed<WB, 8, G, FAST>(c, tid, nt);
  else if (wpr <= 12) backsub_fixed<WB, 12, G, FAST>(c, tid, nt);
  else if (wpr <= 24) backsub_fixed<WB, 24, G, FAST>(c, tid, nt);
  else {
    const NRQ_GAS uint16_t *pivslot = c.template arr<uint16_t>(c.h->off_pivslot);
    const NRQ_GAS uint32_t *wt = c.template arr<uint32_t>(c.h->off_wt);
    const uint32_t stride = c.h->npiv_pad, npiv = c.h->npiv;
    const uint8_t *t4 = c.t4();
    for (uint32_t k = tid; k < npiv; k += nt) {
      uint32_t s = pivslot[k];
      SV<WB> acc = lds_get<WB, G>(c.slots(), s);
      for (uint32_t w = 0; w < wpr; w++) {
        uint32_t bits = wt[(size_t)w * stride + k];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
          uint32_t nib = (bits >> (4u * q)) & 15u;
          SV<WB> t = t4_get<WB, G>(t4, (w * 8u + q) * 16u + nib);
          sv_xor<WB>(acc, t);
        }
      }
      lds_put<WB, G>(c.slots(), s, acc);
    }
  }
}

/* phase 6a: park the inactive columns in the slots the plan reserved for them */
template <int WB, int G = 1> SB_HD void ph_park(const StripCtx<WB, G> &c, uint32_t tid, uint32_t nt) {
  const NRQ_GAS uint16_t *uslot = c.template arr<uint16_t>(c.h->off_uslot);
  for (uint32_t x = tid; x < c.h->u; x += nt) lds_put<WB, G>(c.slots(), uslot[x], lds_get<WB, G>(c.cu(), x));
}

/* phase 6b: results: intermediate symbols (optional) and generated symbols, into this strip's OUTPUT staging buffer
 * (element i < L: intermediate symbol i, if the job wants them; then the nout generated symbols).  Results leave
 * for their rows in HBM a line group at a time (pf_scatter): written strip by strip, every 16-byte piece would be a
 * partial-line write of its own (measured: 3.7x the bytes at the HBM interface). */
#ifndef NRQ_STORE_CHUNK
#define NRQ_STORE_CHUNK 8u /* (even) */
#endif
#ifndef NRQ_STORE_TRIP
#define NRQ_STORE_TRIP 32u /* (a multiple of the chunk) */
#endif
/* bytes that may be read behind out_slots[]: a trip from the last list's start, and a second one if a list were longer than one
 * trip (RFC 6330 lists are at most 32 entries = one trip; the slack covers RQ_LT_COLS_MAX_REAL) */
#define NRQ_STORE_SLACK (16u + NRQ_STORE_TRIP * 4u)
#define NRQ_LT_LIST_MAX 33u /* (= RQ_LT_COLS_MAX_REAL of rq_math.h; nrq_device.hip checks the two against each other) */
static_assert(NRQ_LT_LIST_MAX <= 2u * NRQ_STORE_TRIP, "ph_store reads at most two trips of a list; out_slots[] slack is sized for that");
template <int WB, int G = 1> SB_HD uint32_t out_elems(const nrq_job *job, const nrq_plan_hdr *h) { return (job->inter ? h->L : 0u) + job->nout; }
/* FAST: the form for the big workgroup (168 registers per thread); the 256- and 64-thread variants, built for 96-128 registers,
 * keep the lean loop (measured at K=1000: store phase 20 k -> 31 k clocks with the fast form there) */
/* The NRQ_STORE_TRIP slot numbers from entry e on, two per word, as 16-byte loads from a 2-byte-aligned address (the memory
 * path takes them).  With a 16-bit load per entry under `e + k < end` the compiler made a branch per entry and waited for each
 * load inside it: 32 trips to L2 one after the other, 15 k clocks for the 820 symbols of a decode strip.  The loads run past the
 * end of the list, the last list's past the end of the array: out_slots[] is followed by NRQ_STORE_SLACK bytes that may be
 * read (nrq_device.hip, planner_body.h). */
SB_HD void store_trip(const NRQ_GAS uint16_t *osl, uint32_t e, uint32_t end, uint32_t (&raw)[NRQ_STORE_TRIP / 2u]) {
  (void)end;
#ifdef __HIP_DEVICE_COMPILE__
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef u4 u4a __attribute__((aligned(2)));
#pragma unroll
  for (uint32_t k = 0; k < NRQ_STORE_TRIP / 8u; k++) {
    const u4 w = *reinterpret_cast<const NRQ_GAS u4a *>(osl + e + 8u * k);
    raw[4u * k] = w.x; raw[4u * k + 1u] = w.y; raw[4u * k + 2u] = w.z; raw[4u * k + 3u] = w.w;
  }
#else
#pragma unroll
  for (uint32_t k = 0; k < NRQ_STORE_TRIP / 2u; k++) /* (the emulator's arrays have no slack: stay inside the list) */
    raw[k] = (e + 2u * k < end ? (uint32_t)osl[e + 2u * k] : 0u) | (e + 2u * k + 1u < end ? (uint32_t)osl[e + 2u * k + 1u] : 0u) << 16;
#endif
}
/* entry e + k of the list as an element of the image: its slot, or (past the end of the list) the lane's scratch slot */
SB_HD uint32_t store_slot(const uint32_t (&raw)[NRQ_STORE_TRIP / 2u], uint32_t k, uint32_t e, uint32_t end, uint32_t zero_at) {
  return e + k < end ? ((raw[k / 2u] >> (16u * (k & 1u))) & 0xFFFFu) + NRQ_SCRATCH : zero_at;
}
template <int WB, int G = 1, bool FAST = true> SB_HD void ph_store(const StripCtx<WB, G> &c, NRQ_GAS uint8_t *ostage, uint32_t tid, uint32_t nt) {
  constexpr int STB = 8;
  const NRQ_GAS uint16_t *colslot = c.template arr<uint16_t>(c.h->off_colslot);
  const uint32_t L = c.h->L, ni = c.job->inter ? L : 0u;
  for (uint32_t base = tid; base < ni; base += STB * nt) {
    uint32_t sl[STB];
#pragma unroll
    for (int q = 0; q < STB; q++) {
      uint32_t col = base + (uint32_t)q * nt;
      sl[q] = col < L ? colslot[col] : 0u;
    }
#pragma unroll
    for (int q = 0; q < STB; q++) {
      uint32_t col = base + (uint32_t)q * nt;
      if (col < L) g_put_stream<WB>(ostage + (size_t)col * (WB * G), WB, lds_get<WB, G>(c.slots(), sl[q]));
    }
  }
  const NRQ_GAS uint32_t *cptr = gptr<uint32_t>(c.job->out_cptr);
  const NRQ_GAS uint16_t *osl = gptr<uint16_t>(c.job->out_slots);
  if constexpr (FAST) {
  /* A generated symbol per thread and pass: its list bounds (fetched a pass ahead), then the slot numbers NRQ_STORE_CHUNK at a
   * time in one trip, then the strips from LDS -- ALL of the chunk's, unconditionally: an entry beyond the end of the list reads
   * the lane's scratch slot, which holds zeros (ph_clear; the padding ops of the stream XOR it into itself), so the loop body
   * has no branch and its LDS reads are in flight together.  (With a test per entry every read was waited for on its own: 14 k
   * clocks for the 860 symbols of a decode strip.)  The waves of a pass go on as long as one lane has entries left; an LT list
   * has 7 entries on average and 33 at most. */
  const uint32_t nout = c.job->nout;
  constexpr uint32_t CH = NRQ_STORE_CHUNK;
  const uint32_t zero_at = tid & (NRQ_SCRATCH - 1u); /* (element index from the start of the image) */
  uint32_t e_n = 0, end_n = 0;
  if (tid < nout) { e_n = cptr[tid]; end_n = cptr[tid + 1]; }
  const uint32_t npass = (nout + nt - 1u) / nt;
  for (uint32_t pass = 0, q = tid; pass < npass; pass++, q += nt) {
    uint32_t e = e_n;
    const uint32_t end = end_n;
    e_n = end_n = 0;
    if (q + nt < nout) { e_n = cptr[q + nt]; end_n = cptr[q + nt + 1]; }
    SV<WB> acc = sv_zero<WB>();
    while (NRQ_WAVE_ANY(e < end)) {
      /* one trip to L2 for the next NRQ_STORE_TRIP entries of every lane's list (nearly always all that is left), then the
       * strips, a chunk of entries at a time while a lane of the wave still has some */
      constexpr uint32_t TR = NRQ_STORE_TRIP;
      uint32_t raw[TR / 2u];
      store_trip(osl, e, end, raw);
#pragma unroll
      for (uint32_t k0 = 0; k0 < TR; k0 += CH) {
        if (k0 && !NRQ_WAVE_ANY(e + k0 < end)) break;
        SV<WB> v[CH];
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) v[k] = lds_get<WB, G>(c.lds, store_slot(raw, k0 + k, e, end, zero_at));
#pragma unroll
        for (uint32_t k = 0; k + 1u < CH; k += 2u)
#pragma unroll
          for (int i = 0; i < SV<WB>::ND; i++) acc.w[i] = nrq_xor3(acc.w[i], v[k].w[i], v[k + 1u].w[i]);
      }
      e += TR;
    }
    if (q < nout) g_put_stream<WB>(ostage + (size_t)(ni + q) * (WB * G), WB, acc);
  }
  } else {
  /* a generated symbol per thread and pass: list bounds, then ALL slot numbers of the list in one trip (an LT list has at most
   * 30 + 3 entries: a chunk of NRQ_STORE_CHUNK covers it; eight at a time it was a chain of up to four dependent trips to L2
   * per symbol, and a wave waited for its longest list), then the strips from LDS.  The bounds of the thread's next symbol
   * are fetched meanwhile. */
  const uint32_t nout = c.job->nout;
  constexpr uint32_t CH = 32u;
  uint32_t e_n = 0, end_n = 0;
  if (tid < nout) { e_n = cptr[tid]; end_n = cptr[tid + 1]; }
  for (uint32_t q = tid; q < nout; q += nt) {
    uint32_t e = e_n;
    const uint32_t end = end_n;
    if (q + nt < nout) { e_n = cptr[q + nt]; end_n = cptr[q + nt + 1]; }
    SV<WB> acc = sv_zero<WB>();
    while (e < end) {
      uint32_t sl[CH];
#pragma unroll
      for (uint32_t k = 0; k < CH; k++) sl[k] = e + k < end ? (uint32_t)osl[e + k] : NRQ_NOSLOT;
#pragma unroll
      for (uint32_t k = 0; k < CH; k++)
        if (sl[k] != NRQ_NOSLOT) sv_xor<WB>(acc, lds_get<WB, G>(c.slots(), sl[k]));
      e += CH;
    }
    g_put_stream<WB>(ostage + (size_t)(ni + q) * (WB * G), WB, acc);
  }
  }
}
/* phase 6b for the SPLIT solve of narrow strips (big blocks, nrq_device.hip): instead of back-substitution and results,
 * the strip's slot image after the dense stage (element i < M: slot i, i.e. Y of the pivot rows) and the values of the
 * inactive columns (element M + x: C_u[x]) go to the output staging buffer; they reach full-width rows of a per-block
 * work buffer through the same scatter, where nrq_backsub_kernel finishes the solve on 32-byte strips. */
template <int WB, int G = 1> SB_HD void ph_store_raw(const StripCtx<WB, G> &c, NRQ_GAS uint8_t *ostage, uint32_t tid, uint32_t nt) {
  const uint32_t M = c.h->M, u = c.h->u;
  constexpr int STB = 4;
  for (uint32_t base = tid; base < M + u; base += STB * nt) {
    SV<WB> v[STB];
#pragma unroll
    for (int q = 0; q < STB; q++) {
      const uint32_t i = base + (uint32_t)q * nt;
      v[q] = i < M ? lds_get<WB, G>(c.slots(), i) : i < M + u ? lds_get<WB, G>(c.cu(), i - M) : sv_zero<WB>();
    }
#pragma unroll
    for (int q = 0; q < STB; q++) {
      const uint32_t i = base + (uint32_t)q * nt;
      if (i < M + u) g_put_stream<WB>(ostage + (size_t)i * (WB * G), WB, v[q]);
    }
  }
}
/* where the results of one line group of one block go */
template <int WB> struct GroupDst {
  NRQ_GAS uint8_t *inter, *out;
  const NRQ_GAS uint32_t *orow;
  uint32_t ni, nout, T, strip0, nstrips; /* ni = intermediate symbols staged (0 or L) */
  uint32_t lsub;
};
/* units [u0, u1) of the scatter, unit = (staged element, piece of the line): whole lines to the symbol rows */
template <int WB, int G, bool PIPELINED, int AL> SB_HD void pf_scatter_impl(const GroupDst<WB> &g, const NRQ_GAS uint8_t *ostage, size_t stage_stride, uint32_t u0,
                                        uint32_t u1, uint32_t p, uint32_t np, uint32_t sub) {
  const uint32_t lsub = g.lsub, pmask = (1u << lsub) - 1u;
  if constexpr (WB == 12 && AL == 1) { /* (the line group with the row's last, partial strip: pf_gather_impl) */
    if ((g.strip0 + (1u << lsub)) * 12u > g.T) { pf_scatter_impl<WB, G, PIPELINED, 2>(g, ostage, stage_stride, u0, u1, p, np, sub); return; }
  }
#ifndef NRQ_SCATTER_PB
#define NRQ_SCATTER_PB 4
#endif
#ifndef NRQ_SCATTER_PUT
#define NRQ_SCATTER_PUT g_put
#endif
#ifndef NRQ_W12_SCATTER_PB
#define NRQ_W12_SCATTER_PB NRQ_SCATTER_PB
#endif
  constexpr int PB = WB == 12 ? NRQ_W12_SCATTER_PB : NRQ_SCATTER_PB;
  if constexpr (!PIPELINED) {
  for (uint32_t base = u0 + p; base < u1; base += PB * np) {
      SV<WB> v[PB];
      uint32_t row[PB];
  #pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t u = base + (uint32_t)q * np, i = u >> lsub;
        row[q] = (u < u1 && i >= g.ni) ? g.orow[i - g.ni] : i;
        v[q] = u < u1 ? g_get_l2<WB>(ostage + (size_t)(u & pmask) * stage_stride + (size_t)i * (WB * G) + sub * WB) : sv_zero<WB>();
      }
  #pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t u = base + (uint32_t)q * np, i = u >> lsub, strip = g.strip0 + (u & pmask);
        if (u >= u1 || strip >= g.nstrips) continue;
        if constexpr (AL != 0) {
          if constexpr (AL == 2) { const uint32_t rem = g.T - strip * WB; g_put_al12((i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + (size_t)strip * WB, rem < 12u ? rem : 12u, v[q]); }
          else g_put_al<WB>((i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + (size_t)strip * WB, v[q]);
        } else if constexpr (G == 1) {
          const uint32_t rem = g.T - strip * WB;
          NRQ_GAS uint8_t *dst = (i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + (size_t)strip * WB;
          NRQ_SCATTER_PUT<WB>(dst, rem < (uint32_t)WB ? rem : (uint32_t)WB, v[q]);
        } else {
          const uint32_t at = strip * (WB * G) + sub * WB;
          if (at >= g.T) continue;
          const uint32_t rem = g.T - at;
          NRQ_GAS uint8_t *dst = (i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + at;
          NRQ_SCATTER_PUT<WB>(dst, rem < (uint32_t)WB ? rem : (uint32_t)WB, v[q]);
        }
      }
    }
    return;
  }
  /* two stages (staged element and its row number -> symbol row), pipelined like pf_gather: the loads of trip t + 1 are on
   * their way before trip t is stored */
  const uint32_t first = u0 + p, step = (uint32_t)PB * np;
  if (first >= u1) return;
  SV<WB> v[PB], vn[PB];
  uint32_t row[PB], rown[PB];
  auto load = [&](uint32_t base, SV<WB> (&vv)[PB], uint32_t (&rr)[PB]) {
#pragma unroll
    for (int q = 0; q < PB; q++) {
      const uint32_t u = base + (uint32_t)q * np, i = u >> lsub;
      rr[q] = (u < u1 && i >= g.ni) ? g.orow[i - g.ni] : i;
      vv[q] = u < u1 ? g_get_l2<WB>(ostage + (size_t)(u & pmask) * stage_stride + (size_t)i * (WB * G) + sub * WB) : sv_zero<WB>();
    }
  };
  load(first, v, row);
  for (uint32_t base = first; base < u1; base += step) {
    if (base + step < u1) load(base + step, vn, rown);
#pragma unroll
    for (int q = 0; q < PB; q++) {
      const uint32_t u = base + (uint32_t)q * np, i = u >> lsub, strip = g.strip0 + (u & pmask);
      if (u >= u1 || strip >= g.nstrips) continue;
      if constexpr (AL != 0) {
        if constexpr (AL == 2) { const uint32_t rem = g.T - strip * WB; g_put_al12((i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + (size_t)strip * WB, rem < 12u ? rem : 12u, v[q]); }
        else g_put_al<WB>((i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + (size_t)strip * WB, v[q]);
      } else if constexpr (G == 1) {
        const uint32_t rem = g.T - strip * WB;
        NRQ_GAS uint8_t *dst = (i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + (size_t)strip * WB;
        NRQ_SCATTER_PUT<WB>(dst, rem < (uint32_t)WB ? rem : (uint32_t)WB, v[q]);
      } else {
        const uint32_t at = strip * (WB * G) + sub * WB;
        if (at >= g.T) continue;
        const uint32_t rem = g.T - at;
        NRQ_GAS uint8_t *dst = (i < g.ni ? g.inter : g.out) + (size_t)row[q] * g.T + at;
        NRQ_SCATTER_PUT<WB>(dst, rem < (uint32_t)WB ? rem : (uint32_t)WB, v[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < PB; q++) { v[q] = vn[q]; row[q] = rown[q]; }
  }
}

template <int WB, int G = 1, bool PIPELINED = false> SB_HD void pf_scatter(const GroupDst<WB> &g, const NRQ_GAS uint8_t *ostage, size_t stage_stride, uint32_t u0,
                                        uint32_t u1, uint32_t p, uint32_t np, uint32_t sub = 0) {
#ifndef NRQ_NO_AL
  if constexpr (G == 1 && WB >= 4) {
    const bool al = g.T % (uint32_t)(WB == 12 ? 4 : WB) == 0u && ((reinterpret_cast<uintptr_t>(g.inter) | reinterpret_cast<uintptr_t>(g.out)) & SVAlign<WB>::mask) == 0;
    if (al) { pf_scatter_impl<WB, G, PIPELINED, true>(g, ostage, stage_stride, u0, u1, p, np, sub); return; }
  }
#endif
  pf_scatter_impl<WB, G, PIPELINED, false>(g, ostage, stage_stride, u0, u1, p, np, sub);
}

#endif /* NRQ_SOLVE_BODY_H */
