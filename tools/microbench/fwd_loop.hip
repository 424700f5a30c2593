// Microbenchmark (gfx950): the solve kernel's OWN row pipeline (solve_body.h fwd_rows_half, two waves, op words streamed
// from global memory) on a synthetic op stream, alone on the GPU (one workgroup) or with every CU running a copy.
// Build: hipcc --offload-arch=gfx950 -O3 -w -I nanorq_amd/csrc tools/microbench/fwd_loop.hip -o fwd_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "solve_body.h"

#define NSLOT 8480
#ifndef NAPS
#define NAPS 1u
#endif
// variants of the loop body (VAR): 1 = op words stay in registers (no global loads); 2 = plain shifts instead of SDWA;
// 3 = loads, but no LDS work; 4 = scalar base + 32-bit lane offset addressing of the op words
#if !defined(__HIP_DEVICE_COMPILE__)
template <int OFF, int VAR> __device__ void loop_var(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <int OFF, int VAR> __device__ void loop_addr(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <int OFF> __device__ void loop_hoist(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
template <int OFF> __device__ void loop_quad(const NRQ_GAS uint32_t *, uint32_t, uint32_t) {}
#else
template <int OFF, int VAR> __device__ __forceinline__ void loop_var(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  constexpr uint32_t P = NRQ_PIPE, NS = NRQ_PIPE + 1u, U = NRQ_RING;
  typedef uint32_t V __attribute__((ext_vector_type(2)));
  const NRQ_GAS uint32_t *nxt = ops + lane;
  uint32_t o[U];
  V v[NS];
#pragma unroll
  for (uint32_t k = 0; k < U; k++) o[k] = VAR == 1 ? nxt[k * NRQ_ROW + NRQ_RING * NRQ_ROW] : NRQ_NOP_AT(lane);
#pragma unroll
  for (uint32_t k = 0; k < NS; k++) { V z = {}; v[k] = z; }
  uint32_t voff = lane * 4u;
  for (uint32_t base = 0; base < nrows + P; base += U, nxt += U * NRQ_ROW, voff += U * NRQ_ROW * 4u) {
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t j = (k + U - P) % U;
      if (VAR != 3) {
        const uint32_t a = (VAR == 2 ? (o[j] & 0xFFFFu) << 4 : row_addr_lo<16>(o[j])) + (uint32_t)OFF;
        const V x = v[(k + NS - P) % NS];
        __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, a), __builtin_bit_cast(unsigned long long, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (VAR == 4) {
        uint32_t r;
        asm volatile("global_load_dword %0, %1, %2 offset:0" : "=v"(r) : "v"(voff + (k + U - P) * NRQ_ROW * 4u), "s"(ops) : "memory");
        o[j] = r;
      } else if (VAR != 1) o[j] = nxt[(k + U - P) * NRQ_ROW];
      if (VAR != 3) v[k % NS] = *NRQ_LDSP(V, (VAR == 2 ? (o[k] >> 16) << 4 : row_addr_hi<16>(o[k])) + (uint32_t)OFF);
      else { v[k % NS].x ^= o[k]; }
    }
  }
  if (VAR == 3 && v[0].x + v[1].x + v[2].x == 0x12345u) *NRQ_LDSP(uint32_t, 0) = 1u;
}
// the packed op words, but both LDS addresses of a step are formed BEFORE the step waits for its LDS data
template <int OFF> __device__ __forceinline__ void loop_hoist(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  constexpr uint32_t P = NRQ_PIPE, NS = NRQ_PIPE + 1u, U = NRQ_RING;
  typedef uint32_t V __attribute__((ext_vector_type(2)));
  const NRQ_GAS uint32_t *nxt = ops + lane;
  uint32_t o[U];
  V v[NS];
#pragma unroll
  for (uint32_t k = 0; k < U; k++) o[k] = NRQ_NOP_AT(lane);
#pragma unroll
  for (uint32_t k = 0; k < NS; k++) { V z = {}; v[k] = z; }
  for (uint32_t base = 0; base < nrows + P; base += U, nxt += U * NRQ_ROW) {
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t j = (k + U - P) % U;
      const uint32_t alo = row_addr_lo<16>(o[j]) + (uint32_t)OFF, ahi = row_addr_hi<16>(o[k]) + (uint32_t)OFF;
      __builtin_amdgcn_sched_barrier(0);
      const V x = v[(k + NS - P) % NS];
      __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, alo), __builtin_bit_cast(unsigned long long, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      v[k % NS] = *NRQ_LDSP(V, ahi);
      o[j] = nxt[(k + U - P) * NRQ_ROW];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// op words fetched four rows at a time: the stream holds the words of rows 4g..4g+3 of a lane next to each other (one 16-byte load)
template <int OFF> __device__ __forceinline__ void loop_quad(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  constexpr uint32_t P = NRQ_PIPE, NS = NRQ_PIPE + 1u, U = NRQ_RING; // U multiple of 4 and of NS
  static_assert(U % 4u == 0, "whole quads");
  typedef uint32_t V __attribute__((ext_vector_type(2)));
  typedef uint32_t Q __attribute__((ext_vector_type(4)));
  const NRQ_GAS Q *nxt = reinterpret_cast<const NRQ_GAS Q *>(ops) + lane;
  uint32_t o[U];
  V v[NS];
#pragma unroll
  for (uint32_t k = 0; k < U; k++) o[k] = NRQ_NOP_AT(lane);
#pragma unroll
  for (uint32_t k = 0; k < NS; k++) { V z = {}; v[k] = z; }
  for (uint32_t base = 0; base < nrows + P; base += U, nxt += (U / 4u) * NRQ_ROW) {
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t j = (k + U - P) % U;
      const uint32_t alo = row_addr_lo<16>(o[j]) + (uint32_t)OFF, ahi = row_addr_hi<16>(o[k]) + (uint32_t)OFF;
      __builtin_amdgcn_sched_barrier(0);
      const V x = v[(k + NS - P) % NS];
      __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, alo), __builtin_bit_cast(unsigned long long, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      v[k % NS] = *NRQ_LDSP(V, ahi);
      if (j % 4u == 3u) { // the quad whose last row has just been applied: refill it (U rows ahead)
        const Q w = nxt[(j / 4u + (j < k ? U / 4u : 0u)) * NRQ_ROW];
        o[j - 3u] = w.x; o[j - 2u] = w.y; o[j - 1u] = w.z; o[j] = w.w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// the same pipeline with byte addresses held in registers (5), or streamed as 64-bit records {target address, source address} (7)
template <int OFF, int VAR> __device__ __forceinline__ void loop_addr(const NRQ_GAS uint32_t *ops, uint32_t nrows, uint32_t lane) {
  constexpr uint32_t P = NRQ_PIPE, NS = NRQ_PIPE + 1u, U = NRQ_RING;
  typedef uint32_t V __attribute__((ext_vector_type(2)));
  const NRQ_GAS V *nxt = reinterpret_cast<const NRQ_GAS V *>(ops) + lane;
  V o[U];
  V v[NS];
#pragma unroll
  for (uint32_t k = 0; k < U; k++) { o[k].x = lane * 16u + OFF; o[k].y = lane * 16u + OFF; }
#pragma unroll
  for (uint32_t k = 0; k < NS; k++) { V z = {}; v[k] = z; }
  for (uint32_t base = 0; base < nrows + P; base += U, nxt += U * NRQ_ROW) {
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t j = (k + U - P) % U;
      const V x = v[(k + NS - P) % NS];
      __hip_atomic_fetch_xor(NRQ_LDSP(unsigned long long, o[j].x), __builtin_bit_cast(unsigned long long, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (VAR == 7) o[j] = nxt[(k + U - P) * NRQ_ROW];
      v[k % NS] = *NRQ_LDSP(V, o[k].y);
    }
  }
}
#endif
typedef uint32_t mv4 __attribute__((ext_vector_type(4)));
// traffic like the solve kernel's data movers: 16-byte pieces from one buffer to another, `depth` requests in flight per lane
__device__ __forceinline__ void nap(uint32_t units) { for (uint32_t i = 0; i < units; i++) __builtin_amdgcn_s_sleep(127); }
__device__ __forceinline__ void mover(const mv4 *__restrict__ a, mv4 *__restrict__ b, uint32_t n, uint32_t t, uint32_t nt, uint32_t iters, uint32_t flags) {
  const uint32_t pipelined = flags & 1u, naps = flags >> 8;
  for (uint32_t it = 0; it < iters; it++) {
    const uint32_t kind = (flags >> 1) & 7u;
    if (kind) {
      for (uint32_t i = t; i < n; i += 4 * nt) {
        mv4 v[4] = {};
        if (kind == 1u) {
#pragma unroll
          for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(a + ((i + q * nt) * 977u) % n);
          if (v[0].x == 0x12345u && v[1].y == 77u && v[2].z == 5u && v[3].w == 9u) b[0] = v[0];
        } else if (kind == 2u) {
#pragma unroll
          for (int q = 0; q < 4; q++) { v[q].x = i; __builtin_nontemporal_store(v[q], b + (i + q * nt) % n); }
        } else {
          uint32_t x = i;
          for (int q = 0; q < 400; q++) x = x * 1664525u + 1013904223u;
          if (x == 0x12345u) b[0] = v[0];
        }
        nap(naps);
      }
    } else if (!pipelined) {
      for (uint32_t i = t; i < n; i += 4 * nt) {
        mv4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(a + ((i + q * nt) * 977u) % n);
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_nontemporal_store(v[q], b + (i + q * nt) % n);
        nap(naps);
      }
    } else {
      mv4 v[4], w[4];
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(a + ((t + q * nt) * 977u) % n);
      for (uint32_t i = t; i < n; i += 4 * nt) {
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = __builtin_nontemporal_load(a + ((i + 4 * nt + q * nt) * 977u) % n);
#pragma unroll
        for (int q = 0; q < 4; q++) __builtin_nontemporal_store(v[q], b + (i + q * nt) % n);
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = w[q];
        nap(naps);
      }
    }
  }
}
__global__ __launch_bounds__(768) void k(const uint32_t *__restrict__ ops, uint32_t nrows, unsigned long long *out, uint32_t mode, const mv4 *mva = nullptr, mv4 *mvb = nullptr, uint32_t mvn = 0, uint32_t mviters = 0, uint32_t mvpipe = 0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t tid = threadIdx.x, wv = tid >> 6;
  for (uint32_t i = tid; i < NSLOT * 4; i += 768) ((uint32_t *)smem)[i] = i * 2654435761u;
  __syncthreads();
  const NRQ_GAS uint32_t *o = gptr<uint32_t>(ops) + (size_t)blockIdx.x * 0; // every workgroup the same stream
  unsigned long long t0 = clock64();
  if (wv < 2) __builtin_amdgcn_s_setprio(3);
  if (mode == 0) { if (wv == 0) fwd_rows_half<16, 0>(o, nrows, tid); else if (wv == 1) fwd_rows_half<16, 8>(o, nrows, tid & 63u); }
#define VARIANT(V) if (mode == 10 + V) { if (wv == 0) loop_var<0, V>(o, nrows, tid); else if (wv == 1) loop_var<8, V>(o, nrows, tid & 63u); }
  VARIANT(1) VARIANT(2) VARIANT(3)
  if (mode == 48) { if (wv == 0) loop_quad<0>(o, nrows, tid); else if (wv == 1) loop_quad<8>(o, nrows, tid & 63u); }
  if (mode == 28) { if (wv == 0) loop_hoist<0>(o, nrows, tid); }
  if (mode == 38) { if (wv == 0) loop_hoist<0>(o, nrows, tid); else if (wv == 1) loop_hoist<8>(o, nrows, tid & 63u); }
  if (mode == 25) { if (wv == 0) loop_addr<0, 5>(o, nrows, tid); }
  if (mode == 27) { if (wv == 0) loop_addr<0, 7>(o, nrows, tid); }
  if (mode == 37) { if (wv == 0) loop_addr<0, 7>(o, nrows, tid); else if (wv == 1) loop_addr<8, 7>(o, nrows, tid & 63u); }
  if (mode == 21) { if (wv == 0) loop_var<0, 1>(o, nrows, tid); }
  if (mode == 20) { if (wv == 0) fwd_rows_half<16, 0>(o, nrows, tid); }  // one wave only
  else if (mode == 1 && wv == 2) fwd_rows<16>(o, nrows, tid & 63u); // (one wave, whole slots: for comparison only)
  if (mviters && (wv & 3u) >= 2u) mover(mva + (size_t)blockIdx.x * mvn, mvb + (size_t)blockIdx.x * mvn, mvn, (wv >> 2) * 128u + ((wv & 3u) - 2u) * 64u + (tid & 63u), 384u, mviters, mvpipe);
  unsigned long long t1 = clock64();
  if ((tid & 63u) == 0 && wv < 2) out[blockIdx.x * 2 + wv] = t1 - t0;
}

int main() {
  const uint32_t nrows = 1400, total = nrows + NRQ_RING + NRQ_PAD_ROWS;
  uint32_t x = 12345;
  for (int conf = 0; conf < 4; conf++) {
    std::vector<uint32_t> h((size_t)total * 64);
    for (size_t i = 0; i < h.size(); i++) {
      const uint32_t row = i / 64, lane = i % 64;
      if (row < NRQ_RING || row >= NRQ_RING + nrows || conf == 3) { h[i] = NRQ_NOP_AT(i); continue; }
      x = x * 1664525u + 1013904223u;
      const uint32_t src = (x >> 8) % (NSLOT - 128);
      x = x * 1664525u + 1013904223u;
      uint32_t dst = (x >> 8) % (NSLOT - 128);
      if (conf == 1) dst = (dst & ~7u) | ((lane & 15u) >> 1);                 // two targets per class and 16-lane block
      if (conf == 2) dst = (dst & ~63u) | ((lane * 37u + row) & 63u);          // 64 different residues mod 64
      h[i] = NRQ_OP(dst, src);
    }
    uint32_t *d_ops, *d_ops64; unsigned long long *d_out;
    std::vector<uint32_t> h64(h.size() * 2);
    for (size_t i = 0; i < h.size(); i++) { h64[2 * i] = (h[i] & 0xFFFFu) * 16u; h64[2 * i + 1] = (h[i] >> 16) * 16u; }
    uint32_t *d_opsq; std::vector<uint32_t> hq(h.size() + 1024, 0u);
    for (size_t i = 0; i < h.size(); i++) { const size_t r = i / 64, l = i % 64; hq[(r / 4) * 256 + l * 4 + (r % 4)] = h[i]; }
    hipMalloc(&d_opsq, hq.size() * 4); hipMemcpy(d_opsq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_ops, h.size() * 4); hipMalloc(&d_ops64, h64.size() * 4); hipMalloc(&d_out, 8 * 1024);
    hipMemcpy(d_ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ops64, h64.data(), h64.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * 16);
    const char *cn[] = {"random", "dst 2 per class", "dst spread", "all padding"};
    static mv4 *mva = nullptr, *mvb = nullptr;
    const uint32_t mvn = 16384; // 256 KB per workgroup and buffer
    if (!mva) { hipMalloc(&mva, (size_t)256 * mvn * 16); hipMalloc(&mvb, (size_t)256 * mvn * 16); hipMemset(mva, 1, (size_t)256 * mvn * 16); }
    for (uint32_t grid : {256u})
    for (uint32_t mode : {0u})
    for (uint32_t mv : {0u, 1u, 3u, 4u, 5u}) { // no movers / plain movers / pipelined movers (both napping between trips: ~one trip per 8 k clocks, like the kernel's)
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(768), NSLOT * 16, 0, mode == 48 ? d_opsq : d_ops, NRQ_RING + nrows, d_out, mode, mva, mvb, mvn, mv ? 1u : 0u, (mv == 2u ? 1u : mv >= 3u ? (mv - 2u) << 1 : 0u) | (NAPS << 8));
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> o(grid * 2);
      hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
      printf("%-16s mode %2u workgroups %3u movers %u  %6.1f clk per row (wave 0)\n", cn[conf], mode, grid, mv, (double)o[0] / (NRQ_RING + nrows));
    }
    hipFree(d_ops); hipFree(d_out);
  }
  return 0;
}
