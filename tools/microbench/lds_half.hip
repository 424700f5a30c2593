// Microbenchmark (gfx950): the forward pass as the solve kernel runs it -- TWO waves, each on one 8-byte half of every
// 16-byte slot: per row a lane reads its half of a random source slot (ds_read_b64) and XORs it into its half of a
// random target slot (ds_xor_b64), software-pipelined.  Layout A: slots of 16 bytes, half h at offset 8h (what the
// kernel does).  Layout B: two planes of 8-byte elements, half h in plane h.  Prints shader clocks per row.
// Build: hipcc --offload-arch=gfx950 -O3 -w lds_half.hip -o lds_half
// Measured (MI355X, two waves, clocks per row): random slots 54.5 (A) / 41.8 (B); conflict-free 28 / 26; only the
// targets conflict-free 35 / 28.5; only the sources 43.5 / 32.5; two lanes per bank everywhere 29 -- a 2-way conflict
// is free, the cost of random slots is the 4-5-way conflicts of the LDS atomics.  In the solve kernel layout B made
// NO difference (rows stayed at ~58 clocks): with op fetch and unpack a row is ~10 instructions of ONE wave, and a
// single wave issues one every ~5 clocks -- the pipeline is issue-latency bound, not LDS bound, since it was split
// over two waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define NSLOT 8704
#define LDS_BYTES (NSLOT * 16)
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
__device__ __forceinline__ unsigned long long rd(uint32_t a) { return *(lds_u64 *)(uintptr_t)a; }
__device__ __forceinline__ void xr(uint32_t a, unsigned long long v) {
  __hip_atomic_fetch_xor((lds_u64 *)(uintptr_t)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LAYOUT 0: addr = slot*16 + 8*h;  1: addr = h*NSLOT*8 + slot*8
template <int LAYOUT> __global__ __launch_bounds__(256) void k(const uint32_t *__restrict__ slots, uint32_t iters, unsigned long long *out, uint32_t nwaves) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t tid = threadIdx.x, h = tid >> 6, lane = tid & 63u;
  for (uint32_t i = tid; i < LDS_BYTES / 4; i += 256) ((uint32_t *)smem)[i] = i * 2654435761u;
  __syncthreads();
  if (h >= nwaves) return;
  uint32_t s[8], d[8];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const uint32_t ss = slots[(q * 2) * 64 + lane], dd = slots[(q * 2 + 1) * 64 + lane]; // both waves: the same ops
    // LAYOUT 2: 16-byte slots whose halves are swapped when bit 4 of the slot number is set
    s[q] = LAYOUT == 0 ? ss * 16u + 8u * (h & 1u) : LAYOUT == 1 ? (h & 1u) * NSLOT * 8u + ss * 8u : ss * 16u + 8u * ((h ^ (ss >> 4)) & 1u);
    d[q] = LAYOUT == 0 ? dd * 16u + 8u * (h & 1u) : LAYOUT == 1 ? (h & 1u) * NSLOT * 8u + dd * 8u : dd * 16u + 8u * ((h ^ (dd >> 4)) & 1u);
  }
  unsigned long long t0 = clock64();
  unsigned long long v0 = rd(s[0]), v1 = rd(s[1]);
  for (uint32_t it = 0; it < iters; it++) { // P = 2
    unsigned long long v2 = rd(s[2]); xr(d[0], v0);
    v0 = rd(s[3]); xr(d[1], v1);
    v1 = rd(s[4]); xr(d[2], v2);
    v2 = rd(s[5]); xr(d[3], v0);
    v0 = rd(s[6]); xr(d[4], v1);
    v1 = rd(s[7]); xr(d[5], v2);
    v2 = rd(s[0]); xr(d[6], v0);
    v0 = v2; v2 = rd(s[1]); xr(d[7], v1);
    v1 = v2;
  }
  xr(d[0], v0); xr(d[1], v1);
  unsigned long long t1 = clock64();
  if (lane == 0) out[h] = t1 - t0;
}

int main() {
  std::vector<uint32_t> hs(16 * 64);
  uint32_t x = 12345;
  for (int conf = 0; conf < 10; conf++) {
    for (size_t i = 0; i < hs.size(); i++) {
      x = x * 1664525u + 1013904223u;
      const uint32_t lane = i % 64, q = i / 64;
      // 0: random slots; 1: consecutive slots; 2: random, but the 64 lanes of a row hit 64 different residues mod 64
      const uint32_t rnd = (x >> 8) % NSLOT, spread = ((x >> 8) % (NSLOT / 64)) * 64 + ((lane * 37 + q) & 63);
      const bool is_dst = q & 1;
      // 5: residues mod 32 distinct within each half of the wave; 6: at most two lanes per residue mod 64
      const uint32_t spread32 = ((x >> 8) % (NSLOT / 32)) * 32 + ((lane * 13 + q) & 31);
      const uint32_t pair2 = ((x >> 8) % (NSLOT / 64)) * 64 + (((lane >> 1) * 37 + q) & 63);
      // 7: targets: every 16-lane group holds each residue mod 8 twice (sources random); 8: targets: every 16-lane group holds
      // 16 different (residue mod 8, bit 4) classes; 9: as 8, and the sources of every 32-lane half hold 32 different residues mod 32
      const uint32_t blk = (x >> 8) % (NSLOT / 32);
      const uint32_t d7 = blk * 32 + ((x >> 3) & 3u) * 8 + (((lane & 15u) >> 1) + q) % 8;
      const uint32_t d8 = blk * 32 + ((lane + q) & 7u) + 8u * ((x >> 5) & 1u) + 16u * (((lane >> 3) + q) & 1u);
      const uint32_t s9 = blk * 32 + ((lane * 5 + q) & 31u);
      if (conf >= 7) { hs[i] = conf == 7 ? (is_dst ? d7 : rnd) : conf == 8 ? (is_dst ? d8 : rnd) : (is_dst ? d8 : s9); continue; }
      hs[i] = conf == 0 ? rnd : conf == 1 ? (lane + q * 97) % NSLOT : conf == 2 ? spread : conf == 3 ? (is_dst ? spread : rnd)
              : conf == 4 ? (is_dst ? rnd : spread) : conf == 5 ? spread32 : pair2;
    }
    uint32_t *d_s; unsigned long long *d_out;
    hipMalloc(&d_s, hs.size() * 4); hipMalloc(&d_out, 64);
    hipMemcpy(d_s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    const uint32_t iters = 4000;
    const char *cn[] = {"random", "linear", "spread", "dst-spr", "src-spr", "mod32", "2-way", "dst8x2", "dst16", "d16+s32"};
#define RUN(L, NW) do { hipFuncSetAttribute((const void *)k<L>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL(k<L>, dim3(1), dim3(256), LDS_BYTES, 0, d_s, iters, d_out, NW); hipDeviceSynchronize(); \
    unsigned long long o[4]; hipMemcpy(o, d_out, 32, hipMemcpyDeviceToHost); \
    printf("%-7s layout %s waves=%d  %6.1f clk per row\n", cn[conf], L == 1 ? "B (two 8-byte planes)" : L == 2 ? "C (halves swapped by bit 4)" : "A (16-byte slots)   ", NW, (double)o[0] / iters / 8.0); } while (0)
    RUN(0, 1); RUN(0, 2); RUN(2, 2);
    hipFree(d_s); hipFree(d_out);
  }
  return 0;
}
