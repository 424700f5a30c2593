// Microbenchmark (gfx950): how fast can a CU gather B bytes out of every PITCH-byte row?  One workgroup per CU
// (150 KB of LDS forces that), G workgroups active, each reading its own 10.5 MB block (8192 rows) at byte offset
// (wg % 80) * 16 like the solve kernel's strips do.  Prints clocks per workgroup and requests per clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define PITCH 1280
#define ROWS 8192
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int NB /* 16-byte pieces per row */, int SHARE /* workgroups sharing a block */>
__global__ __launch_bounds__(768) void k(const uint8_t *__restrict__ src, unsigned long long *out, uint32_t nblocks) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t wg = blockIdx.x, tid = threadIdx.x;
  // SHARE > 0: consecutive workgroups share a block (different XCDs).  SHARE < 0: the -SHARE workgroups of ONE XCD
  // that follow each other (wg, wg+8, wg+16, ...) share a block, like the solve kernel's strips of a line group
  const uint32_t blk = SHARE > 0 ? (wg / SHARE) % nblocks : ((wg >> 3) / (-SHARE) * 8 + (wg & 7)) % nblocks;
  const uint32_t strip = SHARE > 0 ? (wg % SHARE) : ((wg >> 3) % (-SHARE));
  const uint8_t *base = src + (size_t)blk * ROWS * PITCH + (size_t)strip * 16 * NB;
  u4 acc = {0, 0, 0, 0};
  __syncthreads();
  unsigned long long t0 = clock64();
  for (uint32_t r0 = tid; r0 < ROWS; r0 += 768 * 4) {
    u4 v[4][NB];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t r = r0 + q * 768;
#pragma unroll
      for (int b = 0; b < NB; b++)
        v[q][b] = r < ROWS ? *reinterpret_cast<const u4 *>(base + (size_t)r * PITCH + b * 16) : acc;
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int b = 0; b < NB; b++) acc ^= v[q][b];
  }
  reinterpret_cast<u4 *>(smem)[tid] = acc;
  __syncthreads();
  unsigned long long t1 = clock64();
  if (tid == 0) out[wg] = t1 - t0;
}
template <int NB, int SHARE> void run(const uint8_t *d, unsigned long long *d_out, uint32_t G, uint32_t nblocks, const char *what) {
  hipFuncSetAttribute((const void *)k<NB, SHARE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL((k<NB, SHARE>), dim3(G), dim3(768), 150 * 1024, 0, d, d_out, nblocks);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(G);
  hipMemcpy(h.data(), d_out, G * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += (double)x;
  s /= G;
  printf("%-28s G=%4u  bytes/row=%3d  avg %8.0f clk/workgroup  %.2f clk/request/CU  chip %.1f req/clk  useful %.2f TB/s at 2.4 GHz\n", what, G, NB * 16, s,
         s / (ROWS * NB), (double)G * ROWS * NB / s, (double)G * ROWS * NB * 16 / s * 2.4e9 / 1e12);
}
int main() {
  const uint32_t nblocks = 256;
  uint8_t *d; unsigned long long *d_out;
  hipMalloc(&d, (size_t)nblocks * ROWS * PITCH + 4096);
  hipMemset(d, 1, (size_t)nblocks * ROWS * PITCH + 4096);
  hipMalloc(&d_out, 4096 * 8);
  // SHARE=1: every workgroup its own block (no line sharing at all); SHARE=8: 8 workgroups share each 128-byte line
  run<1, 1>(d, d_out, 8, nblocks, "own block, 16 B");
  run<1, 1>(d, d_out, 32, nblocks, "own block, 16 B");
  run<1, 1>(d, d_out, 256, nblocks, "own block, 16 B");
  run<1, 8>(d, d_out, 256, nblocks, "8 share a line, 16 B");
  run<1, 80>(d, d_out, 256, nblocks, "80 share a block, 16 B");
  run<1, -8>(d, d_out, 256, nblocks, "8 of one XCD share a line");
  run<1, -32>(d, d_out, 256, nblocks, "32 of one XCD share a block");
  run<4, 1>(d, d_out, 256, nblocks, "own block, 64 B");
  run<8, 1>(d, d_out, 256, nblocks, "own block, 128 B");
  run<8, 1>(d, d_out, 32, nblocks, "own block, 128 B");
  return 0;
}
