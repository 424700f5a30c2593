// Microbenchmark (gfx950): VALU issue with INDEPENDENT instructions (16 accumulators): clocks per wave-instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 8192
template <int KIND> __global__ void k(uint32_t *out, unsigned long long *clk, uint32_t seed) {
  uint32_t r[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = (threadIdx.x + i) * 2654435761u ^ seed;
  uint32_t m = seed * 77u + threadIdx.x;
  __syncthreads();
  unsigned long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N / 16; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (KIND == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[j]) : "v"(m));
      if (KIND == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(r[j]) : "v"(m));
      if (KIND == 2) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r[j]) : "v"(m));
      if (KIND == 3) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[j]) : "v"(m));
      if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[j]) : "v"(m));
      if (KIND == 5) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[j]) : "v"(m));
    }
  }
  unsigned long long t1 = clock64();
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) x ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int KIND> void run(const char *name) {
  uint32_t *out; unsigned long long *clk, h;
  (void)hipMalloc(&out, 4 * 1024 * 64); (void)hipMalloc(&clk, 8);
  for (int nt : {64, 256, 512, 768, 1024}) {
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(nt), 0, 0, out, clk, 1u);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(nt), 0, 0, out, clk, 2u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-18s %4d threads (%d waves/SIMD): %.2f clk per wave-instruction, %.2f clk per instruction of the SIMD\n", name, nt, (nt + 255) / 256, (double)h / N, (double)h / N / ((nt + 255) / 256));
  }
}
int main() { run<0>("v_xor_b32"); run<1>("v_bitop3_b32"); run<2>("v_pk_mul_lo_u16"); run<3>("v_add_u32_sdwa"); run<4>("v_mul_lo_u32"); run<5>("v_and_or_b32"); return 0; }
