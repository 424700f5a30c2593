// Microbenchmark (gfx950): what a TAKEN branch costs a wave (the instruction buffer is refilled from the instruction cache) --
// 16 taken s_branch per loop trip against 16 s_nop, near targets (skipping 1 instruction) and far ones (skipping 1 KB of code);
// one wave per CU, 16 waves per CU (a 1024-thread workgroup), every CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/branch_cost.hip -o branch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 500
#define BR_NEAR "s_branch 1f\n\ts_nop 0\n1:\n\t"
#define BR_FAR "s_branch 1f\n\t.rept 256\n\ts_nop 0\n\t.endr\n1:\n\t" /* 256 s_nop = 1 KB skipped */
#define X16(s) s s s s s s s s s s s s s s s s
__global__ __launch_bounds__(1024) void k(unsigned long long *out) {
  unsigned long long t[4];
  t[0] = clock64();
  _Pragma("unroll 1") for (int i = 0; i < N; i++) asm volatile(X16("s_nop 0\n\t") ::: "memory");
  t[1] = clock64();
  _Pragma("unroll 1") for (int i = 0; i < N; i++) asm volatile(X16(BR_NEAR) ::: "memory");
  t[2] = clock64();
  _Pragma("unroll 1") for (int i = 0; i < N; i++) asm volatile(X16(BR_FAR) ::: "memory");
  t[3] = clock64();
  if (threadIdx.x == 0) for (int i = 0; i < 3; i++) out[blockIdx.x * 4 + i] = t[i + 1] - t[i];
}
int main() {
  unsigned long long *out; hipMalloc(&out, 256 * 4 * 8);
  for (int nt : {64, 1024}) for (int grid : {1, 256}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(grid), dim3(nt), 0, 0, out); hipDeviceSynchronize(); }
    unsigned long long r[4]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    printf("threads %4d grid %3d: per loop trip of 16: s_nop %.0f  taken branch (near) %.0f  taken branch (1 KB away) %.0f clocks\n", nt, grid,
           (double)r[0] / N, (double)r[1] / N, (double)r[2] / N);
  }
  return 0;
}
