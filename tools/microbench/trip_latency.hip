// Microbenchmark (gfx950): what a dependent "trip" costs a planner workgroup -- LDS read, LDS atomic with return, global load
// that hits L2, flat load of the same, workgroup barrier (16 waves) -- alone on the GPU and with every CU running a copy.
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/trip_latency.hip -o trip_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define N 200
__global__ __launch_bounds__(1024) void k(const uint32_t *__restrict__ chain, uint32_t nchain, unsigned long long *out, uint32_t *sink) {
  __shared__ uint32_t lds[8192];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 8192; i += blockDim.x) lds[i] = (i * 2654435761u + 12345u) & 8191u;
  __syncthreads();
  unsigned long long t[8];
  uint32_t x = tid & 8191u, acc = 0;
  // (a) LDS pointer chase, every wave
  t[0] = clock64();
  for (int i = 0; i < N; i++) x = lds[x];
  acc += x;
  t[1] = clock64();
  // (b) LDS atomic with return, dependent
  for (int i = 0; i < N; i++) x = (atomicAdd(&lds[x], 0u) + x) & 8191u;
  acc += x;
  t[2] = clock64();
  // (c) global pointer chase (L2 resident buffer)
  uint32_t g = (tid * 97u + blockIdx.x * 13u) % nchain;
  const __attribute__((address_space(1))) uint32_t *gc = (const __attribute__((address_space(1))) uint32_t *)chain;
  for (int i = 0; i < N; i++) g = gc[g];
  acc += g;
  t[3] = clock64();
  // (d) barrier alone
  for (int i = 0; i < N; i++) __syncthreads();
  t[4] = clock64();
  // (e) LDS read + barrier (a steering read per phase)
  for (int i = 0; i < N; i++) { x = lds[x]; __syncthreads(); }
  acc += x;
  t[5] = clock64();
  // (f) only wave 0 chases LDS, the others wait at one barrier
  if (tid < 64) { for (int i = 0; i < N; i++) x = lds[x]; acc += x; }
  __syncthreads();
  t[6] = clock64();
  // (g) only wave 0 chases global
  if (tid < 64) { for (int i = 0; i < N; i++) g = gc[g]; acc += g; }
  __syncthreads();
  t[7] = clock64();
  if (tid == 0) for (int i = 0; i < 7; i++) out[blockIdx.x * 8 + i] = t[i + 1] - t[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
  const uint32_t nchain = 65536; // 256 KB
  std::vector<uint32_t> h(nchain);
  for (uint32_t i = 0; i < nchain; i++) h[i] = (uint32_t)(((uint64_t)i * 40503u + 977u) % nchain);
  uint32_t *d, *sink; unsigned long long *out;
  hipMalloc(&d, nchain * 4); hipMalloc(&sink, 4); hipMalloc(&out, 256 * 8 * 8);
  hipMemcpy(d, h.data(), nchain * 4, hipMemcpyHostToDevice);
  const char *names[7] = {"LDS read chain", "LDS atomic-rtn chain", "global load chain (L2)", "barrier", "LDS read + barrier", "wave 0 LDS chain", "wave 0 global chain"};
  for (int nt : {64, 256, 1024}) for (int grid : {1, 256}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(grid), dim3(nt), 0, 0, d, nchain, out, sink); hipDeviceSynchronize(); }
    unsigned long long r[8];
    hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    printf("threads %4d grid %3d:", nt, grid);
    for (int i = 0; i < 7; i++) printf("  %s %.0f", names[i], (double)r[i] / N);
    printf("\n");
  }
  return 0;
}
