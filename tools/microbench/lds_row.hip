// Microbenchmark (gfx950): cost of one "row" of the forward pass on ONE wave -- 64 lanes, each
// ds_read_b128 from a random 16-byte slot and XORs it into another random slot -- under different
// instruction mixes.  Prints shader clocks per row.  Build: hipcc --offload-arch=gfx950 -O3 lds_row.hip -o lds_row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define LDS_BYTES (140 * 1024)
#define NSLOT (LDS_BYTES / 16)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u4;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

__device__ __forceinline__ uint4 rd(uint32_t a) { u32x4 t = *(lds_u4 *)(uintptr_t)a; return make_uint4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void xr(uint32_t a, uint4 v) {
  __hip_atomic_fetch_xor((lds_u64 *)(uintptr_t)a, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_xor((lds_u64 *)(uintptr_t)(a + 8), (unsigned long long)v.z | ((unsigned long long)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rmw(uint32_t a, uint4 v) {
  u32x4 t = *(lds_u4 *)(uintptr_t)a;
  t.x ^= v.x; t.y ^= v.y; t.z ^= v.z; t.w ^= v.w;
  *(lds_u4 *)(uintptr_t)a = t;
}

template <int MODE> __global__ __launch_bounds__(256) void k(const uint32_t *__restrict__ addr, uint32_t iters, unsigned long long *out, uint32_t nwaves) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < LDS_BYTES / 4; i += 256) ((uint32_t *)smem)[i] = i * 2654435761u;
  __syncthreads();
  if (tid >= 64 * nwaves) return;
  // 8 (src,dst) byte-address pairs per lane
  uint32_t s[8], d[8];
#pragma unroll
  for (int q = 0; q < 8; q++) { s[q] = addr[(q * 2) * 256 + tid]; d[q] = addr[(q * 2 + 1) * 256 + tid]; }
  unsigned long long t0 = clock64();
  if (MODE == 0) {          // pipelined P=3: R(k) ... X(k-3)
    uint4 v0 = rd(s[0]), v1 = rd(s[1]), v2 = rd(s[2]);
    for (uint32_t it = 0; it < iters; it++) {
      uint4 v3 = rd(s[3]); xr(d[0], v0);
      v0 = rd(s[4]); xr(d[1], v1);
      v1 = rd(s[5]); xr(d[2], v2);
      v2 = rd(s[6]); xr(d[3], v3);
      v3 = rd(s[7]); xr(d[4], v0);
      v0 = rd(s[0]); xr(d[5], v1);
      v1 = rd(s[1]); xr(d[6], v2);
      v2 = rd(s[2]); xr(d[7], v3);
    }
    xr(d[0], v0); xr(d[1], v1); xr(d[2], v2);
  } else if (MODE == 1) {   // dependent: read, wait, xor, next
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) { uint4 v = rd(s[q]); xr(d[q], v); __builtin_amdgcn_s_waitcnt(0xc07f); }
    }
  } else if (MODE == 2) {   // non-atomic read-modify-write
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) { uint4 v = rd(s[q]); rmw(d[q], v); }
    }
  } else if (MODE == 3) {   // reads only, pipelined (sum to keep them alive)
    uint4 acc = {0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) { uint4 v = rd(s[q]); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if (acc.x == 0x1234567u) out[100] = acc.y + acc.z + acc.w;
  } else if (MODE == 4) {   // atomics only
    uint4 v = {tid, tid * 3u, tid * 5u, tid * 7u};
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) xr(d[q], v);
    }
  } else if (MODE == 5) {   // dependent VALU chain, 8 ops per "row"
    uint32_t x = tid;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 64; q++) x = x * 3u + d[q & 7];
    }
    if (x == 0x1234567u) out[100] = x;
  } else if (MODE == 6) {   // independent VALU, 64 ops
    uint32_t x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = tid + q;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = x[q] * 3u + d[q];
    }
    uint32_t a = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) a ^= x[q];
    if (a == 0x1234567u) out[100] = a;
  } else if (MODE == 7) {   // pipelined P=3 with the unpack the kernel would do: op word -> two byte addresses
    uint32_t op[8];
#pragma unroll
    for (int q = 0; q < 8; q++) op[q] = (d[q] >> 4) | ((s[q] >> 4) << 16);
#define SA(o) (((o) >> 16) << 4)
#define DA(o) (((o) & 0xFFFFu) << 4)
    uint4 v0 = rd(SA(op[0])), v1 = rd(SA(op[1])), v2 = rd(SA(op[2]));
    for (uint32_t it = 0; it < iters; it++) {
      uint4 v3 = rd(SA(op[3])); xr(DA(op[0]), v0);
      v0 = rd(SA(op[4])); xr(DA(op[1]), v1);
      v1 = rd(SA(op[5])); xr(DA(op[2]), v2);
      v2 = rd(SA(op[6])); xr(DA(op[3]), v3);
      v3 = rd(SA(op[7])); xr(DA(op[4]), v0);
      v0 = rd(SA(op[0])); xr(DA(op[5]), v1);
      v1 = rd(SA(op[1])); xr(DA(op[6]), v2);
      v2 = rd(SA(op[2])); xr(DA(op[7]), v3);
#pragma unroll
      for (int q = 0; q < 8; q++) op[q] = __builtin_amdgcn_readfirstlane(it) == 0xFFFFFFFFu ? 0 : op[q]; // keep the unpack in the loop
    }
    xr(DA(op[0]), v0); xr(DA(op[1]), v1); xr(DA(op[2]), v2);
  }
  unsigned long long t1 = clock64();
  if ((tid & 63) == 0) out[tid >> 6] = t1 - t0;
}

int main() {
  std::vector<uint32_t> h(16 * 256);
  uint32_t x = 12345;
  for (int conf = 0; conf < 2; conf++) {
    for (size_t i = 0; i < h.size(); i++) {
      x = x * 1664525u + 1013904223u;
      uint32_t lane = i % 256, q = i / 256;
      h[i] = conf == 0 ? ((x >> 8) % NSLOT) * 16 : ((lane % 64) * 16 + q * 1024 + (lane / 64) * 32768) % LDS_BYTES;
    }
    uint32_t *d_addr; unsigned long long *d_out;
    hipMalloc(&d_addr, h.size() * 4); hipMalloc(&d_out, 1024 * 8);
    hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const uint32_t iters = 2000;
    const char *names[] = {"pipelined R+X atomics P=3", "dependent R,X,wait", "non-atomic RMW", "reads only", "atomics only (2 x b64)",
                           "VALU dependent x64", "VALU independent x64", "pipelined + unpack of op words"};
#define RUN(M, NW) do { hipFuncSetAttribute((const void *)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(256), LDS_BYTES, 0, d_addr, iters, d_out, NW); hipDeviceSynchronize(); \
    unsigned long long o[4]; hipMemcpy(o, d_out, 32, hipMemcpyDeviceToHost); \
    printf("%-8s waves=%d  %-34s %8.1f clk per row (8 rows/iter; VALU modes: per 8 instr)\n", conf ? "linear" : "random", NW, names[M], (double)o[0] / iters / 8.0); } while (0)
    RUN(0, 1); RUN(1, 1); RUN(2, 1); RUN(3, 1); RUN(4, 1); RUN(5, 1); RUN(6, 1); RUN(7, 1);
    RUN(0, 4); RUN(3, 4); RUN(4, 4); RUN(6, 4);
    hipFree(d_addr); hipFree(d_out);
  }
  return 0;
}
