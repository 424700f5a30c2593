// Microbenchmark (gfx950): what ONE wave pays per row of 64 XOR ops in the solve kernel's own row pipeline (solve_body.h fwd_rows)
// at every strip width -- the floor of the forward window of narrow strips, where a single wave walks the op stream (the LDS of a CU
// holds one image; byte columns of a 2- or 4-byte strip cannot be shared between waves).  Random full rows (big blocks' streams are
// op-count bound: 63 of 64 lanes filled), slot count of the block sizes that use the width, op words quad-interleaved as in a plan.
// Build: hipcc --offload-arch=gfx950 -O3 -w -I nanorq_amd/csrc tools/microbench/narrow_rows.hip -o tools/microbench/narrow_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "solve_body.h"

template <int WB, int HALF> __global__ __launch_bounds__(192) void k(const uint32_t *__restrict__ ops, uint32_t nrows, uint32_t nslot, unsigned long long *out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t tid = threadIdx.x, wv = tid >> 6;
  for (uint32_t i = tid; i < nslot * WB / 4u; i += blockDim.x) ((uint32_t *)smem)[i] = i * 2654435761u;
  __syncthreads();
  const NRQ_GAS uint32_t *o = gptr<uint32_t>(ops);
  const unsigned long long t0 = clock64();
  if constexpr (WB == 12) { // three waves, a dword each
    if (wv == 0) fwd_rows_third<0>(o, nrows, tid); else if (wv == 1) fwd_rows_third<4>(o, nrows, tid & 63u); else fwd_rows_third<8>(o, nrows, tid & 63u);
  } else if constexpr (HALF) { // the 16- and 8-byte strips' form: two waves, half the width each
    if (wv == 0) fwd_rows_half<WB, 0>(o, nrows, tid); else fwd_rows_half<WB, WB / 2>(o, nrows, tid & 63u);
  } else {
    if (wv == 0) fwd_rows<WB>(o, nrows, tid);
  }
  const unsigned long long t1 = clock64();
  if ((tid & 63u) == 0 && wv == 0) out[blockIdx.x] = t1 - t0;
}

// spread: the 64 ops of a row take their targets (and sources) from 64 different residues of the slot number mod 64 -- what a
// bank-aware placement of a level's ops into rows could reach at best; random otherwise (what the planners produce)
template <int WB, int HALF> static void run(const char *what, uint32_t nslot, uint32_t nrows, int spread = 0) { /* 1: targets and sources, 2: targets only, 3: sources only */
  const uint32_t total = NRQ_STREAM_ROWS(nrows + NRQ_PAD_ROWS);
  std::vector<uint32_t> h((size_t)total * 64, 0u);
  uint32_t x = 12345;
  for (uint32_t row = 0; row < total; row++)
    for (uint32_t lane = 0; lane < 64; lane++) {
      uint32_t w = NRQ_NOP_AT(lane);
      if (row < nrows && lane != 63u) {
        x = x * 1664525u + 1013904223u; uint32_t src = (x >> 8) % (nslot - 64u);
        x = x * 1664525u + 1013904223u; uint32_t dst = (x >> 8) % (nslot - 64u);
        if (spread == 1 || spread == 3) { src = (src & ~63u) | ((lane * 37u + row) & 63u); if (src >= nslot - 64u) src -= 64u; }
        if (spread == 1 || spread == 2) { dst = (dst & ~63u) | ((lane * 29u + 7u * row) & 63u); if (dst >= nslot - 64u) dst -= 64u; }
        if (spread == 4) { /* the planners' rule (plan.h): lane 2d + (r & 1) of a 16-lane block holds a target of class d = slot mod 8; sources random */
          dst = (dst & ~7u) | ((lane >> 1) & 7u); if (dst >= nslot - 64u) dst -= 64u; }
        if (spread == 5) { /* ... and the sources two per class and block as well */
          dst = (dst & ~7u) | ((lane >> 1) & 7u); if (dst >= nslot - 64u) dst -= 64u;
          src = (src & ~7u) | (((lane >> 1) + 3u) & 7u); if (src >= nslot - 64u) src -= 64u; }
        w = NRQ_OP(dst, src);
      }
      h[NRQ_OP_INDEX(row, lane)] = w;
    }
  uint32_t *d_ops; unsigned long long *d_out;
  hipMalloc(&d_ops, h.size() * 4); hipMalloc(&d_out, 8 * 256);
  hipMemcpy(d_ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k<WB, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (uint32_t grid : {1u, 256u}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<WB, HALF>), dim3(grid), dim3(WB == 12 ? 192 : 128), (nslot + 64u) * WB, 0, d_ops, nrows, nslot, d_out); hipDeviceSynchronize(); }
    std::vector<unsigned long long> o(grid);
    hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : o) s += (double)v;
    printf("%-46s workgroups %3u  %6.1f clocks per row of 64 ops  (%5.2f clocks per op-byte)\n", what, grid, s / grid / nrows, s / grid / nrows / (63.0 * WB));
  }
  hipFree(d_ops); hipFree(d_out);
}

int main() {
  printf("# one wave of the row pipeline (solve_body.h fwd_rows / fwd_rows_half), random full rows, no data movers beside it\n");
  run<2, 0>("2-byte strips, K'=56403 (57.4 k slots)", 57400u, 6000u);
  run<4, 0>("4-byte strips, K=27000 (27.7 k slots)", 27700u, 3000u);
  run<8, 1>("8-byte strips, two waves x 4 bytes, K=10000", 10500u, 1400u);
  run<16, 1>("16-byte strips, two waves x 8 bytes, K=8192", 8480u, 1400u);
  run<12, 0>("12-byte strips, three waves x 4 bytes, K=10000", 10500u, 1400u);
  printf("# the same with every row's 64 targets and 64 sources on 64 different slot residues mod 64 (no two lanes on one LDS bank group)\n");
  run<4, 0>("4-byte strips, spread", 27700u, 3000u, 1);
  run<8, 1>("8-byte strips, spread", 10500u, 1400u, 1);
  run<12, 0>("12-byte strips, spread", 10500u, 1400u, 1);
  run<16, 1>("16-byte strips, spread", 8480u, 1400u, 1);
  run<16, 1>("16-byte strips, targets spread only", 8480u, 1400u, 2);
  run<16, 1>("16-byte strips, sources spread only", 8480u, 1400u, 3);
  run<16, 1>("16-byte strips, targets by the planners' rule", 8480u, 1400u, 4);
  run<16, 1>("16-byte strips, targets and sources by that rule", 8480u, 1400u, 5);
  run<12, 0>("12-byte strips, targets by the planners' rule", 10500u, 1400u, 4);
  run<8, 1>("8-byte strips, targets by the planners' rule", 10500u, 1400u, 4);
  return 0;
}
