/*
 * solve_emu.cpp -- CPU emulation of the gfx950 solve workgroup (TEST SUPPORT ONLY).
 *
 * Compiles nanorq_amd/csrc/solve_body.h -- the exact per-thread phase functions the HIP kernel
 * runs -- with g++ and executes the 256 threads of each phase sequentially, with the workgroup
 * barriers of nrq_solve_kernel turned into loop boundaries.  It lets the non-GPU test tier
 * exercise strip indexing, the packed GF(256) arithmetic, the HDPC Horner evaluation and the
 * chunk/sync schedule against the oracle.  It is not part of the product and is never shipped in
 * libnanorq_hip.so.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../nanorq_amd/csrc/solve_body.h"

template <int WB> static int run_strip(const nrq_job &job, uint32_t T, uint32_t strip, const uint8_t *kc) {
  const uint32_t NT = NRQ_CHUNK;
  StripCtx<WB> c;
  c.job = job;
  c.plan = reinterpret_cast<const uint8_t *>(job.plan);
  c.h = reinterpret_cast<const nrq_plan_hdr *>(c.plan);
  if (c.h->status) return 0;
  c.kc = kc;
  c.lay = nrq_lds_plan(c.h, WB);
  std::vector<uint8_t> lds(c.lay.total + 64, 0xA5); /* garbage-filled like real LDS */
  c.lds = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~(uintptr_t)15);
  c.T = T;
  c.strip = strip;
  uint32_t rem = T - strip * WB;
  c.valid = rem < (uint32_t)WB ? rem : (uint32_t)WB;
#define PHASE(fn) for (uint32_t t = 0; t < NT; t++) fn<WB>(c, t, NT)
  PHASE(ph_load);
  const uint32_t *ops = c.template arr<uint32_t>(c.h->off_ops);
  const uint32_t nch = c.h->nchunk1 + c.h->nchunk2;
  for (uint32_t ch = 0; ch < nch; ch++)
    for (uint32_t t = 0; t < NT; t++) ph_op<WB>(c, ops[(size_t)ch * NT + t]);
  PHASE(ph_hdpc);
  PHASE(ph_dense_fold);
  PHASE(ph_dense_free);
  PHASE(ph_dense_cu);
  PHASE(ph_tables);
  PHASE(ph_backsub);
  PHASE(ph_park);
  PHASE(ph_store);
#undef PHASE
  return 1;
}

extern "C" uint32_t emu_lds_bytes(const uint8_t *plan, uint32_t wb) {
  return nrq_lds_plan(reinterpret_cast<const nrq_plan_hdr *>(plan), wb).total;
}

/* every pointer in `job` is a host pointer here */
extern "C" int emu_solve(const nrq_job *job, uint32_t T, uint32_t wb, const uint8_t *kc) {
  const uint32_t nstrips = (T + wb - 1) / wb;
  int r = 1;
  for (uint32_t s = 0; s < nstrips && r; s++) {
    switch (wb) {
      case 16: r = run_strip<16>(*job, T, s, kc); break;
      case 8: r = run_strip<8>(*job, T, s, kc); break;
      case 4: r = run_strip<4>(*job, T, s, kc); break;
      case 2: r = run_strip<2>(*job, T, s, kc); break;
      default: return -1;
    }
  }
  return r;
}
