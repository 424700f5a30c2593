/* planner_host.h -- host symbolic stage (see planner_host.cpp). Plain C ABI. */
#ifndef NRQ_PLANNER_HOST_H
#define NRQ_PLANNER_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Per-K' HDPC constants (plan.h: nrq_kconst_hdr + arrays). Caller frees with nrq_host_free. 0 = ok. */
int nrq_host_kconst_build(uint32_t K, uint8_t **out, uint32_t *out_bytes);
/* Plan for the constraint system whose LT rows are isis[0..nrows): the first K' entries are the
 * rows S+H.. (ISI j itself for a present source/padding symbol, or the ISI of the repair symbol put
 * in that row), the rest are surplus repair rows L.. (reference nanorq.c:527-565).
 * Returns 0 and a malloc'ed arena (hdr.status tells solvable / singular); <0 on bad arguments. */
int nrq_host_plan_build(uint32_t K, uint32_t nrows, const uint32_t *isis, const uint8_t *kconst, uint8_t **out,
                        uint32_t *out_bytes);
void nrq_host_free(void *p);
#ifdef __cplusplus
}
#endif
#endif
