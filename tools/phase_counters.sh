#!/bin/bash
# Runs ON THE GPU BOX (gpurun): hardware counters of the headline solve kernel PER PHASE.  The library is built eight times in the
# build container (tools/phase_counters.sh build): build_var/lib_stop{0..6}.so end every strip behind phase p (-DNRQ_STOP_AFTER=p:
# load, forward, HDPC, GF(2) combinations, dense, tables, back-substitution), the regular library is the whole strip.  Every
# variant runs the same batch (tools/phase_driver.py: K=8192, T=1280, 256 blocks) under rocprofv3 twice -- kernel trace (duration)
# and one PMC pass (SQ_* counters; never together with a trace: MI355X_MICROARCH.md) -- and tools/phase_counters.py differences
# consecutive variants into the per-phase table profiles/r6_phase_counters.txt.
#   bash tools/phase_counters.sh build            (build container, ~6 min on 4 cores)
#   gpurun --timeout 1800 -- 'bash tools/phase_counters.sh run'
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
case "${1:-run}" in
build)
  mkdir -p build_var
  for p in 0 1 2 3 4 5 6; do
    ( python -c "
from nanorq_amd import build as b
b.build_lib(out='$REPO/build_var/lib_stop$p.so', extra=['-DNRQ_STOP_AFTER=$p'])" > /tmp/build_stop$p.log 2>&1; echo "stop$p built rc=$?" ) &
    [ $((p % 4)) -eq 3 ] && wait
  done
  wait
  ;;
run)
  export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
  OUT=$REPO/gpurun_out/phase; rm -rf $OUT; mkdir -p $OUT
  CTRS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"
  cd /tmp
  for v in 0 1 2 3 4 5 6 full; do
    lib=$REPO/build_var/lib_stop$v.so; [ "$v" = full ] && lib=$REPO/nanorq_amd/libnanorq_hip.so
    [ -f "$lib" ] || { echo "no $lib"; continue; }
    NANORQ_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr_$v -- python $REPO/tools/phase_driver.py > $OUT/tr_$v.log 2>&1
    NANORQ_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $CTRS -d $OUT/pmc_$v -- python $REPO/tools/phase_driver.py > $OUT/pmc_$v.log 2>&1
  done
  python $REPO/tools/phase_counters.py $OUT > $OUT/r6_phase_counters.txt
  find $OUT -name "*.db" -delete; rm -rf $OUT/tr_*/ $OUT/pmc_*/
  cat $OUT/r6_phase_counters.txt
  ;;
esac
