#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes of the default bench workload,
# summarised into gpurun_out/prof/ (copy what you want judged into profiles/).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0"
cd /tmp
timeout 600 python $REPO/bench.py > "$OUT/bench_1gpu.json" 2> "$OUT/bench_1gpu.err"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
python $REPO/tools/rocprof_summary.py stats $(find "$OUT/trace" -name '*.db' | head -1) > "$OUT/kernel_stats.txt"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
done
python $REPO/tools/rocprof_summary.py pmc $(find "$OUT" -path '*pmc*' -name '*.db' | sort) > "$OUT/pmc.txt"
python $REPO/tools/rocprof_summary.py pmcjson $(find "$OUT" -path '*pmc*' -name '*.db' | sort) > "$OUT/pmc_hbm.json"
find "$OUT" -name "*.db" -delete
ls -la "$OUT"
