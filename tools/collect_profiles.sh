#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line (with its own live PMC passes), the rocprofv3 kernel trace
# of the same command, the PMC passes again as a stand-alone file, and the other BASELINE configs -- all into
# gpurun_out/prof/ (copy what you want judged into profiles/).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r3'
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r6}
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --pmc off --no-e2e"
cd /tmp
timeout 900 python $REPO/bench.py > "$OUT/${TAG}_bench_1gpu.json" 2> "$OUT/bench_1gpu.err"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
python $REPO/tools/rocprof_summary.py stats $(find "$OUT/trace" -name '*.db' | head -1) > "$OUT/${TAG}_kernel_stats.txt"
echo "# (avg_us is over ALL launches of the run: 1 warm-up step + 3 timed steps, i.e. it includes the cold first launch -- the max column; steady_avg_us leaves the warm-up step's launches out and is the figure that roofline.avg_launch_ms of ${TAG}_bench_1gpu.json must agree with)" >> "$OUT/${TAG}_kernel_stats.txt"
python $REPO/tools/rocprof_summary.py timeline $(find "$OUT/trace" -name "*.db" | head -1) -1 > "$OUT/${TAG}_kernel_timeline.txt"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
done
python $REPO/tools/rocprof_summary.py pmc $(find "$OUT" -path '*pmc*' -name '*.db' | sort) > "$OUT/${TAG}_pmc.txt"
python $REPO/tools/rocprof_summary.py pmcjson $(find "$OUT" -path '*pmc*' -name '*.db' | sort) > "$OUT/${TAG}_pmc.json"
# large-K and small-K kernel traces
for cfg in "cfg5 56403 1280 8 0.2" "cfg4 27000 65504 1 0.1" "cfg1 100 1024 8192 0.06" "K1000 1000 1280 2048 0.06" "K10000 10000 1280 256 0.06"; do set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/tr_$1" -- python $REPO/bench.py --K $2 --T $3 --blocks $4 --loss $5 --steps 3 --warmup 1 --cpu-sample 0 --pmc off --no-e2e > "$OUT/tr_$1.log" 2>&1
  python $REPO/tools/rocprof_summary.py stats $(find "$OUT/tr_$1" -name '*.db' | head -1) > "$OUT/${TAG}_kernel_stats_$1.txt"
done
find "$OUT" -name "*.db" -delete
rm -rf "$OUT"/trace "$OUT"/pmc? "$OUT"/tr_*/
bash $REPO/tools/bench_configs.sh $TAG > "$OUT/${TAG}_configs_summary.txt" 2>&1
cp $REPO/gpurun_out/cfg/${TAG}_bench_*.json "$OUT/" 2>/dev/null
bash $REPO/tools/reference_benchmark.sh > "$OUT/${TAG}_reference_benchmark.txt" 2> "$OUT/refbench.err"
python $REPO/tools/bench_receiver.py 8192 1280 128 0.1 > "$OUT/${TAG}_receiver_K8192.json" 2> "$OUT/recv.err"
python $REPO/tools/bench_one_object.py --devices 0 --blocks 8 > "$OUT/${TAG}_one_object_cfg5_8blocks_1gpu.json" 2> "$OUT/oneobj.err"
python $REPO/tools/bench_object_api.py 8192 1280 64 --json > "$OUT/${TAG}_object_api_K8192.json" 2> "$OUT/obj.err"
python $REPO/tools/bench_object_api.py 1000 1280 64 --json > "$OUT/${TAG}_object_api_K1000.json" 2>> "$OUT/obj.err"
ls -la "$OUT"
