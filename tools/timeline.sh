# Runs ON THE GPU BOX: kernel timeline (three queues) of the last steps of a big-block config, plan-ahead off / depth 1 / 2
#   bash tools/timeline.sh K T blocks loss
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tlb
for pa in off 1 2; do
  rm -rf /tmp/tlb_$pa
  if [ $pa = off ]; then A="--plan-ahead off"; else A="--plan-ahead on --plan-ahead-depth $pa"; fi
  rocprofv3 --kernel-trace -d /tmp/tlb_$pa -- python $R/bench.py --K $1 --T $2 --blocks $3 --loss $4 $A --steps 4 --warmup 2 --cpu-sample 0 --pmc off --no-e2e > $R/gpurun_out/tlb/bench_$pa.json 2> /dev/null
  python $R/tools/rocprof_summary.py timeline $(find /tmp/tlb_$pa -name "*.db" | head -1) -1 > $R/gpurun_out/tlb/timeline_$pa.txt
  rm -rf /tmp/tlb_$pa
done
