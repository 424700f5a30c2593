# Runs ON THE GPU BOX: kernel timeline of the last step for a config, small vs big planner state
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in small big; do
  if [ $mode = big ]; then export NRQ_PLAN_BIG_STATE=1; else unset NRQ_PLAN_BIG_STATE; fi
  rm -rf $R/gpurun_out/tl_$mode
  rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$mode -- python $R/bench.py --K $1 --T $2 --blocks $3 --loss 0.06 --steps 3 --warmup 1 --cpu-sample 0 --pmc off --no-e2e > /dev/null 2>&1
  echo "== $mode"
  python $R/tools/rocprof_summary.py timeline $(find $R/gpurun_out/tl_$mode -name "*.db" | head -1) 60 | grep -E "nrq_|index_fill" | tail -12
  find $R/gpurun_out/tl_$mode -name "*.db" -delete
done
