# Runs ON THE GPU BOX: planner_ms and throughput over the block sizes, for variant libraries (build_var/lib_*.so)
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'planner', round(d['detail']['planner_ms'],2))"; }
for lib in ${LIBS:-default}; do
  if [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/build_var/lib_$lib.so; fi
  echo "== lib: $lib"
  echo -n "K100: "; run --K 100 --T 1024 --blocks 8192 --loss 0.06 --steps 6
  echo -n "K500: "; run --K 500 --blocks 4096 --loss 0.06 --steps 6
  echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6
  echo -n "headline: "; run --steps 10
  echo -n "K10000: "; run --K 10000 --blocks 256 --loss 0.06 --steps 6
  echo -n "cfg5 plans in the call: "; run --K 56403 --blocks 8 --loss 0.2 --steps 6 --warmup 3 --plan-ahead off
done
