# Runs ON THE GPU BOX: planner variants (build_var/lib_*.so) at the block sizes whose peeling state is out of LDS
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'planner', round(d['detail']['planner_ms'],2))"; }
for lib in ${LIBS:-default}; do
  if [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/build_var/lib_$lib.so; fi
  echo "== lib: $lib"
  NRQ_PROF=1 python bench.py --K 56403 --blocks 8 --loss 0.2 --steps 1 --warmup 1 --cpu-sample 0 --no-replan --pmc off --no-e2e --plan-ahead off 2>&1 | grep -E "planner nblk" | tail -1 | cut -c1-260
  echo -n "headline: "; run --steps 10
  echo -n "K10000: "; run --K 10000 --blocks 256 --loss 0.06 --steps 6
  echo -n "K20000: "; run --K 20000 --blocks 64 --loss 0.1 --steps 6
  echo -n "cfg4: "; run --K 27000 --T 65504 --blocks 1 --loss 0.1 --steps 4 --warmup 2
  echo -n "cfg5: "; run --K 56403 --blocks 8 --loss 0.2 --steps 6 --warmup 3
  echo -n "cfg5 plans in the call: "; run --K 56403 --blocks 8 --loss 0.2 --steps 6 --warmup 3 --plan-ahead off
done
