# Runs ON THE GPU BOX: peeling-round variants of the planner (build_var/lib_*.so) -- planner phase clocks and bench lines
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'planner', round(d['detail']['planner_ms'],2))"; }
for lib in ${LIBS:-default}; do
  if [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/build_var/lib_$lib.so; fi
  echo "== lib: $lib"
  NRQ_PROF=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --pmc off --no-e2e --plan-ahead off 2>&1 | grep -E "planner nblk" | tail -1 | cut -c1-200
  echo -n "headline: "; run --steps 10
  echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6
  echo -n "K5000: "; run --K 5000 --blocks 400 --steps 6
done
