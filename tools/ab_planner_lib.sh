# Runs ON THE GPU BOX: planner phase clocks and timings with variant libraries
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2), 'encplan', round(d['detail']['encode']['plan_ms'],1))"; }
for lib in ${LIBS:-default}; do
  if [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$lib; fi
  echo "== lib: $lib"
  NRQ_PROF=1 python bench.py --K 56403 --blocks 8 --loss 0.2 --steps 1 --warmup 1 --cpu-sample 0 --no-replan --pmc off --no-e2e --plan-ahead off 2>&1 | grep -E "planner nblk" | tail -1
  NRQ_PROF=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --pmc off --no-e2e 2>&1 | grep -E "planner nblk" | tail -1
  echo -n "headline: "; run --steps 10
  echo -n "cfg5: "; run --K 56403 --blocks 8 --loss 0.2 --steps 6 --warmup 3
  echo -n "cfg5 off: "; run --K 56403 --blocks 8 --loss 0.2 --steps 6 --warmup 3 --plan-ahead off
  echo -n "K50000: "; run --K 50000 --blocks 16 --loss 0.06 --steps 6 --warmup 3
  echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6
done
