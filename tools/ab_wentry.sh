# Runs ON THE GPU BOX: the entry pass of big blocks' planner by nrq_wentry_kernel (default) against the planner workgroup itself (NRQ_NO_WENTRY=1)
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2), 'encplan', round(d['detail']['encode']['plan_ms'],1))"; }
for w in 1 0; do
  if [ $w = 1 ]; then export NRQ_NO_WENTRY=1; else unset NRQ_NO_WENTRY; fi
  echo "== NRQ_NO_WENTRY=$w"
  NRQ_PROF=1 python bench.py --K 56403 --blocks 8 --loss 0.2 --steps 1 --warmup 1 --cpu-sample 0 --no-replan --pmc off --no-e2e --plan-ahead off 2>&1 | grep -E "planner nblk" | tail -1
  echo -n "cfg5: "; run --K 56403 --blocks 8 --loss 0.2 --steps 6 --warmup 3
  echo -n "cfg5 off: "; run --K 56403 --blocks 8 --loss 0.2 --steps 6 --warmup 3 --plan-ahead off
  echo -n "cfg4: "; run --K 27000 --T 65504 --blocks 1 --loss 0.1 --steps 6 --warmup 2
  echo -n "K50000: "; run --K 50000 --blocks 16 --loss 0.06 --steps 6 --warmup 3
  echo -n "K20000: "; run --K 20000 --blocks 64 --loss 0.1 --steps 6 --warmup 2
done
