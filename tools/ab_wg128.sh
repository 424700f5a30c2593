# Runs ON THE GPU BOX: 128-thread planner workgroups for the smallest blocks against NRQ_PLAN_NO_WG128=1 (256 threads)
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'planner', round(d['detail']['planner_ms'],2))"; }
for rep in 1 2; do
for v in wg128; do
  if [ $v = wg256 ]; then export NRQ_PLAN_NO_WG128=1; else unset NRQ_PLAN_NO_WG128; fi
  echo "== $v"
  echo -n "K100: "; run --K 100 --T 1024 --blocks 8192 --loss 0.06 --steps 6
  echo -n "K256: "; run --K 256 --blocks 8192 --loss 0.06 --steps 6
  echo -n "K500: "; run --K 500 --blocks 4096 --loss 0.06 --steps 6
  echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6
  echo -n "K2000: "; run --K 2000 --blocks 1024 --loss 0.06 --steps 6
  echo -n "K4000: "; run --K 4000 --blocks 512 --loss 0.06 --steps 6
done
done
