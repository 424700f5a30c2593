# Runs ON THE GPU BOX: plans issued a step ahead at the headline and for small blocks (the planner stays on the solve stream: what
# goes is the host's wait in the middle of the step)
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2))"; }
for pa in "off" "on --plan-ahead-depth 1" "on --plan-ahead-depth 2"; do
  echo "== plan-ahead $pa"
  echo -n "headline: "; run --steps 10 --plan-ahead $pa
  echo -n "headline: "; run --steps 10 --plan-ahead $pa
  echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6 --plan-ahead $pa
  echo -n "K100: "; run --K 100 --T 1024 --blocks 8192 --loss 0.06 --steps 6 --plan-ahead $pa
  echo -n "K5000: "; run --K 5000 --blocks 512 --loss 0.06 --steps 6 --plan-ahead $pa
done
