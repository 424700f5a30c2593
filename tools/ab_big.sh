# Runs ON THE GPU BOX: the big-block configs (default settings: decode plans two steps ahead) + the planner's phase clocks
B="python bench.py --steps 6 --warmup 3 --cpu-sample 0 --pmc off --no-e2e"
run() { echo -n "$1: "; shift; env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), 'Gbit/s', round(d['ms_per_step'],2), 'ms; enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2), 'encplan', round(d['detail']['encode']['plan_ms'],1), 'enc host', round(d['detail']['encode']['host_ms'],1))"; }
run "cfg5" $B --K 56403 --blocks 8 --loss 0.2
run "cfg5 oh16" $B --K 56403 --blocks 8 --loss 0.2 --overhead 16
run "cfg4" $B --K 27000 --T 65504 --blocks 1 --loss 0.1
run "K50000" $B --K 50000 --blocks 16 --loss 0.06
run "K20000" $B --K 20000 --blocks 64 --loss 0.1
run "K10000" $B --K 10000 --blocks 256 --loss 0.06
NRQ_PROF=1 python bench.py --K 56403 --blocks 8 --loss 0.2 --steps 1 --warmup 1 --cpu-sample 0 --no-replan --pmc off --no-e2e --plan-ahead off 2>&1 | grep -E "planner nblk" | tail -1
NRQ_PROF=1 python bench.py --K 27000 --T 65504 --blocks 1 --loss 0.1 --steps 1 --warmup 1 --cpu-sample 0 --no-replan --pmc off --no-e2e --plan-ahead off 2>&1 | grep -E "planner nblk" | tail -1
NRQ_PROF=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --pmc off --no-e2e 2>&1 | grep -E "planner nblk" | tail -1
for i in 1 2; do python bench.py --steps 10 --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('headline', round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2))"; done
python bench.py --K 1000 --blocks 2048 --loss 0.06 --steps 6 --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('K1000', round(d['value'],1), round(d['ms_per_step'],2), 'planner', round(d['detail']['planner_ms'],2))"
