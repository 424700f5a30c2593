B="python bench.py --steps 6 --warmup 2 --cpu-sample 0 --pmc off --no-e2e --plan-ahead off"
run() { echo -n "$1: "; shift; env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2))"; }
run "cfg4 base" $B --K 27000 --T 65504 --blocks 1 --loss 0.1
run "cfg4 nocheck" $B --K 27000 --T 65504 --blocks 1 --loss 0.1 --check-blocks 0
run "cfg4 reserve16" NRQ_RESERVE_CUS=16 $B --K 27000 --T 65504 --blocks 1 --loss 0.1
run "cfg4 noreplan" $B --K 27000 --T 65504 --blocks 1 --loss 0.1 --no-replan
run "K20000 base" $B --K 20000 --blocks 64 --loss 0.1
run "K20000 nocheck" $B --K 20000 --blocks 64 --loss 0.1 --check-blocks 0
run "K20000 noreplan" $B --K 20000 --blocks 64 --loss 0.1 --no-replan
run "K10000 base" $B --K 10000 --blocks 256 --loss 0.06
run "K10000 nocheck" $B --K 10000 --blocks 256 --loss 0.06 --check-blocks 0
