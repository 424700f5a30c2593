# Runs ON THE GPU BOX: which workgroup shape small blocks want -- the single-wave variant (12 per CU) or the 256-thread one (4 per CU) --
# separately for encode launches (one plan) and decode launches (a plan per block): NRQ_TINY_DIV / NRQ_TINY_DIV_DEC sweeps
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'wg', d['detail']['encode']['wg_threads'], 'dec', round(d['detail']['decode_solve_ms'],2), 'wg', d['detail']['decode']['wg_threads'], 'planner', round(d['detail']['planner_ms'],2))"; }
for cfg in "12 6" "16 4" "24 3" "9 2"; do set -- $cfg
export NRQ_TINY_DIV=$1 NRQ_TINY_DIV_DEC=$2
echo "== tiny_div enc $1 dec $2"
echo -n "K256: "; run --K 256 --blocks 8192 --loss 0.06 --steps 6
echo -n "K350: "; run --K 350 --blocks 8192 --loss 0.06 --steps 6
echo -n "K500: "; run --K 500 --blocks 4096 --loss 0.06 --steps 6
echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6
echo -n "K2000: "; run --K 2000 --blocks 1024 --loss 0.06 --steps 6
echo -n "K3000: "; run --K 3000 --blocks 512 --loss 0.06 --steps 6
done
