# Runs ON THE GPU BOX: per-phase clocks of the solve kernel for small blocks
cd $GRAFT_REPO_ROOT
for cfg in "100 1024 8192 0.06" "1000 1280 2048 0.06" "500 1280 4096 0.06"; do set -- $cfg
  echo "== K=$1 T=$2 blocks=$3"
  NRQ_PROF=1 timeout 300 python bench.py --K $1 --T $2 --blocks $3 --loss $4 --steps 1 --warmup 1 --cpu-sample 0 --pmc off --no-e2e 2>&1 | grep NRQ_PROF | tail -3 | cut -c1-400
  timeout 300 python bench.py --K $1 --T $2 --blocks $3 --loss $4 --steps 3 --warmup 1 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); de=d['detail']
print('   %.1f Gbit/s  %.2f ms/step | enc solve %.2f dec solve %.2f ms | dec plan_ms %.2f | lds %s grid %s wg %s sps %s' % (d['value'], d['ms_per_step'], de['encode_solve_ms'], de['decode_solve_ms'], de['decode']['plan_ms'], de['decode']['lds_bytes'], de['decode']['grid'], de['decode']['wg_threads'], de['decode']['strips_per_slot']))"
done
