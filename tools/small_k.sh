# Runs ON THE GPU BOX: small-block configs, planner with small / full-size workgroup state
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "100 1024 8192" "500 1280 4096" "1000 1280 2048" "2000 1280 1024"; do set -- $cfg
  for mode in small big; do
    if [ $mode = big ]; then export NRQ_PLAN_BIG_STATE=1; else unset NRQ_PLAN_BIG_STATE; fi
    timeout 300 python bench.py --K $1 --T $2 --blocks $3 --loss 0.06 --steps 5 --warmup 2 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); de=d['detail']
print('K=$1 $mode: %.1f Gbit/s %.2f ms/step enc %.2f dec %.2f other %.2f host_planned %s' % (d['value'], d['ms_per_step'], de['encode_solve_ms'], de['decode_solve_ms'], d['ms_per_step']-de['encode_solve_ms']-de['decode_solve_ms'], d['config']['host_planned_blocks']))"
  done
done
