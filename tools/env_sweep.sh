# Runs ON THE GPU BOX: bench lines of configs under different environment settings:  CFGS="K T blocks loss;..." ENVS="A=1;B=2 C=3" bash tools/env_sweep.sh
cd $GRAFT_REPO_ROOT
IFS=";" read -ra CFG_LIST <<< "${CFGS}"
IFS=";" read -ra ENV_LIST <<< "${ENVS}"
for cfg in "${CFG_LIST[@]}"; do set -- $cfg
  for e in "${ENV_LIST[@]}"; do
    env $e timeout 300 python bench.py --K $1 --T $2 --blocks $3 --loss $4 --steps 6 --warmup 2 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); de=d['detail']
print('K=$1 %-22s %.1f Gbit/s %.2f ms/step enc %.2f dec %.2f planner %.2f' % ('$e', d['value'], d['ms_per_step'], de['encode_solve_ms'], de['decode_solve_ms'], de.get('planner_ms') or 0))"
  done
done
