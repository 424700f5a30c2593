# Runs ON THE GPU BOX: bench lines of several configs for the current library and the named variants
#   CFGS="K T blocks loss;..." bash tools/ab_list.sh var1 var2
cd $GRAFT_REPO_ROOT
IFS=";" read -ra CFG_LIST <<< "${CFGS:-1000 1280 2048 0.06}"
for cfg in "${CFG_LIST[@]}"; do set -- $cfg; K=$1; T=$2; B=$3; L=$4
  for v in cur ${VARS}; do
    if [ $v = cur ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/nanorq_amd/build_var_$v.so; fi
    timeout 300 python bench.py --K $K --T $T --blocks $B --loss $L --steps 6 --warmup 2 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); de=d['detail']
print('K=$K %-6s %.1f Gbit/s %.2f ms/step enc %.2f dec %.2f planner %.2f' % ('$v', d['value'], d['ms_per_step'], de['encode_solve_ms'], de['decode_solve_ms'], de.get('planner_ms') or 0))"
  done
done
