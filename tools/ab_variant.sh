# Runs ON THE GPU BOX: same bench configs with the current library and with a variant (NANORQ_HIP_LIB), alternating
cd $GRAFT_REPO_ROOT
VAR=$1
IFS=";" read -ra CFG_LIST <<< "${CFGS:-500 1280 4096 0.06;1000 1280 2048 0.06}"   # CFGS="K T blocks loss;K T blocks loss"
for cfg in "${CFG_LIST[@]}"; do set -- $cfg
  for rep in 1 2 3; do
    for lib in cur var; do
      if [ $lib = var ]; then export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/nanorq_amd/$VAR; else unset NANORQ_HIP_LIB; fi
      timeout 300 python bench.py --K $1 --T $2 --blocks $3 --loss $4 --steps 5 --warmup 2 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); de=d['detail']
print('K=$1 $lib: %.1f Gbit/s %.2f ms/step enc %.2f dec %.2f' % (d['value'], d['ms_per_step'], de['encode_solve_ms'], de['decode_solve_ms']))"
    done
  done
done
