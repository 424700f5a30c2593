# Runs ON THE GPU BOX: small-block configs with variant libraries
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2), 'wg', d['detail']['encode']['wg_threads'], 'grid', d['detail']['encode']['grid'])"; }
for lib in ${LIBS:-default}; do
  if [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$lib; fi
  echo "== lib: $lib"
  echo -n "K100 T1024: "; run --K 100 --T 1024 --blocks 8192 --loss 0.06 --steps 6
  echo -n "K256: "; run --K 256 --blocks 8192 --loss 0.06 --steps 6
  echo -n "K500: "; run --K 500 --blocks 4096 --loss 0.06 --steps 6
  echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6
  echo -n "K2000: "; run --K 2000 --blocks 1024 --loss 0.06 --steps 6
done
