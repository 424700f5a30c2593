# Runs ON THE GPU BOX: the headline and cfg5 planner with variant libraries (NANORQ_HIP_LIB)
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2))"; }
for lib in ${LIBS:-""}; do
  if [ -z "$lib" ] || [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$lib; fi
  echo "== lib: ${lib:-default}"
  echo -n "headline: "; run --steps 10
  echo -n "cfg5 off: "; run --K 56403 --blocks 8 --loss 0.2 --steps 5 --plan-ahead off
  echo -n "cfg5 on2: "; run --K 56403 --blocks 8 --loss 0.2 --steps 5
  echo -n "K1000: "; run --K 1000 --blocks 2048 --loss 0.06 --steps 6
done
