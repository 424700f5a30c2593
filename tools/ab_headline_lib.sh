# Runs ON THE GPU BOX: the headline (and two neighbours that use the big workgroup) with variant libraries
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],2), 'dec', round(d['detail']['decode_solve_ms'],2), 'planner', round(d['detail']['planner_ms'],2), 'wg', d['detail']['encode']['wg_threads'])"; }
for lib in ${LIBS:-default}; do
  if [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$lib; fi
  echo "== lib: $lib"
  echo -n "headline: "; run --steps 10
  echo -n "headline: "; run --steps 10
  echo -n "K5000: "; run --K 5000 --blocks 512 --loss 0.06 --steps 6
  echo -n "K10000: "; run --K 10000 --blocks 256 --loss 0.06 --steps 6
done
