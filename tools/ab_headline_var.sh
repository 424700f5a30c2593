# Runs ON THE GPU BOX: headline solve-kernel times for variant libraries (build_var/lib_*.so), A/B A/B
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --cpu-sample 0 --pmc off --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), 'enc', round(d['detail']['encode_solve_ms'],3), 'dec', round(d['detail']['decode_solve_ms'],3))"; }
for rep in 1 2; do
for lib in ${LIBS:-default}; do
  if [ "$lib" = default ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/build_var/lib_$lib.so; fi
  echo -n "$lib: "; run --steps 10
done
done
