# Runs ON THE GPU BOX: NRQ_PROF phase clocks + bench line of the headline for the current library and the variants named
cd $GRAFT_REPO_ROOT
for v in cur "$@"; do
  if [ $v = cur ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/nanorq_amd/build_var_$v.so; fi
  echo "== $v"
  NRQ_PROF=1 NRQ_PROF_BASE=1 timeout 300 python bench.py --steps 2 --warmup 1 --pmc off --no-e2e --cpu-sample 0 ${BENCH_ARGS:-} 2>&1 | grep NRQ_PROF | grep -v planner | tail -2 | cut -c1-400
  timeout 300 python bench.py --steps 10 --warmup 3 --pmc off --no-e2e --cpu-sample 0 ${BENCH_ARGS:-} | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); de=d['detail']
print('%.1f Gbit/s %.2f ms/step enc %.2f dec %.2f' % (d['value'], d['ms_per_step'], de['encode_solve_ms'], de['decode_solve_ms']))"
done
