export TMPDIR=/tmp; cd /tmp
for v in base var_ldsbar; do
  if [ $v = base ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/nanorq_amd/$v.so; fi
  rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 > /tmp/tr.log 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats $(find /tmp/tr -name '*.db' | head -1) | head -8
done
