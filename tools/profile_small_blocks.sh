export TMPDIR=/tmp; cd /tmp
for cfg in "--K 100 --T 1024 --blocks 8192" "--K 1024 --T 1280 --blocks 2048"; do
  rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py $cfg --steps 4 --warmup 1 --cpu-sample 0 > /tmp/tr.log 2>&1
  echo "== $cfg"; tail -1 /tmp/tr.log | cut -c1-250; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats $(find /tmp/tr -name '*.db' | head -1) | head -9
done
