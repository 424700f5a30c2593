# A/B of library variants on small-block configs (run on the GPU box): tools/ab_small.sh var1 var2 ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset NANORQ_HIP_LIB; else export NANORQ_HIP_LIB=$GRAFT_REPO_ROOT/nanorq_amd/$v.so; fi
  line="$v:"
  for cfg in "100 1024 8192" "256 1280 4096" "512 1280 4096" "1024 1280 2048" "1500 1280 1024" "2048 1280 1024" "3000 1280 512"; do
    set -- $cfg
    a=$(timeout 200 python bench.py --K $1 --T $2 --blocks $3 --steps 4 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    line="$line K=$1 $a |"
  done
  echo "$line"
done
