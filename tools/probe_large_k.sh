# Runs ON THE GPU BOX: where the time of the large-K configs goes (planner / host plan / solve), NRQ_PROF marks included.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "56403 1280 8 0.2" "27000 65504 1 0.1" "50000 1280 16 0.06" "20000 1280 64 0.1"; do set -- $cfg
  echo "== K=$1 T=$2 blocks=$3 loss=$4"
  NRQ_PROF=1 timeout 600 python bench.py --K $1 --T $2 --blocks $3 --loss $4 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "NRQ_PROF|metric" | cut -c1-1800 | tail -8
  echo "-- no-replan"
  timeout 600 python bench.py --K $1 --T $2 --blocks $3 --loss $4 --steps 2 --warmup 1 --cpu-sample 0 --no-replan 2>&1 | tail -1 | cut -c1-1800
done
