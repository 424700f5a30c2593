cd $GRAFT_REPO_ROOT
NRQ_PROF=1 timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep "NRQ_PROF" | head -12
echo "== cfg2"; timeout 200 python bench.py --K 1024 --blocks 2048 --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | cut -c1-200
echo "== cfg1"; timeout 200 python bench.py --K 100 --T 1024 --blocks 8192 --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | cut -c1-200
echo "== cfg4"; timeout 300 python bench.py --K 27000 --T 65504 --blocks 1 --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | cut -c1-200
echo "== cfg5"; timeout 300 python bench.py --K 56403 --T 1280 --blocks 8 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | cut -c1-200
echo "== hostplanner"; NRQ_HOST_PLANNER=1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | cut -c1-200
echo "== overhead2"; timeout 200 python bench.py --overhead 2 --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | cut -c1-200
