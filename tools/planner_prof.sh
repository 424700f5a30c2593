# Runs ON THE GPU BOX: planner clocks (NRQ_PROF) and bench lines for the headline and cfg5
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
NRQ_PROF=1 python bench.py --steps 1 --warmup 1 --pmc off --no-e2e --cpu-sample 0 2>&1 | grep "planner nblk" | tail -1
NRQ_PROF=1 python bench.py --K 56403 --blocks 8 --loss 0.2 --steps 1 --warmup 1 --cpu-sample 0 --no-replan --pmc off --no-e2e 2>&1 | grep -E "planner nblk" | tail -1
for i in 1 2; do python bench.py --pmc off --no-e2e --cpu-sample 0 2>&1 | tail -1 | cut -c1-130; done
python bench.py --K 56403 --blocks 8 --loss 0.2 --steps 4 --warmup 2 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | cut -c1-130
python bench.py --K 1000 --blocks 2048 --loss 0.06 --steps 4 --warmup 2 --cpu-sample 0 --pmc off --no-e2e 2>&1 | tail -1 | cut -c1-130
