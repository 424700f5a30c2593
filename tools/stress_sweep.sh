#!/bin/bash
# Runs ON THE GPU BOX (gpurun): tools/stress_decode.py over the block sizes each planner form covers (state in LDS with the
# chained peel, the dense stage over the rowstate image, the compact state, the segmented run, the small-block workgroups).
#   gpurun --timeout 900 -- 'bash tools/stress_sweep.sh > gpurun_out/prof/r5_stress_decode.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-$PWD}
while read -r args; do
  echo "== K T blocks loss iterations: $args"
  timeout 600 python tools/stress_decode.py $args 2>&1 | tail -2
done <<'CASES'
8192 64 256 0.1 60
8192 64 256 0.3 30
9400 32 256 0.1 30
2000 32 1024 0.2 40
5000 32 512 0.06 40
20000 16 64 0.1 15
56403 8 8 0.2 10
56403 8 8 0.45 6
700 32 2048 0.1 30
1000 32 2048 0.5 20
100 32 8192 0.2 20
10 32 8192 0.3 20
11000 36 128 0.2 15
10000 44 128 0.06 20
9000 50 128 0.1 15
CASES
