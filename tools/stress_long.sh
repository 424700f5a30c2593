#!/bin/bash
# Runs ON THE GPU BOX (gpurun): tools/stress_decode.py at length -- ~3 million decodes over the sizes of every planner form
# (chained peel with the state in LDS, dense stage over the rowstate image, compact state, small workgroups), every block compared
# with its source.  ~2.5 minutes.
#   gpurun --timeout 2700 -- 'bash tools/stress_long.sh' > profiles/r6_stress_long.txt
cd ${GRAFT_REPO_ROOT:-$PWD}
while read -r args; do
  out=$(timeout 900 python tools/stress_decode.py $args 2>&1 | grep -v amdgpu)
  echo "== $args: $(echo "$out" | tail -1); blocks sent to the host planner (capacity or plan check): $(echo "$out" | grep -c "host planner took")"
done <<'CASES'
8192 32 256 0.1 200
8192 32 256 0.45 60
9400 32 256 0.2 80
10500 32 128 0.1 60
12000 32 128 0.1 40
6500 32 256 0.15 100
4000 32 512 0.1 100
3000 32 512 0.3 100
2000 32 1024 0.2 150
1000 32 2048 0.1 100
700 32 2048 0.1 150
500 32 4096 0.3 60
300 32 4096 0.15 60
100 32 8192 0.2 60
26 32 8192 0.3 40
CASES
