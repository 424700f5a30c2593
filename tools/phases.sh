#!/bin/bash
# Runs ON THE GPU BOX (gpurun): per-phase shader clocks of the solve kernel (NRQ_PROF=1: load / fwd / hdpc / bin / dense / tables /
# backsub / store per sampled strip, encode launch then decode launch) and of the planner, for library variants / environment
# switches (VARIANTS, as in tools/ab.sh) over configurations (CFGS="K T blocks loss;...").
#   gpurun --timeout 600 -- 'VARIANTS="cur;dual:NRQ_DUAL=1" bash tools/phases.sh'
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
IFS=";" read -ra VAR_LIST <<< "${VARIANTS:-cur}"
IFS=";" read -ra CFG_LIST <<< "${CFGS:-8192 1280 256 0.1}"
for cfg in "${CFG_LIST[@]}"; do
  set -- $cfg; K=$1; T=$2; B=$3; P=$4; shift 4
  for var in "${VAR_LIST[@]}"; do
    label=${var%%:*}; envs=""
    [[ "$var" == *:* ]] && envs=${var#*:}
    (
      IFS=","; for kv in $envs; do
        case "$kv" in NANORQ_HIP_LIB=*) export NANORQ_HIP_LIB=$REPO/${kv#NANORQ_HIP_LIB=};; *) export "$kv";; esac
      done; unset IFS
      echo "== K=$K T=$T x$B $label"
      NRQ_PROF=1 NRQ_PROF_BASE=${PROF_BASE:-2} timeout 600 python bench.py --K $K --T $T --blocks $B --loss $P --steps 1 --warmup 1 --cpu-sample 0 --alg-sample 0 --pmc off --no-e2e "$@" 2>&1 | grep NRQ_PROF | tail -${LINES_:-3} | cut -c1-420
    )
  done
done
