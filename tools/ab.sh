#!/bin/bash
# Runs ON THE GPU BOX (gpurun): A/B of library variants and environment switches over bench.py configurations, alternating
# so that box-to-box and minute-to-minute drift hits both sides alike.  One line per run.
#   VARIANTS="label[:VAR=val,VAR=val...];label2:..."   (label "cur" = the library as built; NANORQ_HIP_LIB=... picks a variant .so,
#                                                        paths relative to the repository root)
#   CFGS="K T blocks loss [extra bench.py args];..."    (default: the headline)
#   REPS=n (default 2)   STEPS=n (default 6)
#   gpurun --timeout 900 -- 'VARIANTS="cur;dual:NRQ_DUAL=1" CFGS="8192 1280 256 0.1;10000 1280 256 0.06" bash tools/ab.sh'
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
IFS=";" read -ra VAR_LIST <<< "${VARIANTS:-cur}"
IFS=";" read -ra CFG_LIST <<< "${CFGS:-8192 1280 256 0.1}"
for cfg in "${CFG_LIST[@]}"; do
  set -- $cfg; K=$1; T=$2; B=$3; P=$4; shift 4
  for rep in $(seq 1 ${REPS:-2}); do
    for var in "${VAR_LIST[@]}"; do
      label=${var%%:*}; envs=""
      [[ "$var" == *:* ]] && envs=${var#*:}
      (
        IFS=","; for kv in $envs; do
          case "$kv" in NANORQ_HIP_LIB=*) export NANORQ_HIP_LIB=$REPO/${kv#NANORQ_HIP_LIB=};; *) export "$kv";; esac
        done; unset IFS
        timeout 600 python bench.py --K $K --T $T --blocks $B --loss $P --steps ${STEPS:-6} --warmup 2 --cpu-sample 0 --alg-sample 0 --pmc off --no-e2e "$@" 2> /tmp/ab.err | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); de = d['detail']; e = de['encode']; dc = de['decode']
    print('K=$K T=$T x$B %-10s %7.1f Gbit/s %6.2f ms/step | enc %.2f dec %.2f planner %.2f ms | WB %s/%s wg %s/%s grid %s/%s' % ('$label', d['value'], d['ms_per_step'],
          de['encode_solve_ms'] or 0, de['decode_solve_ms'] or 0, de['planner_ms'] or 0, e['strip_bytes'], dc['strip_bytes'], e['wg_threads'], dc['wg_threads'], e['grid'], dc['grid']))
except Exception as ex:
    print('K=$K T=$T x$B $label FAILED', ex, open('/tmp/ab.err').read()[-300:])"
      )
    done
  done
done
