#!/bin/bash
# Runs ON THE GPU BOX: the default bench line into gpurun_out/r3/<name>.json and a one-screen summary of it.
#   gpurun -- 'bash tools/bench_line.sh name [bench.py args]'
REPO=${GRAFT_REPO_ROOT:-$PWD}
NAME=${1:-bench}; shift
mkdir -p $REPO/gpurun_out/r3
cd /tmp; export TMPDIR=/tmp
timeout 1500 python $REPO/bench.py "$@" > $REPO/gpurun_out/r3/$NAME.json 2> $REPO/gpurun_out/r3/$NAME.err
tail -3 $REPO/gpurun_out/r3/$NAME.err
python3 - $REPO/gpurun_out/r3/$NAME.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r = d.get("roofline") or {}
print("value %.1f Gbit/s  %.2f ms/step | bound %s frac %s frac_physical %s avg_launch_ms %s" % (d["value"], d["ms_per_step"], r.get("bound"), r.get("frac"), r.get("frac_physical"), r.get("avg_launch_ms")))
print("binding", r.get("binding"))
print("traffic", r.get("traffic"), "compulsory", r.get("compulsory"), "ratio", r.get("traffic_over_compulsory"))
print("e2e", d.get("e2e"))
c = d.get("cpu_baseline") or {}
print("cpu", c.get("value"), "precalc", c.get("precalc"), "all", c.get("all_cores"))
print("detail", {k: d["detail"][k] for k in ("planner_ms", "encode_solve_ms", "decode_solve_ms")}, d["config"].get("host_planned_blocks"), d["check"])
PY
