#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the BASELINE configs other than the default one (cfg1, cfg2, cfg4, cfg5 and the K values of
# the reference's chart) through bench.py, one JSON line each into gpurun_out/cfg/ -- copy what is to be judged to profiles/.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/bench_configs.sh [tag]'
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r6}
OUT=$REPO/gpurun_out/cfg
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
run() { # name K T blocks loss overhead [cpu-sample] [further bench.py arguments]      (ONLY=regex in the environment: just the matching lines)
  local name=$1 K=$2 T=$3 B=$4 P=$5 OH=$6 CS=${7:-2}
  if [ -n "$ONLY" ] && ! [[ "$name" =~ $ONLY ]]; then return 0; fi
  shift 7 2>/dev/null || shift $#
  timeout 900 python $REPO/bench.py --K $K --T $T --blocks $B --loss $P --overhead $OH --steps 5 --warmup 2 --cpu-sample $CS "$@" > "$OUT/${TAG}_bench_$name.json" 2> "$OUT/${TAG}_bench_$name.err"
  set -- $name
  python3 - "$OUT/${TAG}_bench_$1.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    b = (d.get("roofline") or {}).get("binding") or {}
    r3 = lambda x: None if x is None else round(x, 3)
    w = d["config"].get("decode_launch_widths") or {}
    print("%-22s %-28s %8.1f Gbit/s  %7.2f ms/step  solves %s / %s ms planner %s ms | lds_frac %s conflict %s issue %s hbm %s clock %s GHz | decode widths %s second lists %s | e2e %s" % (
        sys.argv[1].split("_bench_")[-1][:-5], d["metric"].split(",")[1].strip(), d["value"], d["ms_per_step"], r3(d["detail"]["encode_solve_ms"]), r3(d["detail"]["decode_solve_ms"]),
        r3(d["detail"]["planner_ms"]), r3(b.get("lds_frac")), r3(b.get("lds_bank_conflict_share")), r3(b.get("issue_frac")), r3(b.get("hbm_frac")), r3(b.get("clock_ghz")),
        w.get("widths"), w.get("second_list_launches"), r3((d.get("e2e") or {}).get("value"))))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run cfg1_K100_T1024      100  1024 8192 0.06 0
run cfg2_K1024_T1280    1024  1280 2048 0.05 0
run cfg4_K27000_T65504 27000 65504    1 0.10 0 0
run cfg5_K56403_T1280  56403  1280    8 0.20 0 1
# the XOR-only path (overhead >= H: no GF(256) work, reference precode.c:362-363) and the small-overhead variants of SURVEY 8(d)
run cfg2_K1024_oh52     1024  1280 2048 0.06 52
run cfg3_K8192_oh2      8192  1280  256 0.10 2
# heavier loss (reference lib/precode.c:176-203: u grows with the loss): the blocks whose image outgrows the 16-byte strip run as a second list
run cfg3_K8192_loss30   8192  1280  256 0.30 0 0 --patterns 8 --steps 16
run cfg3_K8192_loss30_one_list 8192 1280 256 0.30 0 0 --patterns 8 --steps 16 --one-list
run cfg5_K56403_oh16   56403  1280    8 0.20 16 1
run K1000_T1280         1000  1280 2048 0.06 0
run K256_T1280           256  1280 8192 0.06 0
run K500_T1280           500  1280 4096 0.06 0
run K5000_T1280         5000  1280  512 0.06 0
run K9000_T1280         9000  1280  256 0.06 0
run K10000_T1280       10000  1280  256 0.06 0
run K10000_T1280_8byte 10000  1280  256 0.06 0 0 --no-wb12
run K11000_T1280       11000  1280  256 0.06 0
run K20000_T1280       20000  1280   64 0.10 0 1
run K50000_T1280       50000  1280   16 0.06 0 1
