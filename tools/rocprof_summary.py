#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into small text files for profiles/.

  python tools/rocprof_summary.py stats  <kernel-trace results.db>     per-kernel launch statistics
  python tools/rocprof_summary.py pmc    <pmc results.db> [...]        per-kernel counter values
  python tools/rocprof_summary.py pmcjson <pmc results.db> [...]       solve-kernel counters per launch as JSON (bench.py reads
                                                                       profiles/r1_pmc_hbm.json for roofline.traffic)

Kernel names are shortened to the part before '('.  Durations in microseconds.
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "")[:70]


def stats(path, warm=2):
    """per (kernel, grid): all launches, and the STEADY launches -- without the first `warm` of the pair (the warm-up step of
    the profiled command: bench.py --warmup 1 launches the solve kernel twice, encode and decode, before its timed region; the
    cold first launch is the max column) -- whose average is what bench.py's roofline.avg_launch_ms measures with HIP events"""
    db = sqlite3.connect(path)
    rows = db.execute("select name, grid_x, duration, start from kernels order by start").fetchall()
    agg = defaultdict(list)
    for name, grid, dur, _ in rows:
        agg[(short(name), grid)].append(dur / 1000.0)
    total = sum(sum(v) for v in agg.values())
    print("# rocprofv3 --kernel-trace --stats: per (kernel, grid) launch statistics, microseconds; steady_* = without the first %d launches" % warm)
    print("%-64s %10s %6s %12s %10s %10s %10s %6s %8s %12s" % ("kernel", "grid_x", "calls", "total_us", "avg_us", "min_us", "max_us", "pct",
                                                             "steady_n", "steady_avg_us"))
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
        st = v[warm:] if len(v) > warm + 1 else v
        print("%-64s %10d %6d %12.1f %10.1f %10.1f %10.1f %6.2f %8d %12.1f" % (name, grid, len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                                            100.0 * sum(v) / total, len(st), sum(st) / len(st)))


def pmc(paths):
    print("# rocprofv3 --pmc: per (kernel, grid) counter values (KB for FETCH_SIZE / WRITE_SIZE as reported)")
    print("%-64s %10s %-12s %6s %14s %14s" % ("kernel", "grid", "counter", "calls", "avg_value", "max_value"))
    for path in paths:
        db = sqlite3.connect(path)
        rows = db.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
        agg = defaultdict(list)
        for name, grid, cname, val in rows:
            agg[(short(name), grid, cname)].append(val)
        for (name, grid, cname), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
            print("%-64s %10d %-12s %6d %14.1f %14.1f" % (name, grid, cname, len(v), sum(v) / len(v), max(v)))


def pmcjson(paths):
    """Counters of the full-batch nrq_solve_kernel launches, per launch.  Launches alternate encode, decode (bench.py)."""
    import json
    per = defaultdict(list)  # counter -> [value per dispatch, in dispatch order]
    grid_max = 0
    rows_all = []
    for path in paths:
        db = sqlite3.connect(path)
        rows = db.execute("select kernel_name, grid_size, counter_name, value, dispatch_id from counters_collection "
                          "order by dispatch_id").fetchall()
        rows = [r for r in rows if r[0].startswith("void nrq_solve_kernel") or r[0].startswith("nrq_solve_kernel")]
        rows_all += rows
        for r in rows:
            grid_max = max(grid_max, r[1])
    for name, grid, cname, val, did in rows_all:
        if grid == grid_max:
            per[cname].append(val)
    out = {"method": "rocprofv3 --pmc, one counter group per pass (tools/collect_profiles.sh) of `python bench.py --steps 3 --warmup 1 "
                     "--cpu-sample 0`; full-batch nrq_solve_kernel launches only; FETCH_SIZE / WRITE_SIZE are reported in KB (x1024 here); "
                     "MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 -- this kernel reads 16-byte row "
                     "segments (uncalibrated), so the read side is a lower bound, at most 2x higher",
           "K": 8192, "T": 1280, "blocks": 256,  # bench.py's default workload, which collect_profiles.sh runs
           "grid": grid_max, "launches": {k: len(v) for k, v in per.items()}, "per_launch": {}}
    for k, v in per.items():
        scale = 1024.0 if k in ("FETCH_SIZE", "WRITE_SIZE") else 1.0
        enc = [x * scale for x in v[0::2]]
        dec = [x * scale for x in v[1::2]]
        out["per_launch"][k] = {"encode": sum(enc) / max(1, len(enc)), "decode": sum(dec) / max(1, len(dec)),
                                "mean": sum(x * scale for x in v) / max(1, len(v))}
    if "FETCH_SIZE" in out["per_launch"] and "WRITE_SIZE" in out["per_launch"]:
        out["bytes_per_launch"] = out["per_launch"]["FETCH_SIZE"]["mean"] + out["per_launch"]["WRITE_SIZE"]["mean"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__" and sys.argv[1] != "timeline":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmcjson":
        pmcjson(sys.argv[2:])
    else:
        pmc(sys.argv[2:])


def timeline(path, last=40):
    """the last `last` kernel dispatches with start offsets: where the gaps between kernels are (last < 0: the last two steps of a
    bench run, found by their solve kernels)"""
    db = sqlite3.connect(path)
    try:
        rows = db.execute("select name, start, end, grid_x, queue_id from kernels order by start").fetchall()
    except sqlite3.Error:
        rows = [r + (0, 0) for r in db.execute("select name, start, end from kernels order by start").fetchall()]
    solves = [i for i, r in enumerate(rows) if "nrq_solve_kernel" in r[0]]
    if last < 0 and len(solves) >= 5:
        # the last two steps of the run: from the fourth-last solve launch (an encode) to the end of the last one, with what ran before it
        rows = rows[max(0, solves[-4] - 8):solves[-1] + 4]
    else:
        rows = rows[-abs(last):]
    t0 = rows[0][1]
    prev_end = t0
    print("# start_us  gap_before_us  dur_us  end_us  grid_x  queue  kernel")
    for name, s, e, gx, qid in rows:
        print("%10.1f %10.1f %10.1f %10.1f %9s %6s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, gx, qid, short(name)))
        prev_end = max(prev_end, e)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "timeline":
    timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
