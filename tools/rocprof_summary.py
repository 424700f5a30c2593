#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into small text files for profiles/.

  python tools/rocprof_summary.py stats  <kernel-trace results.db>     per-kernel launch statistics
  python tools/rocprof_summary.py pmc    <pmc results.db> [...]        per-kernel counter values

Kernel names are shortened to the part before '('.  Durations in microseconds.
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "")[:70]


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, grid_x, duration from kernels").fetchall()
    agg = defaultdict(list)
    for name, grid, dur in rows:
        agg[(short(name), grid)].append(dur / 1000.0)
    total = sum(sum(v) for v in agg.values())
    print("# rocprofv3 --kernel-trace --stats: per (kernel, grid) launch statistics, microseconds")
    print("%-64s %10s %6s %12s %10s %10s %10s %6s" % ("kernel", "grid_x", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print("%-64s %10d %6d %12.1f %10.1f %10.1f %10.1f %6.2f" % (name, grid, len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                                  100.0 * sum(v) / total))


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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
