#!/usr/bin/env python3
"""Differences the rocprofv3 outputs of the phase-truncated solve-kernel builds (tools/phase_counters.sh) into a per-phase table:
duration, LDS pipeline cycles, bank-conflict cycles, VALU / LDS / SALU instructions, waiting cycles -- encode and decode launch."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

OUT = sys.argv[1]
PHASES = ["load", "forward", "hdpc", "gf2-combine", "dense", "tables", "backsub", "store"]
VARS = ["0", "1", "2", "3", "4", "5", "6", "full"]
NCU, NSE = 256, 32


def solve_rows(db, q):
    return [r for r in db.execute(q).fetchall() if "nrq_solve_kernel" in r[0]]


def durations(v):
    f = glob.glob(os.path.join(OUT, "tr_" + v, "**", "*.db"), recursive=True)
    if not f:
        return None
    rows = solve_rows(sqlite3.connect(f[0]), "select name, duration, start from kernels order by start")
    d = [r[1] / 1000.0 for r in rows]
    d = d[2:] if len(d) > 3 else d      # (the first pair is cold)
    return {"enc": sum(d[0::2]) / max(1, len(d[0::2])), "dec": sum(d[1::2]) / max(1, len(d[1::2]))}


def counters(v):
    f = glob.glob(os.path.join(OUT, "pmc_" + v, "**", "*.db"), recursive=True)
    if not f:
        return None
    rows = solve_rows(sqlite3.connect(f[0]), "select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id")
    per = defaultdict(dict)
    for _, cname, val, did in rows:
        per[did][cname] = per[did].get(cname, 0.0) + val
    ids = sorted(per)
    ids = ids[2:] if len(ids) > 3 else ids
    out = {"enc": defaultdict(float), "dec": defaultdict(float)}
    for i, did in enumerate(ids):
        for k, x in per[did].items():
            out["enc" if i % 2 == 0 else "dec"][k] += x / max(1, len(ids[i % 2::2]))
    return out


dur = {v: durations(v) for v in VARS}
ctr = {v: counters(v) for v in VARS}
print("# Headline solve kernel nrq_solve_kernel<16, 768, 1, 1, true>, K=8192 T=1280, 256 blocks per launch, per PHASE: the library built with")
print("# -DNRQ_STOP_AFTER=p ends every strip behind phase p; a phase's figures are the differences of consecutive builds (phases are")
print("# separated by workgroup barriers; the data movers' gather / scatter of the neighbouring line groups run inside `forward` and `hdpc`).")
print("# us = kernel duration (rocprofv3 --kernel-trace), LDS busy = SQ_LDS_IDX_ACTIVE, conflict = SQ_LDS_BANK_CONFLICT (cycles summed over CUs),")
print("# lds_frac = LDS busy / (256 CUs x the phase's clocks), clocks = SQ_BUSY_CYCLES / 32 shader engines; waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.")
for leg in ("enc", "dec"):
    print("\n== %s launch" % ("encode" if leg == "enc" else "decode"))
    print("%-12s %9s %7s %12s %12s %9s %9s %11s %11s %11s %8s" % ("phase", "us", "share", "LDS busy", "conflict", "confl/busy", "lds_frac", "VALU insts", "LDS insts", "SALU insts", "waiting"))
    prev_d, prev_c = 0.0, defaultdict(float)
    tot_d = dur["full"][leg] if dur.get("full") else None
    tot = ctr["full"][leg] if ctr.get("full") else None
    for ph, v in zip(PHASES, VARS):
        if not dur.get(v) or not ctr.get(v):
            print("%-12s (variant %s missing)" % (ph, v))
            continue
        d = dur[v][leg] - prev_d
        c = {k: ctr[v][leg][k] - prev_c[k] for k in ctr[v][leg]}
        clocks = c.get("SQ_BUSY_CYCLES", 0.0) / NSE
        busy, conf = c.get("SQ_LDS_IDX_ACTIVE", 0.0), c.get("SQ_LDS_BANK_CONFLICT", 0.0)
        print("%-12s %9.1f %6.1f%% %12.3e %12.3e %9.2f %9.2f %11.3e %11.3e %11.3e %8.2f" % (
            ph, d, 100.0 * d / tot_d if tot_d else 0.0, busy, conf, conf / busy if busy > 0 else 0.0, busy / (NCU * clocks) if clocks > 0 else 0.0,
            c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_LDS", 0.0), c.get("SQ_INSTS_SALU", 0.0),
            c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES", 0.0) > 0 else 0.0))
        prev_d, prev_c = dur[v][leg], defaultdict(float, ctr[v][leg])
    if tot_d and tot:
        clocks = tot.get("SQ_BUSY_CYCLES", 0.0) / NSE
        print("%-12s %9.1f %6.1f%% %12.3e %12.3e %9.2f %9.2f %11.3e %11.3e %11.3e %8.2f" % (
            "whole strip", tot_d, 100.0, tot["SQ_LDS_IDX_ACTIVE"], tot["SQ_LDS_BANK_CONFLICT"], tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"],
            tot["SQ_LDS_IDX_ACTIVE"] / (NCU * clocks), tot["SQ_INSTS_VALU"], tot["SQ_INSTS_LDS"], tot["SQ_INSTS_SALU"], tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"]))
