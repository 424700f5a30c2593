#!/usr/bin/env python3
"""bench.py -- RaptorQ encode+decode throughput of the MI355X path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: under torch.distributed.run (RANK / WORLD_SIZE set) this process is one rank; run on its own it starts the N ranks
itself (nanorq_amd.shard.spawn_ranks) and refuses (exit code 2) when fewer than N GPUs are visible.

Workload (config.workload "cfg3"): BASELINE.json configs[2] -- K=8192 source symbols of T=1280 bytes per
source block, 10 % independent random loss per block, decode with exactly K received symbols
(overhead 0: the GF(256)/HDPC path; a block whose matrix is rank deficient takes one more repair symbol
inside the timed region, counted).  One STEP = one batch of `--blocks` independent source blocks per
GPU: the receiver's copy is damaged again (the head of every lost row overwritten on the device -- counted in
the step: without it the decode of every step after the first would find nothing to repair), encode (source ->
intermediate symbols in HBM + R repair symbols; the encode plan is rebuilt every step, i.e. once per
256-block object like nanorq_precalculate) and decode (per-block plan from the loss pattern + solve +
regeneration of the missing symbols).  Payloads are synthetic, a function of the GLOBAL block id, and
already resident in HBM when the timed region starts.  Blocks shard across GPUs with no data-path
collective (weak scaling); rank 0 prints ONE JSON line.

`value` = 8 * (payload bytes encoded and decoded) / wall time, whole job.
Proof of work: every step destroys the lost rows of every block's receiver copy over their FULL width, poisons a few
full-width rows of EVERY block's repair and intermediate symbols (which rows changes with the step) and leaves a
column-weighted digest of them and of sampled decoded rows of every block.  After the timed region every block is verified
in full on the device (decoded block == source; source symbols regenerated from the intermediate symbols == source; repair
symbols regenerated == the encode's), two blocks are compared with the oracle byte for byte, and every step's digest must
equal the digest of that verified state.  An encode or decode that skips a block, or a column strip of one, fails the run.
`roofline`: the solve kernel (nrq_solve_kernel) --
  achieved/peak/frac  the ALGORITHMIC bytes of SURVEY.md section 8(d) (the row traffic the reference CPU path
                      performs for the same blocks, counted by the oracle on the sampled blocks) over the kernel's
                      average launch duration (HIP events on the launch stream) against the 8 TB/s HBM peak.  The
                      strip solver keeps rows in LDS, so this "reference-equivalent work rate" exceeds 1: it is NOT a
                      utilisation;
  traffic             physical HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE and
                      WRITE_SIZE in separate passes, collected at the end of the run when rocprofv3 is there;
                      `traffic_source` says where the numbers come from), next to `compulsory` (every symbol byte
                      once in, once out) and `staging` (the line-group staging buffers, once each way);
  binding             what bounds an LDS-resident solver: `lds_frac` = cycles the CUs' LDS pipelines are busy /
                      cycles available, `issue_frac` = VALU issue slots used / available, `hbm_frac` = physical
                      traffic / (duration x 8 TB/s); `nearest` names the one closest to 1.
`e2e`: the same step from pinned host buffers to pinned host buffers (PCIe both ways, one stream, nothing
overlapped): never `value`.
`cpu_baseline`: the oracle (C restatement of the reference algorithm with AVX2 GF(256) row kernels;
upstream oblas is absent so this is a "port") timed on this host, one core, on a bounded sample of the
same workload.
"""
import argparse
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

# The library runs up to six streams beside the caller's (two planner streams, the encode-plan stream, the object layer's two
# copy streams).  The HIP runtime multiplexes a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4): with
# more streams than queues, two of them share a queue and run one after the other -- seen as the encode-plan build (14 ms at
# K'=56403) serialised into the solve stream.  Must be set before the runtime initialises (torch import).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SE = 32               # shader engines (8 XCDs x 4): SQ_BUSY_CYCLES is summed over them
SIMD_PER_CU = 4
VALU_CYCLES = 2.0       # a wave64 VALU instruction issues over 2 cycles of its SIMD-32 (MI355X_MICROARCH.md, wave scheduling;
                        # tools/microbench/valu_rate.hip: one wave issues one every ~6 clocks, a SIMD still scales at 4 waves)
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES",
               "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--K", type=int, default=8192)
    ap.add_argument("--T", type=int, default=1280)
    ap.add_argument("--blocks", type=int, default=256, help="source blocks per GPU per step")
    ap.add_argument("--loss", type=float, default=0.10)
    ap.add_argument("--overhead", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="host planner threads per rank (0 = cores / ranks)")
    ap.add_argument("--cpu-sample", type=int, default=6, help="blocks timed on the CPU oracle (0 = skip)")
    ap.add_argument("--alg-sample", type=int, default=2,
                    help="with --cpu-sample 0: blocks whose reference-equivalent op counts (SURVEY 8d) the oracle forms on a "
                         "32-byte column slice, so that roofline.frac is there without the full-width CPU leg (0 = skip)")
    ap.add_argument("--no-replan", action="store_true", help="keep the encode plan cached across steps")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU: the step's blocks are split into this many groups, each on its own stream")
    ap.add_argument("--pmc", choices=("auto", "off"), default="auto",
                    help="auto: at 1 GPU, collect HBM / LDS / issue counters of the solve kernel with rocprofv3 passes of this "
                         "command after the timed run (falls back to the committed profiles/ file); off: skip")
    ap.add_argument("--plan-ahead-depth", type=int, default=2, help="planner runs kept in flight ahead of their decode (1 or 2)")
    ap.add_argument("--plan-ahead", choices=("auto", "on", "off"), default="auto",
                    help="issue the decode planner run of the NEXT step while this step's decode is being solved (the symbolic stage "
                         "needs the reception pattern only): nrq_decode_plan_ahead.  auto = for big blocks (L >= 12000) and small ones "
                         "(L < 5000); the headline (L = 8416) builds the plan inside the decode call")
    ap.add_argument("--patterns", type=int, default=4, choices=(1, 2, 4, 8),
                    help="different reception patterns the steps cycle through (step n loses pattern n mod this): no step's decode "
                         "plan can be a leftover of the step before it; a plan issued ahead is the plan of THAT step's pattern")
    ap.add_argument("--one-list", action="store_true", help="round 5's launch rule for comparison: one solve launch per batch at the strip width "
                                                                "EVERY block fits (no second block list)")
    ap.add_argument("--no-wb12", action="store_true", help="round 5's strip widths for comparison (16, 8, 4, 2 bytes: K ~ 8500-12000 on 8-byte strips)")
    ap.add_argument("--replan-late", action="store_true", help="rebuild the next step's encode plan behind the decode launch (where it was until late in "
                                                               "round 6) instead of behind the encode launch: for comparison")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffers-to-host-buffers leg")
    ap.add_argument("--one-object", choices=("auto", "on", "off"), default="auto",
                    help="ONE object over all N GPUs from rank 0's process (NANORQ_HIP_DEVICES=0..N-1, a host thread per device: "
                         "SURVEY 8(e)'s placement for cfg4 / cfg5) next to the process-sharded line; auto = when N > 1")
    ap.add_argument("--one-object-K", type=int, default=56403, help="block size of that object (cfg5: 56403, 8 blocks per GPU)")
    ap.add_argument("--one-object-T", type=int, default=1280)
    ap.add_argument("--one-object-blocks", type=int, default=0, help="blocks of that object (0 = 8 per GPU)")
    ap.add_argument("--check-blocks", type=int, default=2, help="blocks whose results are digested every step and compared with the oracle")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--force-device", type=int, default=-1,
                    help="plumbing tests only: every rank uses this GPU (several ranks on one device; use with gloo)")
    ap.add_argument("--digest-out", default="", help="plumbing tests: write the per-block SHA-256 of this rank's repair symbols here")
    return ap.parse_args()


def algorithmic_bytes(st, K, T, L, nsym_loaded, nsym_generated_rows):
    """B_alg = B_load + T*(3*(n1+nB)+2*n0) + 2*L*T + B_gen."""
    return T * (nsym_loaded + 3 * (st["n1"] + st["nB"]) + 2 * st["n0"] + 2 * L + nsym_generated_rows)


def cpu_baseline(args, src_np, lost_np, nrep_enc):
    """Oracle on one host core over a bounded sample; also yields the algorithmic byte counts."""
    import oracle
    K, T = args.K, args.T
    prm = oracle.params(K)
    n = min(args.cpu_sample, len(src_np))
    esis = np.arange(K, K + nrep_enc, dtype=np.uint32)
    t_enc = t_dec = 0.0
    balg_enc = balg_dec = 0.0
    for b in range(n):
        t0 = time.perf_counter()
        rep, _, st_e = oracle.encode_block(src_np[b], K, T, esis)
        t1 = time.perf_counter()
        lost = lost_np[b]
        keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
        nr = len(lost) + args.overhead
        ok = False
        while not ok:
            recv = np.concatenate([keep, esis[:nr]])
            syms = np.concatenate([src_np[b][keep], rep[:nr]])
            t2 = time.perf_counter()
            ok, out, st_d = oracle.decode_block(recv, syms, K, T)
            t3 = time.perf_counter()
            t_dec += t3 - t2
            nr += 1
        assert np.array_equal(out, src_np[b])
        t_enc += t1 - t0
        balg_enc += algorithmic_bytes(st_e, K, T, prm["L"], K, st_e["gen_rows"])
        balg_dec += algorithmic_bytes(st_d, K, T, prm["L"], K + st_d["overhead"], st_d["gen_rows"])
    # the same sample once more with the AVX-512 + GFNI row kernels where the host has them (oracle/rq_oracle.c row_axpy_gfni: an
    # affine bit matrix per constant -- GFNI's own multiply is fixed to the AES polynomial); reported beside the AVX2 figure
    gfni = None
    if n > 0 and getattr(oracle, "has_gfni", lambda: False)():
        g_enc = g_dec = 0.0
        oracle.set_simd(2)
        try:
            for b in range(n):
                t0 = time.perf_counter()
                rep_g, _, _ = oracle.encode_block(src_np[b], K, T, esis)
                g_enc += time.perf_counter() - t0
                keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost_np[b])
                nr, ok = len(lost_np[b]) + args.overhead, False
                while not ok:
                    t2 = time.perf_counter()
                    ok, out_g, _ = oracle.decode_block(np.concatenate([keep, esis[:nr]]), np.concatenate([src_np[b][keep], rep_g[:nr]]), K, T)
                    g_dec += time.perf_counter() - t2
                    nr += 1
                assert np.array_equal(out_g, src_np[b])
        finally:
            oracle.set_simd(1)
        gfni = {"value": 8.0 * n * K * T / (g_enc + g_dec) / 1e9, "unit": "Gbit/s", "cores": 1, "isa": "AVX-512 + GFNI (vgf2p8affineqb, 64 bytes per instruction)",
                "encode_gbps": 8.0 * n * K * T / g_enc / 1e9, "decode_gbps": 8.0 * n * K * T / g_dec / 1e9}
    # the GPU side amortises ONE encode plan over the step's blocks (the reference's "precalc" column, benchmark.c:95-96):
    # the same on the CPU -- the schedule of the first block is kept and replayed on the others (oracle.encode_block_cached)
    t_pre = None
    if n > 0 and hasattr(oracle, "encode_block_cached"):
        t_pre = 0.0
        for b in range(n):
            t0 = time.perf_counter()
            rep_c = oracle.encode_block_cached(src_np[b], K, T, esis)
            t_pre += time.perf_counter() - t0
            if b == 0:
                rep_0, _, _ = oracle.encode_block(src_np[0], K, T, esis)
                assert np.array_equal(rep_c, rep_0), "cached-plan encode differs from the plain one"
    payload = n * K * T
    # the same work on every hardware thread of the host at once (one block per thread, two rounds): what the whole
    # CPU complex delivers with the reference-equivalent algorithm; reported next to the 1-core figure
    allc = None
    nthr = os.cpu_count() or 1
    if nthr > 1 and n > 0:
        from concurrent.futures import ThreadPoolExecutor
        jobs = list(range(2 * nthr))

        def one(j):
            b = j % n
            rep_, _, _ = oracle.encode_block(src_np[b], K, T, esis)
            keep_ = np.setdiff1d(np.arange(K, dtype=np.uint32), lost_np[b])
            nr_ = len(lost_np[b]) + args.overhead   # the same workload as the 1-core leg: a rank-deficient block takes one more
            ok_ = False
            while not ok_ and nr_ <= len(esis):
                ok_, _, _ = oracle.decode_block(np.concatenate([keep_, esis[:nr_]]), np.concatenate([src_np[b][keep_], rep_[:nr_]]),
                                                K, T)
                nr_ += 1
            return ok_

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            oks = list(ex.map(one, jobs))
        dt = time.perf_counter() - t0
        allc = {"value": 8.0 * len(jobs) * K * T / dt / 1e9, "unit": "Gbit/s", "threads": nthr, "blocks": len(jobs),
                "ok": bool(all(oks))}
    return {
        "value": 8.0 * payload / (t_enc + t_dec) / 1e9, "unit": "Gbit/s", "cores": 1, "kind": "port",
        "sample": "%d blocks of K=%d T=%d, encode (+%d repair) and decode (%.0f%% loss, +%d), oracle/rq_oracle.c "
                  "AVX2=%s, 1 thread" % (n, K, T, nrep_enc, args.loss * 100, args.overhead, oracle.has_avx2()),
        "encode_gbps": 8.0 * payload / t_enc / 1e9, "decode_gbps": 8.0 * payload / t_dec / 1e9,
        "timed": "encode = plan + replay + repair generation per block (reference 'encode' column); decode = plan + replay per block",
        "precalc": ({"value": 8.0 * payload / (t_pre + t_dec) / 1e9, "unit": "Gbit/s", "encode_gbps": 8.0 * payload / t_pre / 1e9,
                     "what": "encode with the schedule of the first block replayed on the others (plan amortised over the "
                             "blocks, as on the GPU side: reference 'precalc' column, benchmark.c:95-96, :210); decode as above"}
                    if t_pre else None),
        "isa": "AVX2 (split-nibble vpshufb)" if oracle.has_avx2() else "scalar log/antilog", "gfni": gfni,
        "host_cpu": _cpu_model(), "host_threads": os.cpu_count(), "all_cores": allc,
    }, balg_enc / n, balg_dec / n


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------ PMC (rocprofv3) ----
def _run_group(cmd, cwd, env, timeout):
    """subprocess with a process group of its own, so that a hung profiler run is ended as a whole"""
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        return p.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        p.wait()
        return -9


def pmc_collect(args):
    """rocprofv3 --pmc passes (one counter group per pass, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
    cannot share a pass) of a short run of THIS command; returns {counter: mean value per full-batch solve launch}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    import sqlite3
    inner = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--alg-sample", "0", "--pmc", "off",
             "--no-e2e", "--K", str(args.K), "--T", str(args.T), "--blocks", str(args.blocks), "--loss", str(args.loss), "--overhead", str(args.overhead)]
    if args.no_replan:
        inner.append("--no-replan")
    if args.no_wb12:
        inner.append("--no-wb12")
    if args.one_list:
        inner.append("--one-list")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = {}
    tmp = tempfile.mkdtemp(prefix="nrq_pmc_", dir="/tmp")
    try:
        for i, grp in enumerate(PMC_PASSES):
            d = os.path.join(tmp, "p%d" % i)
            # (the SQ pass also records the kernel trace -- allowed beside --pmc, unlike the API / copy traces -- so that the
            # counters and the duration they are divided by come from the SAME run: under the profiler kernels run one at a
            # time, while in the timed region planner runs issued ahead execute beside the solve)
            ktrace = ["--kernel-trace"] if "SQ_BUSY_CYCLES" in grp else []
            rc = _run_group([exe, *ktrace, "--pmc", *grp, "-d", d, "--"] + inner, "/tmp", env, 240)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")] if os.path.isdir(d) else []
            if rc != 0 or not dbs:
                return None, "rocprofv3 pass %s failed (rc %s)" % (grp, rc)
            rows = []
            for path in dbs:
                db = sqlite3.connect(path)
                rows += db.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
                if ktrace:
                    try:
                        kr = [r for r in db.execute("select name, grid_x, duration from kernels order by start").fetchall() if "nrq_solve_kernel" in r[0]]
                        gx = max(r[1] for r in kr) if kr else 0
                        du = [r[2] / 1e6 for r in kr if r[1] == gx]
                        du = du[2:] if len(du) > 3 else du          # (the warm-up step's pair is cold)
                        if du:
                            out["_serialized_ms"] = sum(du) / len(du)
                    except sqlite3.Error:
                        pass
                db.close()
            rows = [r for r in rows if "nrq_solve_kernel" in r[0]]
            if not rows:
                return None, "no solve-kernel rows in pass %s" % grp
            gmax = max(r[1] for r in rows)
            for cname in grp:
                v = [r[3] * (1024.0 if cname in ("FETCH_SIZE", "WRITE_SIZE") else 1.0) for r in rows if r[1] == gmax and r[2] == cname]
                if v:
                    out[cname] = sum(v) / len(v)
    except Exception as e:  # noqa: BLE001 -- profiling is best effort, the bench line must still come out
        return None, "pmc collection failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, "rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1` run by this process after the timed region"


def pmc_committed(args):
    """the counters of the committed profile (tools/collect_profiles.sh), if it is of this workload"""
    for name in ("r6_pmc.json", "r5_pmc.json", "r4_pmc.json", "r3_pmc.json", "r2_pmc.json", "r1_pmc_hbm.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if (rec.get("K"), rec.get("T"), rec.get("blocks")) != (args.K, args.T, args.blocks):
            continue
        return {k: v["mean"] for k, v in rec.get("per_launch", {}).items()}, "committed profiles/%s" % name
    return None, "none"


def binding_model(c, avg_ms, ncu):
    """What bounds an LDS-resident solver: LDS pipeline occupancy and VALU issue slots (and physical HBM traffic)."""
    if not c or "SQ_BUSY_CYCLES" not in c:
        return None
    cyc = c["SQ_BUSY_CYCLES"] / N_SE                     # kernel duration in shader cycles
    cu_cycles = cyc * ncu
    out = {"kernel_cycles": cyc, "clock_ghz": cyc / (avg_ms * 1e-3) / 1e9 if avg_ms else None}
    if "SQ_LDS_IDX_ACTIVE" in c:
        out["lds_frac"] = c["SQ_LDS_IDX_ACTIVE"] / cu_cycles
        if "SQ_LDS_BANK_CONFLICT" in c:
            out["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_INSTS_VALU" in c:
        out["issue_frac"] = c["SQ_INSTS_VALU"] * VALU_CYCLES / (cu_cycles * SIMD_PER_CU)
        out["issue_frac_at_4_cycles"] = c["SQ_INSTS_VALU"] * 4.0 / (cu_cycles * SIMD_PER_CU)   # a wave64 instruction as 4 passes of a 16-lane SIMD
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        out["waves_waiting_share"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
        if k in c:
            out[k.lower() + "_per_launch"] = c[k]
    out["model"] = ("lds_frac = SQ_LDS_IDX_ACTIVE / (CUs x kernel cycles), issue_frac = SQ_INSTS_VALU x 2 cycles / (CUs x 4 SIMDs x "
                    "kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines")
    return out


def e2e_leg(args):
    """host buffers -> host buffers (PCIe both ways) through the object API's page-locked path, in a process of its own and
    BEFORE this process touches the GPU: one object of up to 128 of the step's blocks through include/nanorq_batch.h --
    nanorq_generate_symbols_all, nanorq_encode_range_all, nanorq_decoder_add_symbols(_async), nanorq_repair_all; same loss
    patterns as the timed region.  (In this process, or beside it once it holds a GPU context, the object layer's six streams
    share hardware queues / the GPU's process scheduling with the bench context's and torch's: the receiver pipeline's
    upload, sorting and download streams then took turns -- repair 44-46 ms against 31 ms alone.)"""
    K, T, Z = args.K, args.T, min(args.blocks, 128)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_receiver.py"), str(K), str(T), str(Z), str(args.loss), "--legs"]
    env1 = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env1.pop(k, None)
    r = subprocess.run(cmd, env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, "end-to-end leg failed: %s" % r.stderr.decode()[-400:]
    rec = json.loads(r.stdout.decode().strip().split("\n")[-1])
    legs, legs_p = rec["serial"], rec["pipeline"]
    assert legs["ok"] and legs_p["ok"], "end-to-end leg: the decoded object differs from the source"
    gbit = 8.0 * Z * K * T / 1e9
    pipelined_ms = legs["sender_ms"] + 8e-6 * Z * K * T / legs_p["receiver_gbps"]
    return {"value": gbit / (pipelined_ms * 1e-3), "unit": "Gbit/s", "blocks": Z, "ms_total": pipelined_ms,
            "value_four_calls": legs["value"], "ms_total_four_calls": legs["total_ms"], "sender_gbps_two_calls": legs["sender_gbps_two_calls"],
            "generate_gbps": legs["generate_gbps"], "ingest_gbps": legs["add_gbps"], "repair_gbps": legs["repair_gbps"],
            "repair_symbols_ms": legs["repair_symbols_ms"], "received_symbols": legs["received_symbols"],
            "sender_gbps": legs["sender_gbps"], "receiver_gbps": legs_p["receiver_gbps"], "receiver_gbps_serial": legs["receiver_gbps"],
            "receiver_pipeline_ms": {"add": 8e-6 * Z * K * T / legs_p["add_gbps"], "repair": 8e-6 * Z * K * T / legs_p["repair_gbps"]},
            "what": "object API on page-locked memory (nanorq_batch.h), host buffers to host buffers on ONE GPU, the sender and then "
                    "the receiver: value = payload / (sender + receiver), each station as its pipeline -- sender_gbps = ONE "
                    "nanorq_encode_range_all call (upload, solve, repair symbols and their way back overlapped), receiver_gbps = "
                    "nanorq_decoder_add_symbols_async + nanorq_repair_all (ingest, plan, solve and the way back overlapped); "
                    "value_four_calls / sender_gbps_two_calls / receiver_gbps_serial and the per-leg rates (generate, ingest, repair) are "
                    "the waiting calls one after the other.  Run in a process of its own before the timed region.  Each leg crosses "
                    "PCIe once: the per-leg rates stand against ~440 Gbit/s of link per direction.  Never `value` of the bench line."}


def config_name(K, T, loss, overhead):
    """BASELINE.json's name for the configuration actually run (SURVEY.md section 8(d) restates them), else "custom"."""
    table = {(100, 1024): "cfg1", (1024, 1280): "cfg2", (8192, 1280): "cfg3", (27000, 65504): "cfg4", (56403, 1280): "cfg5"}
    name = table.get((K, T))
    if name is None:
        return "custom"
    want_loss = {"cfg1": (0.0, 0.06), "cfg2": (0.05, 0.06), "cfg3": (0.10,), "cfg4": (0.10,), "cfg5": (0.20,)}[name]
    if not any(abs(loss - w) < 1e-9 for w in want_loss):
        return name + "-shape"      # the configuration's block shape at another loss rate
    return name


def algorithmic_sample(args, lost_np, nrep_enc, nblocks):
    """n1 / nB / n0 of the reference-equivalent plan do not depend on T: the oracle run on a 32-byte column slice of synthetic
    symbols gives the algorithmic bytes of SURVEY 8(d) per block when the full-width CPU leg is skipped (--cpu-sample 0) or
    too big to run (cfg4: 1.77 GB per block)."""
    import oracle
    K, T, Ts = args.K, args.T, 32
    prm = oracle.params(K)
    esis = np.arange(K, K + nrep_enc, dtype=np.uint32)
    rng = np.random.default_rng(99)
    be = bd = 0.0
    n = min(nblocks, len(lost_np))
    for b in range(n):
        src = rng.integers(0, 256, (K, Ts), dtype=np.uint8)
        rep, _, st_e = oracle.encode_block(src, K, Ts, esis)
        lost = lost_np[b]
        keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
        nr, ok = len(lost) + args.overhead, False
        while not ok:
            ok, out, st_d = oracle.decode_block(np.concatenate([keep, esis[:nr]]), np.concatenate([src[keep], rep[:nr]]), K, Ts)
            nr += 1
        assert np.array_equal(out, src)
        be += algorithmic_bytes(st_e, K, T, prm["L"], K, st_e["gen_rows"])
        bd += algorithmic_bytes(st_d, K, T, prm["L"], K + st_d["overhead"], st_d["gen_rows"])
    return be / n, bd / n


def main():
    args = parse()
    from nanorq_amd import shard
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if not shard.launched():
        if args.gpus > 1:
            # `python bench.py --gpus N` on its own: this process becomes the launcher of N ranks (one process per GPU, RCCL for
            # the barrier and the timing reduction only -- DESIGN.md section 8) and rank 0 prints the one line
            if args.force_device < 0:
                import torch
                have = torch.cuda.device_count() if torch.cuda.is_available() else 0
                if have < args.gpus:
                    sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible: refusing to print a line for fewer "
                                     "GPUs than asked\n" % (args.gpus, have))
                    raise SystemExit(2)
            env_l = dict(os.environ)
            if args.force_device >= 0:
                env_l["NANORQ_FORCE_DEVICE"] = str(args.force_device)   # (CPU placement: every rank beside that one GPU)
            raise SystemExit(shard.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env_l))
    if args.force_device >= 0:
        os.environ.setdefault("NANORQ_FORCE_DEVICE", str(args.force_device))
    shard.bind_self()   # (a rank of torch.distributed.run: its GPU's cores, like spawn_ranks gives its children; before torch comes up)
    import torch
    import nanorq_amd
    from util import loss_pattern

    rank, world, local = shard.env_rank()
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.force_device < 0 and local >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d has no GPU (%d visible)\n" % (rank, torch.cuda.device_count()))
        raise SystemExit(2)
    e2e = e2e_leg(args) if (rank == 0 and world == 1 and not args.no_e2e and args.streams <= 1) else None
    if args.force_device >= 0:
        local = args.force_device
    torch.cuda.set_device(local)
    # RCCL; only the barrier and the timing reduction use it
    shard.init(args.dist_backend, device_id=torch.device("cuda", local) if args.dist_backend == "nccl" else None)
    dev = torch.device("cuda", local)
    # host planner threads: the cores this rank was given (shard.spawn_ranks binds a rank to its GPU's NUMA node), else its share
    ncpu_mine = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = args.threads or (max(1, ncpu_mine) if os.environ.get("NANORQ_RANK_CPUS") else max(1, ncpu_mine // max(1, world)))
    nstreams = max(1, min(args.streams, args.blocks))
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]
    ctxs = [nanorq_amd.Context(local, st.cuda_stream) for st in streams]   # one context per stream, same GPU
    for c_ in ctxs:
        c_.set_threads(threads)
        if args.one_list:
            c_.set_option("no_lists", 1)
        if args.no_wb12:
            c_.set_option("no_wb12", 1)
    ctx = ctxs[0]

    K, T, NB = args.K, args.T, args.blocks
    prm = nanorq_amd.params(K)
    L = prm["L"]
    # weak scaling: the job has world*NB source blocks per step, block b lives on GPU b mod world;
    # payload and loss pattern are functions of the GLOBAL block id, payload generated on the device
    my_blocks = shard.blocks_of(rank, world, world * NB)
    # byte i of global block gb = low byte of a 64-bit mix of (gb * K * T + i): a function of the GLOBAL block id,
    # not of the rank or of how many GPUs share the job; generated on the device, a few blocks per pass
    src = torch.empty((NB, K, T), dtype=torch.uint8, device=dev)
    per = K * T
    ar = torch.arange(per, device=dev, dtype=torch.int64).view(1, per)
    ch = max(1, (32 << 20) // per)
    for b0 in range(0, NB, ch):
        gbs = torch.tensor(my_blocks[b0:b0 + ch], device=dev, dtype=torch.int64).view(-1, 1)
        x = gbs * per + ar + 0x1234567
        x = (x ^ (x >> 31)) * 0x7FB5D329728EA185           # (int64 arithmetic wraps)
        x = (x ^ (x >> 27)) * -0x7E25210B43D22BB3
        x = x ^ (x >> 33)
        src[b0:b0 + ch] = ((x >> 24) & 0xFF).to(torch.uint8).view(-1, K, T)
    del ar, x
    # reception patterns: step n loses pattern n mod NPAT (seed 1000 + pattern: pattern 0 is the one earlier rounds used every step)
    NPAT = args.patterns
    lost_p = [[loss_pattern(K, args.loss, seed=1000 + q, block=gb) for gb in my_blocks] for q in range(NPAT)]
    lost = lost_p[0]
    max_lost = max(len(x) for lp in lost_p for x in lp)
    nrep = max_lost + args.overhead + 3  # repair symbols generated per block by the encoder (incl. spares)
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep = torch.empty((NB, nrep, T), dtype=torch.uint8, device=dev)
    inter = torch.empty((NB, L, T), dtype=torch.uint8, device=dev)
    work = src.clone()  # what the receiver holds: source block with the lost rows destroyed
    # rows (block * K + esi) the channel destroyed: overwritten again at the start of EVERY step
    lost_rows_p = [torch.from_numpy(np.concatenate([b * K + lp[b].astype(np.int64) for b in range(NB)])).to(dev) for lp in lost_p]
    work_rows = work.view(NB * K, T)
    # the WHOLE lost row is destroyed (a decode that skips any column strip leaves 0xEE behind) -- written as 8-byte words:
    # torch's index_fill moves one element per thread, and byte elements made it 0.34 ms per step for 268 MB
    damage, damage_value = work_rows, 0xEE
    if T % 8 == 0:
        damage, damage_value = work.view(torch.int64).view(NB * K, T // 8), -0x1111111111111112   # = 0xEEEEEEEEEEEEEEEE
    lost_arr_p, nlost_p, nr_first_p, nr_avail_p = [], [], [], []
    spare = 3   # repair symbols a block may take beyond (lost + overhead) if its system is rank deficient
    for lp in lost_p:
        la = np.zeros((NB, max_lost + 1), np.uint32)
        for b in range(NB):
            la[b, :len(lp[b])] = lp[b]
        nl = np.array([len(x) for x in lp], np.uint32)
        lost_arr_p.append(la); nlost_p.append(nl)
        nr_first_p.append((nl + args.overhead).astype(np.uint32))
        nr_avail_p.append((nl + args.overhead + spare).astype(np.uint32))
    nlost, nr_first = nlost_p[0], nr_first_p[0]
    resi = np.tile(esis, (NB, 1))
    retries = 0

    # proof of work.  (1) oracle: `chk` blocks are compared with the oracle byte for byte after the timed region.  (2) every
    # step, EVERY block: a few full-width rows of its repair and intermediate symbols (which rows: a function of the step's
    # phase) are poisoned before the encode, and a digest of those rows and of sampled lost rows of every block's receiver
    # copy (destroyed over their full width before the decode) is left behind -- an encode that skips a block or a column
    # strip of one, or a decode that does, leaves poison in the digest.  (3) after the timed region every block is verified in
    # full (systematic property and repair symbols regenerated from the intermediate symbols, all on the device), and the
    # digests are compared with the ones formed from that verified state.
    chk = sorted(set([0, NB - 1][:max(0, args.check_blocks)]))[:args.check_blocks] if args.check_blocks > 0 else []
    nchk_rep = int(min(min(nf[chk].min() for nf in nr_first_p), nrep)) if chk else 0   # repair symbols every reception of theirs uses
    digests = []
    NPHASE, REP_S, INT_S, WRK_S = 8, 2, 8, 8
    rng = np.random.default_rng(4242)
    ph_rep, ph_int, ph_wrk = [], [], []
    if args.check_blocks > 0:
        for ph_ in range(NPHASE):   # (phase ph_ belongs to the steps n with n mod NPHASE == ph_: their pattern is ph_ mod NPAT)
            lq = lost_p[ph_ % NPAT]
            ph_rep.append(torch.from_numpy(np.concatenate([b * nrep + rng.integers(0, nrep, REP_S) for b in range(NB)])).to(dev))
            ph_int.append(torch.from_numpy(np.concatenate([b * L + rng.integers(0, L, INT_S) for b in range(NB)])).to(dev))
            ph_wrk.append(torch.from_numpy(np.concatenate([b * K + lq[b][rng.integers(0, len(lq[b]), WRK_S)].astype(np.int64)
                                                           for b in range(NB) if len(lq[b])])).to(dev))
    # column weights (the far end of T counts); rows are read as 8-byte words: no widened copy of the sampled rows (with 8192
    # blocks of K=100 that copy was 1.2 GB per step)
    W8 = T // 8
    wcol = (torch.arange(max(1, W8), device=dev, dtype=torch.int64) % 251 + 1).view(1, -1)

    # block ranges of the stream groups
    bounds = [(NB * g_) // nstreams for g_ in range(nstreams + 1)]
    groups = [(bounds[g_], bounds[g_ + 1]) for g_ in range(nstreams)]

    replan_early = L >= 12000   # (the library's threshold for device-built encode plans, NRQ_ENCPLAN_DEV_MIN_L)
    # host-built encode plans: rebuilt behind the encode launch while the build is shorter than the encode solve it runs beside
    # (K=8192: ~5 ms against 6.1 -- behind the decode launch it had the decode solve's 5.9 ms and nothing to spare: 17.8 ms steps on
    # a slower host core), behind the decode launch from L = 9000 on, where the build is the longer one and would hold the decode
    # call back (K=10000: 22.1 / 21.4 ms per step behind the encode / the decode launch)
    replan_behind_encode = L < 9000 and not args.replan_late
    # auto = always.  Big blocks: the planner's latency leaves the critical path (two steps' runs side by side).  Many small
    # blocks (the planner stays on the solve stream): what goes is the host's wait for the planner in the MIDDLE of the step and
    # its work behind it (2048 plan headers, the solve launch) with the GPU idle -- K=1000: 18.5 -> 18.1 ms on a quiet host,
    # 28.8 -> 18.1 ms on a busy one; the headline 15.87 -> 15.74.
    # Round 6: auto = big blocks only (L >= 12000: one planner workgroup per block is 10-18 ms of latency on a few CUs, which a
    # receiver of a stream of objects hides behind the solves of the objects before -- with the NEXT object's own reception
    # pattern, see --patterns).  Elsewhere the plan is built inside the decode call, as a receiver does that learns the pattern
    # when it decodes: that is the headline's `value` now (round 5 quoted the run-ahead figure; with a planner of 1.65 ms that
    # owns every CU while it runs the two differ by < 1 % either way: 13.92 against 14.0-14.2 ms per step).
    # Small blocks (L < 5000) keep the run-ahead too: thousands of plan headers come back to the host in the middle of the step and
    # the solve launch is formed from them with the GPU idle -- K=500: 21.2 -> 20.4 ms per step, K=1000 18.4 -> 18.0 (round 6, same box).
    plan_ahead = nstreams == 1 and (args.plan_ahead == "on" or (args.plan_ahead == "auto" and (replan_early or L < 5000)))
    ahead_depth = max(1, min(2, args.plan_ahead_depth))
    ahead_out = 0   # planner runs issued ahead and not consumed yet

    rep_rows_all, int_rows_all = rep.view(NB * nrep, T), inter.view(NB * L, T)
    step_no = 0

    def poison(ph):
        rep_rows_all.index_fill_(0, ph_rep[ph], 0xCD)
        int_rows_all.index_fill_(0, ph_int[ph], 0xCD)

    def digest_of(ph, wrows):
        def dg(t, ix):
            r = t.index_select(0, ix)
            if T % 8 == 0:
                return (r.view(torch.int64) * wcol).sum()     # (wraps: a checksum)
            return r.sum(dtype=torch.int64)
        return torch.stack([dg(t, ix) for t, ix in ((rep_rows_all, ph_rep[ph]), (int_rows_all, ph_int[ph]), (wrows, ph_wrk[ph]))])

    # strip widths of the decode launches: a batch whose blocks do not all fit the widest strip runs as two block lists
    lists = {"launches": 0, "second_list_launches": 0, "second_list_blocks": 0, "widths": {}}

    def pat_of(n):
        return (n % NPHASE) % NPAT

    def step(ahead=None):
        nonlocal retries, step_no, ahead_out
        ahead = plan_ahead if ahead is None else ahead
        enc_stats = dec_stats = None
        ph = step_no % NPHASE
        q = pat_of(step_no)   # this step's reception pattern
        lost_arr, nlost, nr_first, nr_avail = lost_arr_p[q], nlost_p[q], nr_first_p[q], nr_avail_p[q]
        this_step = step_no
        step_no += 1
        # the channel: the receiver's copy loses its rows again, over their full width (part of the step)
        damage.index_fill_(0, lost_rows_p[q], damage_value)
        if ph_rep:
            poison(ph)
        for (lo, hi), c_ in zip(groups, ctxs):
            n_ = hi - lo
            c_.encode_blocks(K, T, n_, src[lo].data_ptr(), K * T, rep[lo].data_ptr(), nrep * T, esis, inter[lo].data_ptr(),
                             L * T)
            enc_stats = enc_stats or c_.stats()
        if not args.no_replan and (replan_early or replan_behind_encode):
            # one encode plan per step (= per 256-block object, like nanorq_precalculate): the plan for the NEXT step's encode is
            # rebuilt here, right behind this step's encode launch.  Big K': by the device planner, asynchronously on a stream of
            # its own (nrq_precalculate only enqueues it), beside this step's work.  Smaller K': on the host (4-6 ms at K=8192),
            # while the GPU runs this step's encode solve -- it used to sit behind the decode launch, where it had the decode solve
            # (5.9 ms) to hide in and nothing to spare: with the decode plan built inside the decode call a slower host core turned
            # every step from 14.0 into 17.8 ms (seen in two of three full default runs late in round 6)
            for c_ in ctxs:
                c_.clear_plan_cache()
                c_.precalculate(K)
        for (lo, hi), c_ in zip(groups, ctxs):
            n_ = hi - lo
            # decode from exactly (lost + overhead) repair symbols per block; a block whose system turns out rank
            # deficient (about 1.5 % at overhead 0) takes one more symbol at a time inside the planner -- the receiver
            # "got another packet" (nanorq_repair_block is retryable, lib/nanorq.c:620-623) without a second pass
            st, used = c_.decode_blocks_lazy(K, T, n_, work[lo].data_ptr(), K * T, lost_arr[lo:hi], nlost[lo:hi], resi[lo:hi],
                                             nr_first[lo:hi], nr_avail[lo:hi], rep[lo].data_ptr(), nrep * T)
            dec_stats = dec_stats or c_.stats()
            ds_ = c_.stats()
            lists["launches"] += 1
            lists["second_list_launches"] += 1 if ds_.get("blocks_b") else 0
            lists["second_list_blocks"] += ds_.get("blocks_b", 0)
            lists["widths"][str(ds_["strip_bytes"])] = lists["widths"].get(str(ds_["strip_bytes"]), 0) + 1
            if not st.all():
                raise RuntimeError("decode failed for %d blocks" % int((st == 0).sum()))
            retries += int((used - nr_first[lo:hi]).sum())
            if ahead:
                # the decode plans of the NEXT steps (same reception pattern every step in this bench; a planner run per step
                # all the same): enqueued now, they run beside this step's decode solve and the next steps' solves
                if c_.stats().get("plan_ahead"):
                    ahead_out -= 1
                else:
                    ahead_out = 0   # (a decode that found no run issued for it has discarded the waiting ones)
                while ahead_out < ahead_depth:
                    # (the run for step this_step + 1 + ahead_out, with THAT step's reception pattern -- a receiver that has the
                    # next objects' packets while this one is being solved; a run whose lists differ from its decode's is discarded)
                    q2 = pat_of(this_step + 1 + ahead_out)
                    c_.decode_plan_ahead(K, T, n_, work[lo].data_ptr(), K * T, lost_arr_p[q2][lo:hi], nlost_p[q2][lo:hi], resi[lo:hi],
                                         nr_first_p[q2][lo:hi], nr_avail_p[q2][lo:hi], rep[lo].data_ptr(), nrep * T)
                    ahead_out += 1
        if ph_rep and nstreams == 1:
            digests.append((ph, digest_of(ph, work_rows)))
        if not args.no_replan and not replan_early and not replan_behind_encode:   # (behind the decode launch: K' ~ 9000-12000, and --replan-late)
            for c_ in ctxs:
                c_.clear_plan_cache()
                c_.precalculate(K)
        return enc_stats, dec_stats

    def barrier():
        shard.barrier(world)
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    retries = 0
    lists.update({"launches": 0, "second_list_launches": 0, "second_list_blocks": 0, "widths": {}})
    digests.clear()   # (step_no runs on: the planner runs issued ahead in the warm-up are those of the first timed steps' patterns)
    for c_ in ctxs:
        c_.ktime_enable(True)
    t0 = time.perf_counter()
    cpu0 = time.process_time()   # (all threads of this rank's process: host plan build, launch paths, torch)
    for _ in range(args.steps):
        enc_stats, dec_stats = step()
    barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_ms = (time.process_time() - cpu0) / args.steps * 1e3
    # solve-kernel launches of all streams on one time axis (HIP events recorded around each launch)
    intervals, ptimes = [], []
    for c_ in ctxs:
        ptimes += c_.ptime_read()
        intervals += c_.ktime_read_intervals(ref=ctx)
        c_.ktime_enable(False)
    ktimes = [d for _, d in intervals]
    rdev = dev if args.dist_backend == "nccl" else None
    elapsed = shard.reduce_max(elapsed, world, device=rdev)   # the slowest rank defines the step time
    # the same step with the decode plan built INSIDE the decode call (a receiver that learns the reception pattern when it
    # decodes): reported beside the headline, never `value`.  The runs still waiting are consumed first (untimed).
    ms_plan_in_call = None
    ms_plan_ahead = None
    if not plan_ahead and world == 1 and args.steps >= 2 and nstreams == 1 and args.plan_ahead == "auto":
        # the other form beside the headline: planner runs issued two steps ahead (never `value`)
        for c_ in ctxs:
            c_.ktime_enable(False)
        for _ in range(ahead_depth + 1):
            step(ahead=True)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        nplain = min(args.steps, 5)
        for _ in range(nplain):
            step(ahead=True)
        torch.cuda.synchronize(dev)
        ms_plan_ahead = (time.perf_counter() - t1) / nplain * 1e3
        step(ahead=False)   # (discards the runs still waiting)
        torch.cuda.synchronize(dev)
    if plan_ahead and world == 1 and args.steps >= 2:
        for c_ in ctxs:
            c_.ktime_enable(False)
        for _ in range(ahead_depth + 1):
            step(ahead=False)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        nplain = min(args.steps, 5)
        for _ in range(nplain):
            step(ahead=False)
        torch.cuda.synchronize(dev)
        ms_plan_in_call = (time.perf_counter() - t1) / nplain * 1e3   # (these steps' digests are checked with the others)
    retries_total = int(shard.reduce_sum(retries, world, device=rdev))

    # ---- correctness of what was timed ----
    assert torch.equal(work, src), "decoded blocks differ from the source blocks"
    check = {"blocks": chk, "steps_digested": len(digests), "oracle": False, "all_blocks_verified": False,
             "rows_per_block_per_step": {"repair": REP_S, "intermediate": INT_S, "decoded": WRK_S} if ph_rep else None}
    if ph_rep:
        # every block in full, on the device: the source symbols regenerated from the block's intermediate symbols must be
        # the block (systematic property, RFC 6330 5.3.3.4.2), and its repair symbols regenerated must be the ones the
        # encode left.  (The decode's output was compared with `src` above.)
        Kp = prm["Kp"]
        vb = max(1, min(NB, (256 << 20) // (K * T)))
        tmp = torch.empty((vb, K, T), dtype=torch.uint8, device=dev)
        tmp_r = torch.empty((vb, nrep, T), dtype=torch.uint8, device=dev)
        src_isis = np.arange(K, dtype=np.uint32)
        rep_isis = (esis + (Kp - K)).astype(np.uint32)
        for b0 in range(0, NB, vb):
            n_ = min(vb, NB - b0)
            tmp.fill_(0x5A)
            tmp_r.fill_(0x5A)
            ctx.gen_symbols(K, T, n_, inter[b0].data_ptr(), L * T, src_isis, tmp.data_ptr(), K * T)
            ctx.gen_symbols(K, T, n_, inter[b0].data_ptr(), L * T, rep_isis, tmp_r.data_ptr(), nrep * T)
            ctx.sync()
            assert torch.equal(tmp[:n_], src[b0:b0 + n_]), "systematic property fails for a block in %d..%d" % (b0, b0 + n_)
            assert torch.equal(tmp_r[:n_], rep[b0:b0 + n_]), "repair symbols of a block in %d..%d are not LT(intermediate)" % (b0, b0 + n_)
        del tmp, tmp_r
        check["all_blocks_verified"] = True
        # the digests every timed step left behind against the ones of the verified state
        src_rows = src.view(NB * K, T)
        expect = {}
        for ph, d in digests:
            if ph not in expect:
                expect[ph] = digest_of(ph, src_rows)
            assert torch.equal(d, expect[ph]), "a timed step left poisoned or wrong rows behind (phase %d)" % ph
    if chk:
        import oracle
        for b in chk:   # the oracle's repair + intermediate symbols of the checked blocks, byte for byte
            sb = src[b].cpu().numpy()
            r_rep, r_int, _ = oracle.encode_block(sb, K, T, esis[:nchk_rep], want_inter=True)
            assert np.array_equal(rep[b, :nchk_rep].cpu().numpy(), r_rep), "repair symbols of block %d differ from the oracle" % b
            assert np.array_equal(inter[b].cpu().numpy(), r_int), "intermediate symbols of block %d differ from the oracle" % b
        check["oracle"] = True
    if args.digest_out:
        import hashlib
        with open(args.digest_out + (".rank%d" % rank if world > 1 else ""), "w") as f:
            json.dump({str(gb): hashlib.sha256(rep[b, :int(nr_first[b])].cpu().numpy().tobytes()).hexdigest()
                       for b, gb in enumerate(my_blocks)}, f)

    # ---- ONE object over the N GPUs of one process (the other placement of SURVEY 8(e)); rank 0 runs it in a process of its
    # own (the object layer reads its device list once) while the other ranks wait at the closing barrier ----
    one_object = None
    shard.finalize(world)   # (the last collective is behind us: the other ranks leave, rank 0 may take its time)
    if rank == 0 and (args.one_object == "on" or (args.one_object == "auto" and world > 1)) and args.force_device < 0:
        for c_ in ctxs:
            c_.sync()
        torch.cuda.empty_cache()
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_one_object.py"), "--devices", ",".join(str(d) for d in range(world)),
               "--K", str(args.one_object_K), "--T", str(args.one_object_T), "--blocks", str(args.one_object_blocks),
               "--loss", "0.2" if args.one_object_K == 56403 else str(args.loss)]
        env1 = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NANORQ_HIP_DEVICE", "NANORQ_RANK_CPUS", "OMP_NUM_THREADS"):
            env1.pop(k, None)
        # (this process is bound to ITS GPU's cores; the one-object run drives every device from one process: all cores again)
        wide = shard.ORIG_AFFINITY or (set(range(os.cpu_count() or 1)) if os.environ.get("NANORQ_RANK_CPUS") else None)

        def _widen():
            if wide:
                try:
                    os.sched_setaffinity(0, wide)
                except OSError:
                    pass
        try:
            # (bounded: the thread-per-device path has never met a node with several GPUs, and the bench line must not wait for it)
            r = subprocess.run(cmd, env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, preexec_fn=_widen)
            one_object = (json.loads(r.stdout.decode().strip().split("\n")[-1]) if r.returncode == 0 else
                          {"error": "rc %d: %s" % (r.returncode, r.stderr.decode()[-300:])})
        except Exception as e:  # noqa: BLE001 -- the bench line must still come out
            one_object = {"error": repr(e)}

    if rank == 0:
        payload_step = world * NB * K * T
        value = 8.0 * payload_step * args.steps / elapsed / 1e9
        cpu, balg_enc, balg_dec = (None, None, None)
        if args.cpu_sample > 0 and world == 1:   # (the CPU baseline is a one-GPU-run entry; N > 1 keeps the op counts for the roofline only)
            src_np = src[:args.cpu_sample].cpu().numpy()
            cpu, balg_enc, balg_dec = cpu_baseline(args, src_np, lost, nrep)
        elif args.alg_sample > 0:
            balg_enc, balg_dec = algorithmic_sample(args, lost, nrep, args.alg_sample)
        # solve-kernel launches in the timed region: [encode, decode(, retries...)] per step
        roof = None
        if ktimes:
            # every step launches the solve kernel 2*nstreams times (encode and decode of each stream group); launches of
            # different streams overlap in time, so the kernel's rate is the algorithmic bytes of all launches divided by
            # the time during which at least one of them was running (= the plain average duration when nstreams == 1)
            avg_ms = sum(ktimes) / len(ktimes)
            busy, cur_s, cur_e = 0.0, None, None
            for s0, d0_ in sorted(intervals):
                if cur_e is None or s0 > cur_e:
                    busy += 0.0 if cur_e is None else cur_e - cur_s
                    cur_s, cur_e = s0, s0 + d0_
                else:
                    cur_e = max(cur_e, s0 + d0_)
            busy += 0.0 if cur_e is None else cur_e - cur_s
            blocks_per_launch = NB / float(nstreams)
            eff_ms = busy / len(ktimes)   # busy time attributable to one launch
            # physical bytes a launch cannot avoid / chooses to move (mean of the encode and the decode launch)
            gaps = float(np.mean([nl.mean() for nl in nlost_p]))
            comp_enc = NB * T * (K + L + nrep)                       # source in; intermediate + repair symbols out
            comp_dec = NB * T * (K + gaps + gaps)                    # received symbols in; recovered symbols out
            stage_enc = NB * T * (L + (L + nrep))                    # slot image + results, once through the staging buffers
            stage_dec = NB * T * ((L + args.overhead) + gaps)
            compulsory, staging = 0.5 * (comp_enc + comp_dec), 0.5 * (stage_enc + stage_dec)
            counters, csrc = (None, "off")
            if args.pmc == "auto" and world == 1 and nstreams == 1:
                for c_ in ctxs:
                    c_.sync()
                counters, csrc = pmc_collect(args)
                if counters is None:
                    why = csrc
                    counters, csrc = pmc_committed(args)
                    csrc += " (live collection: %s)" % why
            traffic = traffic_raw = None
            if counters and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
                # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide (16 B per lane) reads -- every read
                # of this kernel is one -- so it is doubled; WRITE_SIZE is taken as reported
                traffic_raw = counters["FETCH_SIZE"] + counters["WRITE_SIZE"]
                traffic = 2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]
            # utilisations are counters over a duration: both from the profiled (serialized) run when it recorded one
            ser_ms = (counters or {}).get("_serialized_ms")
            bind_ms = ser_ms or avg_ms
            bind = binding_model(counters, bind_ms, torch.cuda.get_device_properties(dev).multi_processor_count)
            if bind is not None and traffic:
                bind["hbm_frac"] = traffic / (bind_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if bind is not None:
                bind["duration_ms_used"] = bind_ms
                bind["duration_source"] = ("kernel trace of the counter pass (kernels one at a time)" if ser_ms else
                                           "HIP events of the timed region (no trace in the counter pass)")
            if bind is not None:
                cand = {k: bind[k] for k in ("lds_frac", "issue_frac", "hbm_frac") if bind.get(k) is not None}
                bind["nearest"] = max(cand, key=cand.get) if cand else None
            # what bounds the kernel: the resource nearest to saturation by the counters (an LDS-resident solver: its rows never
            # leave the CU, so it is not HBM; without counters the committed profiles say "lds")
            bound = {"lds_frac": "lds", "issue_frac": "valu_issue", "hbm_frac": "hbm"}.get((bind or {}).get("nearest"), "lds")
            roof = {"bound": bound, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": traffic,
                    "traffic_as_reported": traffic_raw,
                    "frac_physical": (traffic / (bind_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    # the fractions that say what the kernel is bound by, first class: the LDS pipeline of an LDS-resident solver
                    "lds": ({"busy_frac": bind.get("lds_frac"), "conflict_share": bind.get("lds_bank_conflict_share"),
                             "useful_frac": (bind["lds_frac"] * (1.0 - bind["lds_bank_conflict_share"])
                                             if bind.get("lds_frac") is not None and bind.get("lds_bank_conflict_share") is not None else None)}
                            if bind else None),
                    "avg_launch_ms_serialized": ser_ms, "avg_launch_ms_corunning": avg_ms,
                    "kernel": "nrq_solve_kernel<%d, %d, %d>" % (enc_stats["strip_bytes"], enc_stats["wg_threads"], enc_stats["wg_waves_per_simd"]),
                    "avg_launch_ms": avg_ms, "busy_ms_per_launch": eff_ms, "launches_timed": len(ktimes),
                    "blocks_per_launch": blocks_per_launch,
                    "traffic_source": csrc,
                    "traffic_detail": ({"read": counters.get("FETCH_SIZE"), "written": counters.get("WRITE_SIZE"),
                                        "note": "counters as reported; `traffic` = 2 x read + written (MI355X_MICROARCH.md: on gfx950 "
                                                "FETCH_SIZE counts 16-byte-per-lane reads at half their bytes)"} if counters else None),
                    "compulsory": compulsory, "staging": staging,
                    "traffic_over_compulsory": traffic / compulsory if traffic else None,
                    "traffic_over_compulsory_plus_staging": traffic / (compulsory + staging) if traffic else None,
                    "binding": bind}
            if balg_enc is not None:
                alg_per_launch = 0.5 * (balg_enc + balg_dec) * blocks_per_launch
                achieved = alg_per_launch / (eff_ms * 1e-3) / 1e9
                roof.update({"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_per_launch,
                             "work_rate": {"value": achieved, "unit": "GB/s of reference-equivalent row traffic (SURVEY 8d)",
                                           "over_hbm_peak": achieved / HBM_PEAK_GBS},
                             "algorithmic_bytes_per_block": {"encode": balg_enc, "decode": balg_dec},
                             "note": "`bound` names the PHYSICAL resource nearest to saturation by the counters (lds = roofline.lds.busy_frac, "
                                     "valu_issue = binding.issue_frac, hbm = frac_physical); peak / frac are SURVEY 8(d)'s convention: "
                                     "achieved/frac (= work_rate) = reference-equivalent row traffic (SURVEY 8d) over the launch "
                                     "duration: a work rate, not an HBM utilisation (the strip solver keeps rows in LDS, so it "
                                     "exceeds 1); `bound` names the resource nearest to saturation, `frac_physical` the physical HBM "
                                     "traffic (FETCH_SIZE doubled as the guide prescribes) over duration x 8 TB/s; all utilisations "
                                     "are in `binding`. avg_launch_ms is the per-launch HIP-event duration (what rocprofv3 "
                                     "--kernel-trace reports)"})
        out = {
            "metric": "Gbit/s encode+decode, K=%d T=%d" % (K, T), "value": value, "unit": "Gbit/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: K=%d T=%d, %d blocks/GPU/step, %.0f%% loss, overhead %d, encode(+%d repair)"
                                   "+decode" % (config_name(K, T, args.loss, args.overhead), K, T, NB, args.loss * 100, args.overhead,
                                                nrep),
                       "K": K, "T": T, "blocks_per_gpu": NB, "loss": args.loss, "overhead": args.overhead,
                       "reception_patterns": NPAT,
                       "decode_launch_widths": lists,   # {strip bytes of the (first) list: launches}; second lists = blocks whose image needs a narrower strip
                       "repair_per_block": nrep, "sharding": "blocks over GPUs, no collective",
                       "streams_per_gpu": nstreams,
                       "encode_plan": "cached" if args.no_replan else "rebuilt every step",
                       "planner": ("device (nrq_plan_kernel, one workgroup per block)" if dec_stats["planner"] else
                                   "host, %d threads/rank" % threads),
                       "host_planned_blocks": dec_stats.get("host_planned", 0),
                       "decode_plan": ("issued %d step(s) ahead (nrq_decode_plan_ahead) from the reception pattern of the step it is for "
                                       "(%d patterns in turn): every step runs one planner pass per block, beside the solves of the "
                                       "steps before it" % (ahead_depth, NPAT) if plan_ahead else
                                       "inside the decode call"),
                       "decode_found_plan_ahead": bool(dec_stats.get("plan_ahead", 0)),
                       "ms_per_step_plans_issued_ahead": ms_plan_ahead,
                       "ms_per_step_plan_inside_decode_call": ms_plan_in_call,
                       "value_plan_inside_decode_call": (8.0 * payload_step / (ms_plan_in_call * 1e-3) / 1e9) if ms_plan_in_call else None,
                       "value_plans_issued_ahead": (8.0 * payload_step / (ms_plan_ahead * 1e-3) / 1e9) if ms_plan_ahead else None,
                       "decode_retries": retries_total, "spare_symbols_taken": retries_total,
                       "in_step": "damage of the receiver's copy (every lost row overwritten over its full width: %.0f MB of writes), "
                                  "poisoning of %d repair + %d intermediate rows of EVERY block and a digest of them and of %d decoded "
                                  "rows per block (~0.1 ms per step together)" % (float(nlost.sum()) * T / 1e6, REP_S, INT_S, WRK_S)},
            "check": check,
            "roofline": roof, "e2e": e2e, "one_object": one_object, "cpu_baseline": cpu,
            "detail": {"solve_kernel_ms_sum_per_step": sum(ktimes) / args.steps,
                       # CPU time this rank's process spent per step (every thread; the barrier's wait included when it spins):
                       # what N ranks on N GPUs ask of the host is N x this, on N disjoint core sets (shard.spawn_ranks)
                       "host_cpu_ms_per_step_rank0": host_cpu_ms,
                       "rank_cpus": os.environ.get("NANORQ_RANK_CPUS", ""),
                       "planner_ms": (sum(ptimes) / len(ptimes)) if ptimes else None,
                       # the solve launches of a step are [encode, decode] per stream group, in that order
                       "encode_solve_ms": (sum(ktimes[0::2]) / max(1, len(ktimes[0::2]))) if nstreams == 1 else None,
                       "decode_solve_ms": (sum(ktimes[1::2]) / max(1, len(ktimes[1::2]))) if nstreams == 1 else None,
                       "encode_gbps_device": (8.0 * NB * K * T / (sum(ktimes[0::2]) / max(1, len(ktimes[0::2])) * 1e-3) / 1e9)
                       if nstreams == 1 and ktimes else None,
                       "decode_gbps_device_incl_planner": (8.0 * NB * K * T / ((elapsed / args.steps - sum(ktimes[0::2]) / max(
                           1, len(ktimes[0::2])) * 1e-3)) / 1e9) if nstreams == 1 and ktimes and world == 1 else None,
                       "encode": {k: enc_stats[k] for k in ("plan_ms", "host_ms", "strip_bytes", "lds_bytes", "grid",
                                                            "wg_threads", "strips_per_slot", "npiv", "u", "nlev")},
                       "decode": {k: dec_stats[k] for k in ("plan_ms", "host_ms", "strip_bytes", "lds_bytes", "grid",
                                                            "wg_threads", "strips_per_slot", "npiv", "u", "nlev", "planner")}},
        }
        print(json.dumps(out))
    shard.finalize(world)


if __name__ == "__main__":
    main()
