#!/usr/bin/env python3
"""bench.py -- RaptorQ encode+decode throughput of the MI355X path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (config.workload "cfg3"): BASELINE.json configs[2] -- K=8192 source symbols of T=1280 bytes per
source block, 10 % independent random loss per block, decode with exactly K received symbols
(overhead 0: the GF(256)/HDPC path; a block whose matrix is rank deficient is retried with one more
repair symbol inside the timed region and counted).  One STEP = one batch of `--blocks` independent
source blocks per GPU: encode (source -> intermediate symbols in HBM + R repair symbols; the encode
plan is rebuilt every step, i.e. once per 256-block object like nanorq_precalculate) followed by
decode (per-block plan from the loss pattern + solve + regeneration of the missing symbols).
Payloads are synthetic and already resident in HBM when the timed region starts.  Blocks shard
across GPUs with no data-path collective (weak scaling); rank 0 prints ONE JSON line.

`value` = 8 * (payload bytes encoded and decoded) / wall time, whole job.
`roofline`: the solve kernel (nrq_solve_kernel) against the HBM roofline using the ALGORITHMIC
bytes of SURVEY.md section 8(d) -- the row traffic the reference CPU path performs for the same blocks,
counted live by the oracle on the sampled blocks -- divided by the kernel's average launch
duration measured with HIP events on the launch stream.
`cpu_baseline`: the oracle (C restatement of the reference algorithm with AVX2 GF(256) row kernels;
upstream oblas is absent so this is a "port") timed on this host, one core, on a bounded sample of the
same workload.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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
    ap.add_argument("--no-replan", action="store_true", help="keep the encode plan cached across steps")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU: the step's blocks are split into this many groups, each on its own stream, so "
                         "that one group's (latency-bound, 1 workgroup/block) planner kernel runs beside another group's solve")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--force-device", type=int, default=-1,
                    help="plumbing tests only: every rank uses this GPU (several ranks on one device; use with gloo)")
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
            nr_ = len(lost_np[b]) + args.overhead + 2   # +2: never singular, keeps the parallel leg simple
            ok_, _, _ = oracle.decode_block(np.concatenate([keep_, esis[:nr_]]), np.concatenate([src_np[b][keep_], rep_[:nr_]]),
                                            K, T)
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
        "host_cpu": _cpu_model(), "host_threads": os.cpu_count(), "all_cores": allc,
    }, balg_enc / n, balg_dec / n


def pmc_traffic(args):
    """HBM bytes per solve-kernel launch from the committed rocprofv3 --pmc passes (profiles/r1_pmc_hbm.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command; see the file for caveats).
    Only valid for the default workload; None otherwise."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_hbm.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    if (rec.get("K"), rec.get("T"), rec.get("blocks")) != (args.K, args.T, args.blocks):
        return None
    return rec.get("bytes_per_launch")


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    import torch
    import nanorq_amd
    from util import loss_pattern

    from nanorq_amd import shard
    rank, world, local = shard.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.force_device >= 0:
        local = args.force_device
    torch.cuda.set_device(local)
    # RCCL; only the barrier and the timing reduction use it
    shard.init(args.dist_backend, device_id=torch.device("cuda", local) if args.dist_backend == "nccl" else None)
    dev = torch.device("cuda", local)
    threads = args.threads or max(1, (os.cpu_count() or 1) // max(1, world))
    nstreams = max(1, min(args.streams, args.blocks))
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]
    ctxs = [nanorq_amd.Context(local, st.cuda_stream) for st in streams]   # one context per stream, same GPU
    for c_ in ctxs:
        c_.set_threads(threads)
    ctx = ctxs[0]

    K, T, NB = args.K, args.T, args.blocks
    prm = nanorq_amd.params(K)
    L = prm["L"]
    # weak scaling: the job has world*NB source blocks per step, block b lives on GPU b mod world;
    # payload and loss pattern are functions of the GLOBAL block id, payload generated on the device
    my_blocks = shard.blocks_of(rank, world, world * NB)
    g = torch.Generator(device=dev)
    g.manual_seed(1 + rank)
    src = torch.randint(0, 256, (NB, K, T), dtype=torch.uint8, device=dev, generator=g)
    lost = [loss_pattern(K, args.loss, seed=1000, block=gb) for gb in my_blocks]
    max_lost = max(len(x) for x in lost)
    nrep = max_lost + args.overhead + 3  # repair symbols generated per block by the encoder (incl. spares)
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep = torch.empty((NB, nrep, T), dtype=torch.uint8, device=dev)
    inter = torch.empty((NB, L, T), dtype=torch.uint8, device=dev)
    work = src.clone()  # what the receiver holds: source block with the lost rows destroyed
    for b in range(NB):
        work[b, torch.from_numpy(lost[b].astype(np.int64)).to(dev)] = 0xEE
    lost_arr = np.zeros((NB, max_lost + 1), np.uint32)
    for b in range(NB):
        lost_arr[b, :len(lost[b])] = lost[b]
    nlost = np.array([len(x) for x in lost], np.uint32)
    resi = np.tile(esis, (NB, 1))
    retries = 0
    spare = 3   # repair symbols a block may take beyond (lost + overhead) if its system is rank deficient
    nr_first = (nlost + args.overhead).astype(np.uint32)
    nr_avail = (nlost + args.overhead + spare).astype(np.uint32)

    # block ranges of the stream groups
    bounds = [(NB * g_) // nstreams for g_ in range(nstreams + 1)]
    groups = [(bounds[g_], bounds[g_ + 1]) for g_ in range(nstreams)]

    replan_early = L >= 12000   # (the library's threshold for device-built encode plans, NRQ_ENCPLAN_DEV_MIN_L)

    def step():
        nonlocal retries
        enc_stats = dec_stats = None
        for (lo, hi), c_ in zip(groups, ctxs):
            n_ = hi - lo
            c_.encode_blocks(K, T, n_, src[lo].data_ptr(), K * T, rep[lo].data_ptr(), nrep * T, esis, inter[lo].data_ptr(),
                             L * T)
            enc_stats = enc_stats or c_.stats()
        if not args.no_replan and replan_early:
            # big K': the NEXT step's encode plan is built by the device planner, asynchronously on a stream of its own
            # (nrq_precalculate only enqueues it) -- issued before the decode so that it runs beside this step's work
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
            if not st.all():
                raise RuntimeError("decode failed for %d blocks" % int((st == 0).sum()))
            retries += int((used - nr_first[lo:hi]).sum())
        if not args.no_replan and not replan_early:
            # one encode plan per step (= per 256-block object, like nanorq_precalculate): the plan for the NEXT step's
            # encode is rebuilt on the host here (once per context), while the GPU runs this step's solve
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
    for c_ in ctxs:
        c_.ktime_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc_stats, dec_stats = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # solve-kernel launches of all streams on one time axis (HIP events recorded around each launch)
    intervals = []
    for c_ in ctxs:
        intervals += c_.ktime_read_intervals(ref=ctx)
        c_.ktime_enable(False)
    ktimes = [d for _, d in intervals]
    rdev = dev if args.dist_backend == "nccl" else None
    elapsed = shard.reduce_max(elapsed, world, device=rdev)   # the slowest rank defines the step time
    retries_total = int(shard.reduce_sum(retries, world, device=rdev))

    # correctness of what was timed: every block decoded back to its source
    assert torch.equal(work, src), "decoded blocks differ from the source blocks"

    if rank == 0:
        payload_step = world * NB * K * T
        value = 8.0 * payload_step * args.steps / elapsed / 1e9
        cpu, balg_enc, balg_dec = (None, None, None)
        if args.cpu_sample > 0:
            src_np = src[:args.cpu_sample].cpu().numpy()
            cpu, balg_enc, balg_dec = cpu_baseline(args, src_np, lost, nrep)
        # solve-kernel launches in the timed region: [encode, decode(, retries...)] per step
        roof = None
        if balg_enc is not None and ktimes:
            # every step launches the solve kernel 2*nstreams times (encode and decode of each stream group); launches of
            # different streams overlap in time, so the kernel's rate is the algorithmic bytes of all launches divided by
            # the time during which at least one of them was running (= the plain average duration when nstreams == 1)
            avg_ms = sum(ktimes) / len(ktimes)
            busy, cur_s, cur_e = 0.0, None, None
            for s0, d0 in sorted(intervals):
                if cur_e is None or s0 > cur_e:
                    busy += 0.0 if cur_e is None else cur_e - cur_s
                    cur_s, cur_e = s0, s0 + d0
                else:
                    cur_e = max(cur_e, s0 + d0)
            busy += 0.0 if cur_e is None else cur_e - cur_s
            blocks_per_launch = NB / float(nstreams)
            alg_per_launch = 0.5 * (balg_enc + balg_dec) * blocks_per_launch
            eff_ms = busy / len(ktimes)   # busy time attributable to one launch
            achieved = alg_per_launch / (eff_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args), "kernel": "nrq_solve_kernel<%d, %d, %d>" %
                    (enc_stats["strip_bytes"], enc_stats["wg_threads"], enc_stats["wg_waves_per_simd"]), "avg_launch_ms": avg_ms, "busy_ms_per_launch": eff_ms,
                    "launches_timed": len(ktimes), "blocks_per_launch": blocks_per_launch,
                    "algorithmic_bytes_per_launch": alg_per_launch,
                    "algorithmic_bytes_per_block": {"encode": balg_enc, "decode": balg_dec},
                    "note": "algorithmic bytes = reference-equivalent row traffic (SURVEY 8d), not physical HBM bytes: the "
                            "strip solver keeps rows in LDS. avg_launch_ms is the per-launch HIP-event duration (what rocprofv3 "
                            "--kernel-trace reports); with %d streams launches overlap, so `achieved` divides by the union of "
                            "the launches' busy time per launch" % nstreams}
        out = {
            "metric": "Gbit/s encode+decode, K=%d T=%d" % (K, T), "value": value, "unit": "Gbit/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "cfg3: K=%d T=%d, %d blocks/GPU/step, %.0f%% loss, overhead %d, encode(+%d repair)"
                                   "+decode" % (K, T, NB, args.loss * 100, args.overhead, nrep),
                       "K": K, "T": T, "blocks_per_gpu": NB, "loss": args.loss, "overhead": args.overhead,
                       "repair_per_block": nrep, "sharding": "blocks over GPUs, no collective",
                       "streams_per_gpu": nstreams,
                       "encode_plan": "cached" if args.no_replan else "rebuilt every step",
                       "planner": ("device (nrq_plan_kernel, one workgroup per block)" if dec_stats["planner"] else
                                   "host, %d threads/rank" % threads),
                       "decode_retries": retries_total, "spare_symbols_taken": retries_total},
            "roofline": roof, "cpu_baseline": cpu,
            "detail": {"solve_kernel_ms_sum_per_step": sum(ktimes) / args.steps,
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
