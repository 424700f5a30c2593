#!/usr/bin/env python3
"""Runs ON THE GPU BOX: decode-failure statistics of the HIP path against RFC 6330's design figures.

RaptorQ's selling point (RFC 6330 section 1; Shokrollahi & Luby) is that a block decodes from K' + o received symbols with
failure probability ~1 % / ~0.01 % / ~1e-4 % for o = 0 / 1 / 2, whatever the reception pattern.  The verdict of a decode is
exactly rank(A) < L of the constraint matrix that the tuple / LDPC / HDPC generators and J(K') define, so these rates are a
reference-independent check of the whole generator restatement AND of the device planner's rank verdict: a wrong generator
or a planner that loses rank shows as a rate far off the figures.  (The reference has the same verdict, lib/nanorq.c:620-623,
lib/precode.c:264-315.)

    python tools/failure_rates.py [K=100,1000] [receptions=100000] [loss=0.1,0.3,0.5]  -> one line per (K, overhead), JSON at the end
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import nanorq_amd  # noqa: E402


def rates(ctx, dev, K, T, nrec, losses, overheads=(0, 1, 2), seed=2026):
    NB = int(min(8192, max(256, (64 << 20) // (K * T))))
    rng = np.random.default_rng(seed + K)
    src = torch.randint(0, 256, (NB, K, T), dtype=torch.uint8, device=dev)
    out = {}
    for oh in overheads:
        done = failed = wrong = 0
        while done < nrec:
            p = losses[(done // NB) % len(losses)]
            lost = [np.nonzero(rng.random(K) < p)[0].astype(np.uint32) for _ in range(NB)]
            lost = [x if len(x) else np.array([int(rng.integers(K))], np.uint32) for x in lost]
            ml = max(len(x) for x in lost)
            nrep = ml + oh
            # repair ESIs drawn at random from a window (not the first ones every time): the pattern of received REPAIR symbols varies too
            esis = np.sort(rng.choice(np.arange(K, K + 4 * nrep + 16, dtype=np.uint32), nrep, replace=False)).astype(np.uint32)
            rep = torch.empty((NB, nrep, T), dtype=torch.uint8, device=dev)
            ctx.encode_blocks(K, T, NB, src.data_ptr(), K * T, rep.data_ptr(), nrep * T, esis, 0, 0)
            work = src.clone()
            la = np.zeros((NB, ml), np.uint32)
            for b in range(NB):
                la[b, :len(lost[b])] = lost[b]
            work.view(NB * K, T)[torch.from_numpy(np.concatenate([b * K + lost[b].astype(np.int64) for b in range(NB)])).to(dev)] = 0xEE
            nl = np.array([len(x) for x in lost], np.uint32)
            st = np.asarray(ctx.decode_blocks(K, T, NB, work.data_ptr(), K * T, la, nl, np.tile(esis, (NB, 1)), nl + oh, rep.data_ptr(), nrep * T))
            torch.cuda.synchronize()
            same = (work == src).flatten(1).all(1).cpu().numpy()
            failed += int((st == 0).sum())
            wrong += int(((st != 0) & ~same).sum())   # reported decoded, bytes differ: must never happen
            done += NB
        out[oh] = {"receptions": done, "failed": failed, "rate": failed / done, "decoded_wrong": wrong}
        print("K=%d overhead %d: %d receptions, %d failed (%.5f %%), %d decoded wrong" % (K, oh, done, failed, 100.0 * failed / done, wrong), flush=True)
    return out


def main():
    Ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "100,1000").split(",")]
    nrec = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    losses = [float(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0.1,0.3,0.5").split(",")]
    dev = torch.device("cuda", 0)
    torch.cuda.init(); torch.empty(1, device=dev)
    ctx = nanorq_amd.Context(0, torch.cuda.current_stream(dev).cuda_stream)
    res = {"rfc6330_design": {"0": 1e-2, "1": 1e-4, "2": 1e-6}, "T": 8, "loss_rates": losses}
    for K in Ks:
        res[str(K)] = rates(ctx, dev, K, 8, nrec, losses)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
