"""ONE object spread over the GPUs of ONE process (reference lib/nanorq.c:97-112: the source blocks of an object
share nothing): block sbn lives on device sbn mod N, the batched calls of include/nanorq_batch.h run one host thread per
device with that device's streams and page-locked staging.  Host buffers in, host buffers out (PCIe both ways).

    python tools/bench_one_object.py --devices 0,1,2,3 [--K 56403 --T 1280 --blocks 32 --loss 0.2]

Prints one JSON line: payload Gbit/s per leg (generate / repair symbols / ingest / repair) and for sender, receiver and the
whole transfer; `ok` = the recovered object equals the source.  bench.py runs it from rank 0 for `--gpus N` > 1 (all N
devices in rank 0's process, beside the process-sharded line) and on request at N = 1 (`--one-object`).
The device list must be in the environment before the library's first call: this script sets NANORQ_HIP_DEVICES itself."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="0")
ap.add_argument("--K", type=int, default=56403)
ap.add_argument("--T", type=int, default=1280)
ap.add_argument("--blocks", type=int, default=0, help="source blocks of the object (default: 8 per device)")
ap.add_argument("--loss", type=float, default=0.2)
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
os.environ["NANORQ_HIP_DEVICES"] = args.devices
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))]

import numpy as np  # noqa: E402
from capi import api  # noqa: E402
from object_api_legs import run_pinned  # noqa: E402
from util import loss_pattern  # noqa: E402

ndev = len([d for d in args.devices.split(",") if d.strip() != ""])
K, T = args.K, args.T
Z = args.blocks or 8 * ndev
assert Z <= 256
L = api()
got = int(L.nanorq_devices())
lost = [loss_pattern(K, args.loss, seed=1000, block=b) for b in range(Z)]
# payload: a 16 MiB random tile repeated, every block XORed with its number (cheap to make, no two blocks alike)
t0 = time.perf_counter()
tile = np.random.default_rng(7).integers(0, 256, 16 << 20, dtype=np.uint8)
data = np.empty(Z * K * T, np.uint8)
per = K * T
for b in range(Z):
    blk = data[b * per:(b + 1) * per]
    for o in range(0, per, tile.size):
        n = min(tile.size, per - o)
        np.bitwise_xor(tile[:n], np.uint8((b * 37 + o // tile.size) & 0xFF), out=blk[o:o + n])
t_data = time.perf_counter() - t0
run_pinned(K, T, min(Z, max(ndev, 2)), lost, data=data[:min(Z, max(ndev, 2)) * per])   # warm-up: contexts, constants, plans, pools
legs = run_pinned(K, T, Z, lost, data=data, reps=args.reps)
print(json.dumps({"devices_asked": args.devices, "devices": got, "K": K, "T": T, "blocks": Z, "loss": args.loss,
                  "payload_bytes": int(Z) * per, "ok": legs["ok"], "value": legs["value"], "unit": "Gbit/s",
                  "generate_gbps": legs["generate_gbps"], "repair_symbols_ms": legs["repair_symbols_ms"],
                  "ingest_gbps": legs["add_gbps"], "repair_gbps": legs["repair_gbps"], "sender_gbps": legs["sender_gbps"],
                  "receiver_gbps": legs["receiver_gbps"], "total_ms": legs["total_ms"], "received_symbols": legs["received_symbols"],
                  "what": "ONE object of %d blocks over %d device(s) of one process (block sbn on device sbn mod N, a host "
                          "thread per device, no collective), page-locked host memory in and out; value = payload / (generate + "
                          "repair symbols to host + ingest + repair), the four legs one after the other" % (Z, got)}))
