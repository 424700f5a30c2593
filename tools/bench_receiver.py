"""The receiver's two stations (ingest, repair) of one object through the object API on page-locked memory, serial
(nanorq_decoder_add_symbols, then nanorq_repair_all) and as a pipeline (nanorq_decoder_add_symbols_async): one JSON line.
    python tools/bench_receiver.py [K] [T] [blocks] [loss]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402
from object_api_legs import run_pinned  # noqa: E402
from util import loss_pattern  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
Z = int(sys.argv[3]) if len(sys.argv) > 3 else 128
loss = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
lost = [loss_pattern(K, loss, seed=1000, block=b) for b in range(Z)]
data = np.random.default_rng(1).integers(0, 256, K * T * Z, dtype=np.uint8)
run_pinned(K, T, min(Z, 16), lost, data=data[:min(Z, 16) * K * T])
a = run_pinned(K, T, Z, lost, data=data, reps=3)
b = run_pinned(K, T, Z, lost, data=data, reps=3, async_ingest=True)
assert a["ok"] and b["ok"]
print(json.dumps({"K": K, "T": T, "blocks": Z, "loss": loss,
                  "serial": {k: round(a[k], 1) for k in ("generate_gbps", "add_gbps", "repair_gbps", "sender_gbps", "receiver_gbps")},
                  "pipeline": {"add_ms": round(8e-6 * Z * K * T / b["add_gbps"], 2), "repair_ms": round(8e-6 * Z * K * T / b["repair_gbps"], 2),
                               "receiver_gbps": round(b["receiver_gbps"], 1)}}))
