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

argv = [x for x in sys.argv[1:] if not x.startswith("--")]
K = int(argv[0]) if len(argv) > 0 else 8192
T = int(argv[1]) if len(argv) > 1 else 1280
Z = int(argv[2]) if len(argv) > 2 else 128
loss = float(argv[3]) if len(argv) > 3 else 0.1
lost = [loss_pattern(K, loss, seed=1000, block=b) for b in range(Z)]
data = np.random.default_rng(1).integers(0, 256, K * T * Z, dtype=np.uint8)
run_pinned(K, T, min(Z, 16), lost, data=data[:min(Z, 16) * K * T])
a = run_pinned(K, T, Z, lost, data=data, reps=2 if "--legs" in sys.argv else 3)
# (the first deferred ingestion of a process allocates its staging -- a device buffer for the whole packet stream, page-locked
# address lists --; those come from the library's pools afterwards: best of three by the receiver's rate)
b = run_pinned(K, T, Z, lost, data=data, reps=3, async_ingest=True, key="receiver_gbps")
assert a["ok"] and b["ok"]
if "--legs" in sys.argv:   # everything, for bench.py's `e2e` entry
    print(json.dumps({"K": K, "T": T, "blocks": Z, "loss": loss, "serial": a, "pipeline": b}))
    sys.exit(0)
print(json.dumps({"K": K, "T": T, "blocks": Z, "loss": loss,
                  "serial": {k: round(a[k], 1) for k in ("generate_gbps", "add_gbps", "repair_gbps", "sender_gbps", "receiver_gbps")},
                  "pipeline": {"add_ms": round(8e-6 * Z * K * T / b["add_gbps"], 2), "repair_ms": round(8e-6 * Z * K * T / b["repair_gbps"], 2),
                               "receiver_gbps": round(b["receiver_gbps"], 1)}}))
