"""Object API end to end (host buffers in, host buffers out, PCIe both ways): the reference's per-block calls
(nanorq.h), the batched variants (nanorq_batch.h) on ordinary memory, and the batched variants on page-locked
memory (DMA both ways, upload / solve / download overlapped).  Not bench.py's metric -- that one keeps data in HBM.

    python tools/bench_object_api.py [K] [T] [blocks] [--json]

generate  = nanorq_generate_symbols(_all): object -> GPU, solve, intermediate symbols resident    [Gbit/s of payload]
repair    = nanorq_repair_block / _all: (upload,) decode, recovered object written to the sink     [Gbit/s of payload]
add       = nanorq_decoder_add_symbol(s): ingestion of the received packets                        [ms, and Gbit/s]
PCIe Gen5 x16 moves ~55 GB/s per direction in practice = 440 Gbit/s: the ceiling of every leg that crosses it once.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from capi import api, mem_io, pinned_array, pinned_io  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(argv[0]) if len(argv) > 0 else 1000
T = int(argv[1]) if len(argv) > 1 else 1280
Z = int(argv[2]) if len(argv) > 2 else 64
L = api()
F = K * T * Z
data = np.random.default_rng(1).integers(0, 256, F, dtype=np.uint8)
nrep = max(1, K // 20)
gbit = 8.0 * F / 1e9


def run(mode):
    batched = mode != "per-block"
    pinned = mode == "pinned"
    rq = L.nanorq_encoder_new_ex(F, T, K, 0, 8)
    if pinned:
        io, mem = pinned_io(F)
        mem[:] = data
    else:
        io = mem_io(data)
    L.nanorq_precalculate(rq)
    t0 = time.perf_counter()
    if batched:
        assert L.nanorq_generate_symbols_all(rq, io) == Z
    else:
        for sbn in range(Z):
            assert L.nanorq_generate_symbols(rq, sbn, io)
    t1 = time.perf_counter()
    rep = np.zeros((Z, nrep, T), np.uint8)
    for sbn in range(Z):
        if batched:
            assert L.nanorq_encode_range(rq, rep[sbn].ctypes.data_as(C.c_void_p), K, nrep, sbn, io) == nrep * T
        else:
            for j in range(nrep):
                assert L.nanorq_encode(rq, rep[sbn, j].ctypes.data_as(C.c_void_p), K + j, sbn, io) == T
    t2 = time.perf_counter()
    oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
    L.nanorq_free(rq)
    io.contents.destroy(io)
    # receiver: the first nrep source symbols of every block are lost, the repair symbols arrive instead
    dq = L.nanorq_decoder_new(*oti)
    if pinned:
        oio, out = pinned_io(F)
        out[:] = 0
    else:
        out = np.zeros(F, np.uint8)
        oio = mem_io(out)
    src = data.reshape(Z, K, T)
    addr = None
    if batched:
        n = Z * K
        if pinned:
            addr, blob = pinned_array(n * T)
            blob = blob.reshape(Z, K, T)
        else:
            blob = np.empty((Z, K, T), np.uint8)
        blob[:, :K - nrep] = src[:, nrep:]
        blob[:, K - nrep:] = rep
        tags = np.concatenate([[L.nanorq_tag(s, e) for e in list(range(nrep, K)) + list(range(K, K + nrep))] for s in range(Z)]).astype(np.uint32)
        t3 = time.perf_counter()
        assert L.nanorq_decoder_add_symbols(dq, blob.ctypes.data_as(C.c_void_p), tags.ctypes.data_as(C.POINTER(C.c_uint32)), n, None, oio) == n
    else:
        t3 = time.perf_counter()
        for s in range(Z):
            for e in range(nrep, K):
                L.nanorq_decoder_add_symbol(dq, src[s, e].ctypes.data_as(C.c_void_p), L.nanorq_tag(s, e), oio)
            for j in range(nrep):
                L.nanorq_decoder_add_symbol(dq, rep[s, j].ctypes.data_as(C.c_void_p), L.nanorq_tag(s, K + j), oio)
    t4 = time.perf_counter()
    if batched:
        assert L.nanorq_repair_all(dq, oio) == Z
    else:
        for s in range(Z):
            assert L.nanorq_repair_block(dq, oio, s)
    t5 = time.perf_counter()
    assert np.array_equal(out, data)
    L.nanorq_free(dq)
    oio.contents.destroy(oio)
    if addr:
        L.nanorq_pinned_free(addr)
    return {"mode": mode, "K": K, "T": T, "blocks": Z, "generate_gbps": gbit / (t1 - t0), "repair_symbols_ms": (t2 - t1) * 1e3,
            "add_ms": (t4 - t3) * 1e3, "add_gbps": gbit / (t4 - t3), "repair_gbps": gbit / (t5 - t4)}


run("pinned")  # warm-up: context, plan cache, pool
recs = [run(m) for m in ("per-block", "batched", "pinned")]
if "--json" in sys.argv:
    print(json.dumps({"K": K, "T": T, "blocks": Z, "payload_bytes": F, "repair_symbols_per_block": nrep, "results": recs}))
else:
    names = {"per-block": "per-block calls (nanorq.h)", "batched": "batched calls (nanorq_batch.h)", "pinned": "batched + page-locked memory"}
    for r in recs:
        print("%-30s K=%d T=%d blocks=%d: generate %.1f Gbit/s, %d repair symbols/block in %.1f ms, add %.1f ms (%.0f Gbit/s), repair %.1f Gbit/s"
              % (names[r["mode"]], K, T, Z, r["generate_gbps"], nrep, r["repair_symbols_ms"], r["add_ms"], r["add_gbps"], r["repair_gbps"]))
