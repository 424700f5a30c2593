"""Object API end to end (host buffers in, host buffers out, PCIe both ways): the reference's per-block calls
(nanorq.h) against the batched variants (nanorq_batch.h).  Not bench.py's metric -- that one keeps data in HBM.

    python tools/bench_object_api.py [K] [T] [blocks]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from capi import api, mem_io  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
Z = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L = api()
F = K * T * Z
data = np.random.default_rng(1).integers(0, 256, F, dtype=np.uint8)
nrep = max(1, K // 20)
gbit = 8.0 * F / 1e9


def run(batched):
    rq = L.nanorq_encoder_new_ex(F, T, K, 0, 8)
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
    out = np.zeros(F, np.uint8)
    oio = mem_io(out)
    src = data.reshape(Z, K, T)
    t3 = time.perf_counter()
    if batched:
        blob = np.concatenate([np.concatenate([src[s, nrep:], rep[s]]) for s in range(Z)])
        tags = np.concatenate([[L.nanorq_tag(s, e) for e in list(range(nrep, K)) + list(range(K, K + nrep))] for s in range(Z)]).astype(np.uint32)
        t3 = time.perf_counter()
        L.nanorq_decoder_add_symbols(dq, blob.ctypes.data_as(C.c_void_p), tags.ctypes.data_as(C.POINTER(C.c_uint32)), len(tags), None, oio)
    else:
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
    return {"generate_symbols_gbps": gbit / (t1 - t0), "repair_symbols_ms": (t2 - t1) * 1e3, "add_symbols_ms": (t4 - t3) * 1e3,
            "repair_gbps": gbit / (t5 - t4)}


run(True)  # warm-up: context, plan cache
for name, b in (("per-block calls (nanorq.h)", False), ("batched calls (nanorq_batch.h)", True)):
    r = run(b)
    print("%-32s K=%d T=%d blocks=%d: generate %.1f Gbit/s, %d repair symbols/block in %.1f ms, add %.1f ms, repair %.1f Gbit/s"
          % (name, K, T, Z, r["generate_symbols_gbps"], nrep, r["repair_symbols_ms"], r["add_symbols_ms"], r["repair_gbps"]))
