"""The object API's page-locked path timed leg by leg, for bench.py's `e2e` entry and tools/bench_object_api.py:
host buffers in, host buffers out, PCIe both ways, upload / solve / download overlapped inside the library
(include/nanorq_batch.h).  One object of Z source blocks of K symbols of T bytes; block b loses the source symbols
lost[b] and receives len(lost[b]) + spare repair symbols instead (the object layer takes gaps + 2 up front and holds the
rest in reserve, nanorq_api.c rep_upfront)."""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (the host process's to set before its first HIP call: include/nanorq.h)
import ctypes as C
import time

import numpy as np

from capi import api, pinned_array, pinned_io


def run_pinned(K, T, Z, lost, data=None, spare=2, reps=1, async_ingest=False, key="value"):
    """async_ingest: the packets go in through nanorq_decoder_add_symbols_async (enqueue only): `add` is then the host's
    bookkeeping alone and `repair` contains the wait for the bytes -- only receiver_gbps (= payload / (add + repair)) compares
    with the serial form."""
    L = api()
    F = K * T * Z
    if data is None:
        data = np.random.default_rng(1).integers(0, 256, F, dtype=np.uint8)
    gbit = 8.0 * F / 1e9
    best = None
    for _ in range(reps):
        nrep = [len(lost[b]) + spare for b in range(Z)]
        nmax = max(nrep)
        raddr, rbuf = pinned_array(Z * nmax * T)       # (page-locked target of the repair symbols: allocated outside the timed legs)
        # the sender as ONE call: nanorq_encode_range_all on an object none of whose blocks is solved yet runs upload, solve,
        # repair-symbol generation and the way back of the symbols as one pipeline (both directions of the link at once)
        rq = L.nanorq_encoder_new_ex(F, T, K, 0, 8)
        assert rq and L.nanorq_blocks(rq) == Z
        io, mem = pinned_io(F)
        mem[:] = data
        L.nanorq_precalculate(rq)
        tf0 = time.perf_counter()
        assert L.nanorq_encode_range_all(rq, C.c_void_p(raddr), K, nmax, io) == Z * nmax * T
        t_fused = time.perf_counter() - tf0
        fused_rep = rbuf.copy()
        L.nanorq_free(rq)
        # the two legs one after the other
        rbuf[:] = 0
        rq = L.nanorq_encoder_new_ex(F, T, K, 0, 8)
        L.nanorq_precalculate(rq)
        t0 = time.perf_counter()
        assert L.nanorq_generate_symbols_all(rq, io) == Z
        t1 = time.perf_counter()
        assert L.nanorq_encode_range_all(rq, C.c_void_p(raddr), K, nmax, io) == Z * nmax * T
        rep = [rbuf.reshape(Z, nmax, T)[b, :nrep[b]] for b in range(Z)]
        t2 = time.perf_counter()
        assert np.array_equal(fused_rep, rbuf), "the pipelined sender produced other repair symbols than the two calls"
        oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
        L.nanorq_free(rq)
        io.contents.destroy(io)
        # receiver
        dq = L.nanorq_decoder_new(*oti)
        oio, out = pinned_io(F)
        out[:] = 0
        src = data.reshape(Z, K, T)
        n = sum(K - len(lost[b]) + nrep[b] for b in range(Z))
        addr, blob = pinned_array(n * T)
        blob = blob.reshape(n, T)
        tags = np.empty(n, np.uint32)
        at = 0
        for b in range(Z):
            keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost[b])
            m = len(keep)
            blob[at:at + m] = src[b][keep]
            tags[at:at + m] = (b << 24) | keep
            blob[at + m:at + m + nrep[b]] = rep[b]
            tags[at + m:at + m + nrep[b]] = (b << 24) | (K + np.arange(nrep[b], dtype=np.uint32))
            at += m + nrep[b]
        t3 = time.perf_counter()
        add = L.nanorq_decoder_add_symbols_async if async_ingest else L.nanorq_decoder_add_symbols
        added = add(dq, blob.ctypes.data_as(C.c_void_p), tags.ctypes.data_as(C.POINTER(C.c_uint32)), n, None, oio)
        t4 = time.perf_counter()
        done = L.nanorq_repair_all(dq, oio)
        t5 = time.perf_counter()
        ok = added == n and done == Z and np.array_equal(out, data)
        L.nanorq_free(dq)
        oio.contents.destroy(oio)
        L.nanorq_pinned_free(addr)
        L.nanorq_pinned_free(raddr)
        legs = {"generate_gbps": gbit / (t1 - t0), "repair_symbols_ms": (t2 - t1) * 1e3, "add_gbps": gbit / (t4 - t3),
                "repair_gbps": gbit / (t5 - t4), "total_ms": ((t2 - t0) + (t5 - t3)) * 1e3,
                "value": gbit / ((t2 - t0) + (t5 - t3)), "sender_gbps": gbit / t_fused, "sender_gbps_two_calls": gbit / (t2 - t0), "sender_ms": t_fused * 1e3, "receiver_gbps": gbit / (t5 - t3),
                "ok": bool(ok), "received_symbols": int(n)}
        if best is None or legs[key] > best[key]:
            best = legs
    return best
