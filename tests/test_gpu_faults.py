"""-m gpu tier: the object layer where it is NOT on its happy path (nanorq_amd/csrc/nanorq_api.c).

* fault injection -- nrq_ctx_set_option("fail_after", n): the n-th checked runtime call of the context fails -- through every
  entry point that moves bytes: the call must report failure by the reference's conventions (false / 0 / NANORQ_SYM_ERR,
  reference include/nanorq.h:10-13), leave the object in a state from which the SAME call succeeds, and the bytes of that
  second attempt must be the right ones;
* the per-block decode loop of the unchanged caller (reference benchmark.c:143-151, decode.c) with the other blocks decoded
  ahead in the first call's device batch;
* the orderings between the upload stream and the sorting stream of the deferred ingestion that the round-4 review found
  unguarded (a host-resident block in a chunk of device-resident ones; repair rows grown under a batch in flight);
* source symbols of a solved block whose io has changed since (reference lib/nanorq.c:410-413).
"""
import ctypes as C

import numpy as np
import pytest

from capi import SYM_ADDED, SYM_ERR, api, encode_object, mem_io, pinned_array, pinned_io
from util import payload

pytestmark = pytest.mark.gpu


def _opt(name, value):
    L = api()
    L.nanorq_hip_option.restype = C.c_int
    L.nanorq_hip_option.argtypes = [C.c_size_t, C.c_char_p, C.c_longlong]
    return L.nanorq_hip_option(0, name.encode(), value)


def _faults():
    return _opt("faults_injected", 0)


def _packets(F, T, K, loss, oh, seed):
    data = payload(F, seed=seed)
    c, s, pk = encode_object(data, T, K=K, loss=loss, overhead=oh, seed=seed)
    return data, c, s, pk


def _feed_pinned(L, dq, pk, T, oio, asynchronous=False, results=True):
    addr, view = pinned_array(len(pk) * T)
    view[:] = np.frombuffer(b"".join(p for _, p in pk), np.uint8)
    tags = np.array([t for t, _ in pk], np.uint32)
    res = np.full(len(pk), 77, np.int32)
    fn = L.nanorq_decoder_add_symbols_async if asynchronous else L.nanorq_decoder_add_symbols
    n = fn(dq, C.c_void_p(addr), tags.ctypes.data_as(C.POINTER(C.c_uint32)), len(pk), res.ctypes.data_as(C.POINTER(C.c_int)) if results else None, oio)
    return n, res, addr


# ------------------------------------------------------------------------------------------ fault injection ----
def _books_match(L, dq, pk, res, K):
    """the decoder holds exactly the symbols it reported as stored: per block, repair count and gaps"""
    for b in range(L.nanorq_blocks(dq)):
        nsrc = sum(1 for (t, _), r in zip(pk, res) if r == SYM_ADDED and t >> 24 == b and (t & 0xFFFFFF) < K)
        nrep = sum(1 for (t, _), r in zip(pk, res) if r == SYM_ADDED and t >> 24 == b and (t & 0xFFFFFF) >= K)
        assert L.nanorq_num_repair(dq, b) == nrep, (b, L.nanorq_num_repair(dq, b), nrep)
        assert L.nanorq_num_missing(dq, b) == L.nanorq_block_symbols(dq, b) - nsrc, b


def test_a_failed_packet_batch_is_taken_back_and_can_be_sent_again():
    """nanorq_decoder_add_symbols on the page-locked path: whatever runtime call fails underneath, the decoder believes in
    exactly the symbols it reported NANORQ_SYM_ADDED for -- when the bytes of the batch did not reach the device rows that
    is none of them: every symbol NANORQ_SYM_ERR, bitmaps and repair counts as before the call -- and the symbols reported
    NANORQ_SYM_ERR, sent again, are stored and the object decodes."""
    L = api()
    T, K = 256, 120
    data, c, s, pk = _packets(5 * K * T - 9, T, K, 0.1, 3, seed=5)
    hit = rolled_back = 0
    for n in list(range(1, 40)) + [48, 64, 96]:
        dq = L.nanorq_decoder_new(c, s)
        oio, out = pinned_io(len(data))
        out[:] = 0
        before = _faults()
        assert _opt("fail_after", n) == 0
        added, res, addr = _feed_pinned(L, dq, pk, T, oio)
        _opt("fail_after", 0)
        hit += _faults() > before
        assert added == int((res == SYM_ADDED).sum()) and set(np.unique(res)) <= {SYM_ADDED, SYM_ERR}, (n, np.unique(res))
        _books_match(L, dq, pk, res, K)
        addr2 = None
        if added != len(pk):
            rolled_back += added == 0
            again = [x for x, r in zip(pk, res) if r == SYM_ERR]
            if n % 2:   # ... through the batch call, or one symbol per call (a block taken back is host-resident again)
                a2, r2, addr2 = _feed_pinned(L, dq, again, T, oio)
                assert a2 == len(again) and (r2 == SYM_ADDED).all(), (n, a2)
            else:
                for t, p in again:
                    assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio) == SYM_ADDED, n
        assert L.nanorq_repair_all(dq, oio) == L.nanorq_blocks(dq)
        assert np.array_equal(out, data), n
        L.nanorq_free(dq)
        L.nanorq_pinned_free(addr)
        if addr2:
            L.nanorq_pinned_free(addr2)
        oio.contents.destroy(oio)
    assert hit >= 20 and rolled_back >= 3, "the sweep did not reach the runtime calls of the batch"


@pytest.mark.parametrize("resident", ["device", "host"])
def test_a_failed_repair_all_leaves_the_blocks_retryable(resident):
    """nanorq_repair_all with a runtime failure somewhere in its pipeline (planner launch, a chunk's solve, the way back):
    fewer blocks complete than the object has, nothing wrong is written, and a second call finishes the object."""
    L = api()
    T, K = 512, 300
    data, c, s, pk = _packets(6 * K * T, T, K, 0.08, 2, seed=9)
    hit = 0
    for n in list(range(1, 60, 2)) + [80, 120, 200]:
        dq = L.nanorq_decoder_new(c, s)
        oio, out = pinned_io(len(data))
        out[:] = 0
        if resident == "device":
            added, res, addr = _feed_pinned(L, dq, pk, T, oio)
            assert added == len(pk)
        else:
            addr = None
            for t, p in pk:
                assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio) == SYM_ADDED
        before = _faults()
        _opt("fail_after", n)
        done = L.nanorq_repair_all(dq, oio)
        _opt("fail_after", 0)
        hit += _faults() > before
        assert done <= L.nanorq_blocks(dq)
        # whatever counts as complete is where the caller looks for it, byte for byte -- also when the failure hit a copy on
        # the way back or the wait for it (a block used to count as complete with its rows still on the device)
        for b in range(L.nanorq_blocks(dq)):
            if L.nanorq_num_missing(dq, b) == 0:
                assert np.array_equal(out[b * K * T:(b + 1) * K * T], data[b * K * T:(b + 1) * K * T]), (n, b)
        assert L.nanorq_repair_all(dq, oio) == L.nanorq_blocks(dq), n
        assert np.array_equal(out, data), n
        L.nanorq_free(dq)
        if addr:
            L.nanorq_pinned_free(addr)
        oio.contents.destroy(oio)
    assert hit >= 10


def test_failed_per_block_calls_report_failure_and_succeed_on_retry():
    """nanorq_generate_symbols / nanorq_encode / nanorq_repair_block, one block per call as the reference's programs make
    them: an injected failure gives false / 0, the retry gives the reference's bytes."""
    L = api()
    T, K = 128, 200
    data = payload(3 * K * T, seed=31)
    c, s, pk = encode_object(data, T, K=K, loss=0.1, overhead=2, seed=3)
    want = dict(pk)
    hit = 0
    for n in range(1, 30):
        rq = L.nanorq_encoder_new_ex(len(data), T, K, 0, 8)
        io = mem_io(data)
        before = _faults()
        _opt("fail_after", n)
        ok = L.nanorq_generate_symbols(rq, 1, io)
        buf = (C.c_uint8 * T)()
        got = L.nanorq_encode(rq, buf, K + 1, 1, io) if ok else 0
        _opt("fail_after", 0)
        hit += _faults() > before
        if not ok or got != T:
            assert got in (0, T)
            assert L.nanorq_encode(rq, buf, K + 1, 1, io) == T   # (solves again if the solve was what failed)
        tag = L.nanorq_tag(1, K + 1)
        if tag in want:
            assert bytes(buf) == want[tag], n
        L.nanorq_free(rq)
        io.contents.destroy(io)
        # decoder side
        dq = L.nanorq_decoder_new(c, s)
        out = np.zeros(len(data), np.uint8)
        oio = mem_io(out)
        for t, p in pk:
            L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio)
        before = _faults()
        _opt("fail_after", n)
        ok0 = L.nanorq_repair_block(dq, oio, 0)
        _opt("fail_after", 0)
        hit += _faults() > before
        if not ok0:
            assert L.nanorq_num_missing(dq, 0) > 0
        for b in range(3):
            assert L.nanorq_repair_block(dq, oio, b), (n, b)
        assert np.array_equal(out, data), n
        L.nanorq_free(dq)
        oio.contents.destroy(oio)
    assert hit >= 10


def test_a_failed_allocation_for_repair_rows_fails_the_batch_cleanly():
    """The device rows of a block's repair symbols are grown once per batch (dev_rep_reserve): when that allocation -- or any
    other -- fails, no symbol of the batch counts; two batches, the second one outgrowing the first one's rows."""
    L = api()
    T, K = 64, 400
    data = payload(K * T, seed=77)
    c, s, pk = encode_object(data, T, K=K, loss=0.5, overhead=4, seed=7)     # ~200 repair symbols: beyond the first allocation of 64
    src = [(t, p) for t, p in pk if (t & 0xFFFFFF) < K]
    rep = [(t, p) for t, p in pk if (t & 0xFFFFFF) >= K]
    hit = rolled = 0
    for n in range(1, 24):
        dq = L.nanorq_decoder_new(c, s)
        oio, out = pinned_io(len(data))
        out[:] = 0
        a1, r1, addr1 = _feed_pinned(L, dq, src + rep[:40], T, oio)
        assert a1 == len(src) + 40
        before = _faults()
        _opt("fail_after", n)
        a2, r2, addr2 = _feed_pinned(L, dq, rep[40:], T, oio)
        _opt("fail_after", 0)
        hit += _faults() > before
        if a2 != len(rep) - 40:
            assert a2 == 0 and (r2 == SYM_ERR).all() and L.nanorq_num_repair(dq, 0) == 40, (n, a2)
            rolled += 1
            L.nanorq_pinned_free(addr2)
            a2, r2, addr2 = _feed_pinned(L, dq, rep[40:], T, oio)
            assert a2 == len(rep) - 40
        assert L.nanorq_repair_all(dq, oio) == 1 and np.array_equal(out, data), n
        L.nanorq_free(dq)
        L.nanorq_pinned_free(addr1)
        L.nanorq_pinned_free(addr2)
        oio.contents.destroy(oio)
    assert hit >= 8 and rolled >= 3


# ------------------------------------------------------------------------- the unchanged caller's decode loop ----
def test_repair_block_decodes_the_other_blocks_ahead_and_commits_them_at_their_own_call():
    """`for sbn: nanorq_repair_block(rq, io, sbn)` (reference benchmark.c:143-151): the first call's device batch also decodes
    the other decodable blocks; until its own call comes a block still counts its gaps and its output rows are untouched --
    what a caller can observe is the reference's sequence.  One block cannot be decoded yet (too few symbols): it is left
    alone, fails at its call, and decodes after the missing packets have arrived.  A symbol that arrives for a block decoded
    ahead changes nothing.  A different output context at the later call receives the rows."""
    L = api()
    T, K, Z = 64, 150, 6
    data, c, s, pk = _packets(Z * K * T, T, K, 0.1, 2, seed=12)
    blk = lambda t: t >> 24          # noqa: E731
    held_back = [(t, p) for t, p in pk if blk(t) == 4 and (t & 0xFFFFFF) >= K]
    late = [(t, p) for t, p in pk if blk(t) == 2][-1:]
    first = [(t, p) for t, p in pk if (t, p) not in held_back and (t, p) not in late]
    dq = L.nanorq_decoder_new(c, s)
    out = np.zeros(len(data), np.uint8)
    oio = mem_io(out)
    for t, p in first:
        assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio) == SYM_ADDED
    gaps = [L.nanorq_num_missing(dq, b) for b in range(Z)]
    assert all(g > 0 for g in gaps)
    assert L.nanorq_repair_block(dq, oio, 0)
    assert L.nanorq_num_missing(dq, 0) == 0 and np.array_equal(out[:K * T], data[:K * T])
    # the other blocks: gaps as before, their missing rows not written yet
    assert [L.nanorq_num_missing(dq, b) for b in range(1, Z)] == gaps[1:]
    assert not np.array_equal(out[K * T:2 * K * T], data[K * T:2 * K * T])
    # a late packet for block 2 (decoded ahead): stored like any other, the block stays recovered
    assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(late[0][1]), late[0][0], oio) == SYM_ADDED
    assert not L.nanorq_repair_block(dq, oio, 4) and L.nanorq_num_missing(dq, 4) == gaps[4]
    # block 3 into ANOTHER output context
    out2 = np.zeros(len(data), np.uint8)
    oio2 = mem_io(out2)
    assert L.nanorq_repair_block(dq, oio2, 3)
    lost3 = [e for e in range(K) if not any(t == L.nanorq_tag(3, e) for t, _ in first)]
    for e in lost3:
        assert np.array_equal(out2[(3 * K + e) * T:(3 * K + e + 1) * T], data[(3 * K + e) * T:(3 * K + e + 1) * T])
    for b in (1, 2, 5):
        assert L.nanorq_repair_block(dq, oio, b) and L.nanorq_num_missing(dq, b) == 0
    for t, p in held_back:
        assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio) == SYM_ADDED
    assert L.nanorq_repair_block(dq, oio, 4)
    out[3 * K * T:4 * K * T] |= out2[3 * K * T:4 * K * T]       # (block 3's recovered rows went to the other context)
    assert np.array_equal(out, data)
    L.nanorq_free(dq)
    oio.contents.destroy(oio)
    oio2.contents.destroy(oio2)


def test_blocks_decoded_ahead_are_committed_by_repair_all_too():
    L = api()
    T, K, Z = 32, 90, 4
    data, c, s, pk = _packets(Z * K * T - 5, T, K, 0.12, 3, seed=8)
    dq = L.nanorq_decoder_new(c, s)
    out = np.zeros(len(data), np.uint8)
    oio = mem_io(out)
    for t, p in pk:
        L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio)
    assert L.nanorq_repair_block(dq, oio, 1)
    assert L.nanorq_repair_all(dq, oio) == Z and np.array_equal(out, data)
    assert all(L.nanorq_repair_block(dq, oio, b) for b in range(Z))
    L.nanorq_free(dq)
    oio.contents.destroy(oio)


# --------------------------------------------------------------------- upload stream against sorting stream ----
def test_a_host_resident_block_in_a_chunk_of_deferred_blocks():
    """Block 0 is fed by per-symbol calls (host-resident), blocks 1.. by nanorq_decoder_add_symbols_async (device-resident,
    sorted into rows on the sorting stream): nanorq_repair_all puts them into ONE solve chunk, and that chunk has to wait
    for the host-resident block's upload AND for the deferred batch's last piece."""
    L = api()
    T, K, Z = 1280, 500, 6
    data, c, s, pk = _packets(Z * K * T, T, K, 0.1, 2, seed=15)
    for _ in range(3):
        dq = L.nanorq_decoder_new(c, s)
        oio, out = pinned_io(len(data))
        out[:] = 0
        for t, p in [x for x in pk if x[0] >> 24 == 0]:
            assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio) == SYM_ADDED
        rest = [x for x in pk if x[0] >> 24 != 0]
        n, res, addr = _feed_pinned(L, dq, rest, T, oio, asynchronous=True)
        assert n == len(rest)
        assert L.nanorq_repair_all(dq, oio) == Z
        assert np.array_equal(out, data)
        L.nanorq_free(dq)
        L.nanorq_pinned_free(addr)
        oio.contents.destroy(oio)


def test_repair_rows_grown_under_a_deferred_batch_in_flight():
    """Two nanorq_decoder_add_symbols_async calls back to back: the second one outgrows the repair rows the first one is
    still being sorted into; the rows held before are copied to the new place only after that sort."""
    L = api()
    T, K = 1280, 2000
    data = payload(2 * K * T, seed=19)
    c, s, pk = encode_object(data, T, K=K, loss=0.3, overhead=3, seed=4)      # ~600 repair symbols per block, first allocation 250
    for _ in range(3):
        dq = L.nanorq_decoder_new(c, s)
        oio, out = pinned_io(len(data))
        out[:] = 0
        rep = [x for x in pk if (x[0] & 0xFFFFFF) >= K]
        src = [x for x in pk if (x[0] & 0xFFFFFF) < K]
        first = src + [x for i, x in enumerate(rep) if i % 3 == 0]
        second = [x for i, x in enumerate(rep) if i % 3 != 0]
        n1, _, a1 = _feed_pinned(L, dq, first, T, oio, asynchronous=True)
        n2, _, a2 = _feed_pinned(L, dq, second, T, oio, asynchronous=True)
        assert n1 == len(first) and n2 == len(second)
        assert L.nanorq_repair_all(dq, oio) == 2
        assert np.array_equal(out, data)
        L.nanorq_free(dq)
        L.nanorq_pinned_free(a1)
        L.nanorq_pinned_free(a2)
        oio.contents.destroy(oio)


# ------------------------------------------------------------------------- source symbols of a solved block ----
def test_source_symbols_of_a_solved_block_do_not_depend_on_the_io_any_more():
    """Reference lib/nanorq.c:410-413: once a block is inverted its source symbols are regenerated from the intermediate
    symbols.  A block solved straight out of the caller's memory (>= 1 MiB: read by DMA, no host copy) must not go back
    to that memory -- scribbled over here, and then not passed at all."""
    L = api()
    T, K = 1280, 1000          # 1.28 MB: the region is page-locked in place and read by DMA
    data = payload(K * T, seed=23)
    keep = data.copy()
    rq = L.nanorq_encoder_new_ex(K * T, T, K, 0, 8)
    io = mem_io(data)
    assert L.nanorq_generate_symbols(rq, 0, io)
    data[:] = 0xA5
    buf = (C.c_uint8 * T)()
    for esi in (0, 1, 499, 999):
        assert L.nanorq_encode(rq, buf, esi, 0, io) == T
        assert bytes(buf) == keep[esi * T:(esi + 1) * T].tobytes(), esi
    assert L.nanorq_encode(rq, buf, 7, 0, None) == T and bytes(buf) == keep[7 * T:8 * T].tobytes()
    L.nanorq_free(rq)
    io.contents.destroy(io)


def test_fault_injection_is_a_test_facility_only():
    """A process that did not ask for it (no NANORQ_HIP_FAULT_INJECT=1 when its context is created) cannot make the library's
    runtime calls fail through the public nanorq_hip_option: "fail_after" is an unknown option there, and the library leaves
    the process environment alone (no GPU_MAX_HW_QUEUES appears unless NANORQ_HIP_SET_ENV=1 asks for it)."""
    import os
    import subprocess
    import sys
    code = ("import os, sys, ctypes as C\n"
            "sys.path[:0] = [%r, %r]\n"
            "L = C.CDLL(os.path.join(sys.path[0], 'nanorq_amd', 'libnanorq_hip.so'))\n"   # (the library alone: the python package, a host, sets the variable itself)

            "L.nanorq_hip_option.restype = C.c_int\n"
            "L.nanorq_hip_option.argtypes = [C.c_size_t, C.c_char_p, C.c_longlong]\n"
            "L.nanorq_devices.restype = C.c_size_t\n"
            "assert L.nanorq_devices() >= 1\n"
            "g = C.CDLL(None).getenv; g.restype = C.c_char_p; g.argtypes = [C.c_char_p]\n"   # (the C environment, not python's copy)
            "v = g(b'GPU_MAX_HW_QUEUES')\n"
            "print(L.nanorq_hip_option(0, b'fail_after', 3), v.decode() if v else None)\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    env = {k: v for k, v in os.environ.items() if k not in ("NANORQ_HIP_FAULT_INJECT", "GPU_MAX_HW_QUEUES", "NANORQ_HIP_SET_ENV")}
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert r.stdout.decode().split() == ["-1", "None"], r.stdout
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, NANORQ_HIP_FAULT_INJECT="1", NANORQ_HIP_SET_ENV="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert r.stdout.decode().split() == ["0", "8"], r.stdout
