"""-m gpu tier for the drop-in object API (include/nanorq.h): what the reference's encode.c /
decode.c / benchmark.c do, through the same calls, with the solve on the GPU."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import pytest

from capi import encode_object_batched, decode_object_batched, api, decode_object, encode_object, mem_io
from test_oracle_kat import KAT_SHA, KAT_SMALL
from util import kat_payload, payload

pytestmark = pytest.mark.gpu


def test_kat_through_object_api():
    L = api()
    data = kat_payload(80)
    rq = L.nanorq_encoder_new_ex(80, 8, 10, 0, 8)
    io = mem_io(data)
    buf = (C.c_uint8 * 8)()
    got = {}
    for esi in (10, 11, 12):
        assert L.nanorq_encode(rq, buf, esi, 0, io) == 8  # solves the block on first repair request
        got[esi] = bytes(buf).hex()
    assert got == KAT_SMALL
    assert L.nanorq_encode(rq, buf, 1 << 24, 0, io) == 0  # ESI out of range
    L.nanorq_free(rq)
    io.contents.destroy(io)


@pytest.mark.parametrize("K,T,lo,hi,sha", KAT_SHA[:2])
def test_kat_sha_through_object_api(K, T, lo, hi, sha):
    L = api()
    data = kat_payload(K * T)
    rq = L.nanorq_encoder_new_ex(K * T, T, K, 0, 8)
    io = mem_io(data)
    assert L.nanorq_precalculate(rq) and L.nanorq_generate_symbols(rq, 0, io)
    buf = (C.c_uint8 * T)()
    h = hashlib.sha256()
    for esi in range(lo, hi):
        assert L.nanorq_encode(rq, buf, esi, 0, io) == T
        h.update(bytes(buf))
    assert h.hexdigest() == sha
    # systematic symbols after the solve are the source symbols
    assert L.nanorq_encode(rq, buf, 5, 0, io) == T and bytes(buf) == data[5 * T:6 * T].tobytes()
    L.nanorq_free(rq)
    io.contents.destroy(io)


@pytest.mark.parametrize("K,T,overhead", [(100, 1024, 0), (100, 1024, 5), (1000, 1280, 50), (500, 64, 2)])
def test_benchmark_shaped_roundtrip(K, T, overhead):
    """reference benchmark.c: one block, 6 % loss, `overhead` extra repair symbols, assert(in == out)"""
    data = payload(K * T, seed=K)
    done = False
    for seed in range(1, 5):  # ~1 % of exactly-K receptions are rank deficient: then the harness retries
        c, s, packets = encode_object(data, T, K=K, loss=0.06, overhead=overhead, seed=seed)
        ok, out = decode_object(c, s, packets, len(data))
        if ok:
            assert np.array_equal(out, data)
            done = True
            break
    assert done


def test_multi_block_object_with_ragged_tail():
    """several source blocks of unequal size, object length not a multiple of T (example.make shape)"""
    T = 64
    F = 103 * T - 17
    data = payload(F, seed=3)
    c, s, packets = encode_object(data, T, K=25, loss=0.1, overhead=3, seed=2)
    ok, out = decode_object(c, s, packets, F)
    assert ok and np.array_equal(out, data)


@pytest.mark.parametrize("F,T,K,loss,oh", [(103 * 64 - 17, 64, 25, 0.1, 3), (700 * 1280, 1280, 100, 0.08, 2), (3000 * 256 + 5, 256, 0, 0.05, 4)])
def test_batched_object_api_matches_the_per_block_calls(F, T, K, loss, oh):
    """include/nanorq_batch.h: every block of the object in one device batch -- the packets must be byte-identical
    to the per-block / per-symbol calls, and either decoder must recover the object from either packet set."""
    data = payload(F, seed=13)
    c1, s1, p1 = encode_object(data, T, K=K, loss=loss, overhead=oh, seed=5)
    c2, s2, p2 = encode_object_batched(data, T, K=K, loss=loss, overhead=oh, seed=5)
    assert (c1, s1) == (c2, s2) and p1 == p2
    ok, out = decode_object_batched(c2, s2, p2, F)
    assert ok and np.array_equal(out, data)
    ok, out = decode_object(c1, s1, p2, F)
    assert ok and np.array_equal(out, data)


def test_decode_retry_after_more_symbols():
    L = api()
    K, T = 60, 32
    data = payload(K * T, seed=8)
    c, s, packets = encode_object(data, T, K=K, loss=0.0, overhead=6, seed=1)
    src = [p for p in packets if (p[0] & 0xffffff) < K]
    rep = [p for p in packets if (p[0] & 0xffffff) >= K]
    rq = L.nanorq_decoder_new(c, s)
    out = np.zeros(K * T, np.uint8)
    io = mem_io(out)
    for tag, pl in src[4:]:
        L.nanorq_decoder_add_symbol(rq, (C.c_uint8 * T).from_buffer_copy(pl), tag, io)
    for tag, pl in rep[:3]:
        L.nanorq_decoder_add_symbol(rq, (C.c_uint8 * T).from_buffer_copy(pl), tag, io)
    assert L.nanorq_num_missing(rq, 0) == 4 and not L.nanorq_repair_block(rq, io, 0)
    for tag, pl in rep[3:]:
        L.nanorq_decoder_add_symbol(rq, (C.c_uint8 * T).from_buffer_copy(pl), tag, io)
    assert L.nanorq_repair_block(rq, io, 0) and L.nanorq_num_missing(rq, 0) == 0
    assert np.array_equal(out, data)
    L.nanorq_free(rq)
    io.contents.destroy(io)


def test_reference_benchmark_harness_on_the_hip_path():
    """oracle/_refprog/benchmark_hip = the reference's benchmark.c compiled (in the build container, sources read in
    place) against this library: encode / precalc-encode / decode / decode-with-overhead runs, 6 % loss, ending in
    the harness's own assert(in[i] == out[i]) (reference benchmark.c:233-235)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_refprog", "benchmark_hip")
    if not os.path.isfile(exe):
        pytest.skip("oracle/_refprog/benchmark_hip not built (needs the reference tree at build time)")
    for argv in (["1024", "100", "0"], ["1280", "1000", "5.0"]):
        r = subprocess.run([exe] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (argv, r.stderr.decode()[-500:])
        cols = r.stdout.decode().split()
        assert len(cols) == 5 and int(cols[0]) == int(argv[1]) and all(float(x) > 0 for x in cols[1:])


def test_short_last_block_is_coded_with_block_zeros_table_row(orc):
    """A two-block object whose blocks hold 102 and 101 symbols: the reference codes BOTH with block 0's K' = 114
    (nanorq.c:289, :372) although K=101 has a table row of its own (101).  The packets of block 1 must equal the
    oracle's with that K' passed explicitly -- and differ from what the block's own row would give."""
    L = api()
    T = 48
    F = 203 * T
    data = payload(F, seed=17)
    rq = L.nanorq_encoder_new_ex(F, T, 102, 0, 8)
    io = mem_io(data)
    assert L.nanorq_blocks(rq) == 2 and L.nanorq_block_symbols(rq, 0) == 102 and L.nanorq_block_symbols(rq, 1) == 101
    buf = (C.c_uint8 * T)()
    esis = [101, 102, 103, 150, 5000]
    got = []
    for esi in esis:
        assert L.nanorq_encode(rq, buf, esi, 1, io) == T
        got.append(bytes(buf))
    oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
    L.nanorq_free(rq)
    io.contents.destroy(io)
    src1 = data[102 * T:].reshape(101, T)
    want, _, _ = orc.encode_block(src1, 101, T, esis, Kp=114)
    own, _, _ = orc.encode_block(src1, 101, T, esis)
    assert [w.tobytes() for w in want] == got
    assert [w.tobytes() for w in own] != got
    # and the decoder side: block 1 loses 7 source symbols, receives those repair symbols of the oracle
    lost = [0, 9, 33, 34, 77, 99, 100]
    resi = list(range(101, 101 + len(lost)))
    reps, _, _ = orc.encode_block(src1, 101, T, resi, Kp=114)
    rq = L.nanorq_decoder_new(*oti)
    out = np.zeros(F, np.uint8)
    io = mem_io(out)
    for esi in range(102):
        L.nanorq_decoder_add_symbol(rq, (C.c_uint8 * T).from_buffer_copy(data[esi * T:(esi + 1) * T].tobytes()), L.nanorq_tag(0, esi), io)
    for esi in range(101):
        if esi not in lost:
            L.nanorq_decoder_add_symbol(rq, (C.c_uint8 * T).from_buffer_copy(src1[esi].tobytes()), L.nanorq_tag(1, esi), io)
    for k, esi in enumerate(resi):
        L.nanorq_decoder_add_symbol(rq, (C.c_uint8 * T).from_buffer_copy(reps[k].tobytes()), L.nanorq_tag(1, esi), io)
    assert L.nanorq_repair_block(rq, io, 0) and L.nanorq_repair_block(rq, io, 1)
    assert np.array_equal(out, data)
    L.nanorq_free(rq)
    io.contents.destroy(io)


def test_two_ranks_through_the_real_library(tmp_path):
    """The N>1 path of bench.py through libnanorq_hip.so, started the way the driver starts it: `python bench.py --gpus 2`
    with a clean environment launches its two ranks itself (gloo, both on GPU 0), which shard 16 source blocks -- block b on
    rank b mod 2, no data-path collective -- and must produce, block for block, the repair symbols a single rank produces
    for the same 16 blocks (per-block SHA-256); each rank's decode is verified inside bench.py."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--K", "1024", "--T", "256", "--steps", "1", "--warmup", "1", "--cpu-sample", "0",
            "--pmc", "off", "--no-e2e", "--force-device", "0", "--dist-backend", "gloo"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(base + ["--gpus", "2", "--blocks", "8", "--digest-out", str(tmp_path / "two.json")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["check"]["oracle"]
    assert line["config"]["workload"].startswith("custom")
    r = subprocess.run(base + ["--gpus", "1", "--blocks", "16", "--digest-out", str(tmp_path / "solo.json")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()
    two = dict(json.load(open(str(tmp_path / "two.json") + ".rank0")), **json.load(open(str(tmp_path / "two.json") + ".rank1")))
    solo = json.load(open(tmp_path / "solo.json"))
    assert sorted(two, key=int) == [str(b) for b in range(16)] and two == solo
    # under an external launcher (torch.distributed.run exports these) the process is one rank and starts nothing
    e = dict(env, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671")
    procs = [subprocess.Popen(base + ["--gpus", "2", "--blocks", "8"], env=dict(e, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for k in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][-1])["n_gpus"] == 2


# ------------------------------------------------------------------ streaming path (page-locked memory) ----
def _packets(F, T, K, loss, oh, seed):
    """object bytes, OTI and the packets of a lossy transmission (through the plain per-block calls)"""
    data = payload(F, seed=seed)
    c, s, packets = encode_object(data, T, K=K, loss=loss, overhead=oh, seed=seed)
    return data, c, s, packets


@pytest.mark.parametrize("F,T,K,loss,oh", [(103 * 64 - 17, 64, 25, 0.1, 3), (700 * 1280, 1280, 100, 0.08, 2), (3000 * 256 + 5, 256, 0, 0.05, 4),
                                           (40 * 8192, 8192, 10, 0.2, 1)])
def test_page_locked_contexts_give_the_same_bytes(F, T, K, loss, oh):
    """The DMA paths of the batched calls -- object read straight out of a page-locked context, packets uploaded in one
    piece and sorted into rows on the GPU, decoded blocks written whole into a page-locked context -- against the plain
    calls: identical packets, identical recovered object, identical per-symbol result codes."""
    from capi import SYM_ADDED, pinned_array, pinned_io
    L = api()
    data, c, s, packets = _packets(F, T, K, loss, oh, seed=21)
    # encoder side: batched + page-locked source
    io, mem = pinned_io(F)
    mem[:] = data
    rq = L.nanorq_encoder_new_ex(F, T, K, 0, 8)
    nblk = L.nanorq_blocks(rq)
    assert L.nanorq_generate_symbols_all(rq, io) == nblk
    want = dict(packets)
    Tsz = L.nanorq_symbol_size(rq)
    for sbn in range(nblk):
        nk = L.nanorq_block_symbols(rq, sbn)
        tags = [t for t in want if (t >> 24) == sbn]
        top = max(t & 0xffffff for t in tags) + 1
        buf = np.zeros((top, Tsz), np.uint8)
        assert L.nanorq_encode_range(rq, buf.ctypes.data_as(C.c_void_p), 0, top, sbn, io) == top * Tsz
        for t in tags:
            assert buf[t & 0xffffff].tobytes() == want[t], (sbn, t & 0xffffff, nk)
    L.nanorq_free(rq)
    io.contents.destroy(io)
    # decoder side: page-locked packet buffer (with a duplicate and an out-of-range tag) + page-locked sink
    tags = np.array([t for t, _ in packets] + [packets[0][0], L.nanorq_tag(0, (1 << 24) - 1)], np.uint32)
    addr, blob = pinned_array(len(tags) * Tsz)
    blob[:len(packets) * Tsz] = np.frombuffer(b"".join(p for _, p in packets), np.uint8)
    blob[len(packets) * Tsz:] = 0x11
    dq = L.nanorq_decoder_new(c, s)
    oio, out = pinned_io(F)
    out[:] = 0
    res = np.zeros(len(tags), np.int32)
    added = L.nanorq_decoder_add_symbols(dq, C.c_void_p(addr), tags.ctypes.data_as(C.POINTER(C.c_uint32)), len(tags),
                                         res.ctypes.data_as(C.POINTER(C.c_int)), oio)
    # the same stream of symbols through the per-symbol call gives the reference result codes
    dq2 = L.nanorq_decoder_new(c, s)
    out2 = np.zeros(F, np.uint8)
    oio2 = mem_io(out2)
    res2 = [L.nanorq_decoder_add_symbol(dq2, (C.c_uint8 * Tsz).from_buffer_copy(blob[k * Tsz:(k + 1) * Tsz].tobytes()), int(tags[k]), oio2)
            for k in range(len(tags))]
    assert list(res) == res2 and added == res2.count(SYM_ADDED)
    nb = L.nanorq_blocks(dq)
    assert [L.nanorq_num_missing(dq, b) for b in range(nb)] == [L.nanorq_num_missing(dq2, b) for b in range(nb)]
    assert [L.nanorq_num_repair(dq, b) for b in range(nb)] == [L.nanorq_num_repair(dq2, b) for b in range(nb)]
    done = L.nanorq_repair_all(dq, oio)
    ok2 = all(L.nanorq_repair_block(dq2, oio2, b) for b in range(nb))
    assert (done == nb) == ok2
    if ok2:
        assert np.array_equal(out, data) and np.array_equal(out2, data)
    L.nanorq_free(dq); L.nanorq_free(dq2)
    oio.contents.destroy(oio); oio2.contents.destroy(oio2)
    L.nanorq_pinned_free(addr)


def test_device_resident_blocks_mix_with_per_symbol_calls():
    """A block fed through the page-locked path first and the per-symbol call afterwards (and the other way round); a
    block that stays undecodable keeps what it received, writes it on flush, and is repaired after more symbols."""
    from capi import pinned_array, pinned_io
    L = api()
    K, T = 200, 96
    F = 2 * K * T
    data = payload(F, seed=33)
    c, s, packets = encode_object(data, T, K=K, loss=0.0, overhead=20, seed=3)
    blk = {0: [p for p in packets if (p[0] >> 24) == 0], 1: [p for p in packets if (p[0] >> 24) == 1]}
    src0 = [p for p in blk[0] if (p[0] & 0xffffff) < K]
    rep0 = [p for p in blk[0] if (p[0] & 0xffffff) >= K]
    src1 = [p for p in blk[1] if (p[0] & 0xffffff) < K]
    rep1 = [p for p in blk[1] if (p[0] & 0xffffff) >= K]
    dq = L.nanorq_decoder_new(c, s)
    oio, out = pinned_io(F)
    out[:] = 0

    def add_pinned(pk):
        addr, blob = pinned_array(len(pk) * T)
        blob[:] = np.frombuffer(b"".join(p for _, p in pk), np.uint8)
        tags = np.array([t for t, _ in pk], np.uint32)
        n = L.nanorq_decoder_add_symbols(dq, C.c_void_p(addr), tags.ctypes.data_as(C.POINTER(C.c_uint32)), len(pk), None, oio)
        L.nanorq_pinned_free(addr)
        return n

    def add_single(pk):
        return [L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(p), t, oio) for t, p in pk]

    # block 0: page-locked path first (30 source symbols missing), then single symbols: too few repair symbols at first
    assert add_pinned(src0[30:] + rep0[:10]) == K - 30 + 10
    assert add_single(rep0[10:15]) == [0] * 5
    # (what was received is in the page-locked output since it arrived -- the host writes a source symbol to its place at ingestion, as
    # the reference does, nanorq.c:478-509 -- so there is nothing left for a flush; with "host_rows" off the rows come down here)
    assert np.array_equal(out[30 * T:K * T], data[30 * T:K * T]) and not out[:30 * T].any()
    assert L.nanorq_num_missing(dq, 0) == 30 and not L.nanorq_repair_block(dq, oio, 0)
    assert L.nanorq_decoder_flush(dq, oio) == 0
    assert np.array_equal(out[30 * T:K * T], data[30 * T:K * T]) and not out[:30 * T].any()
    # block 1: single symbols first (host-resident), then the page-locked call falls back to the host path for it
    assert add_single(src1[:50]) == [0] * 50
    assert add_pinned(src1[60:] + rep1[:12]) == K - 60 + 12
    assert add_single(rep0[15:]) == [0] * 5 and add_pinned(rep0[:16] + [rep0[0]] * 16) == 0   # block 0: 5 more, then duplicates only
    assert L.nanorq_repair_all(dq, oio) == 1 and L.nanorq_num_missing(dq, 0) == 30            # 20 repair symbols for 30 gaps
    assert np.array_equal(out[K * T:], data[K * T:])
    assert add_pinned(src0[:16] + src0[16:20]) == 20                                         # 10 gaps left, 20 repair symbols held
    assert L.nanorq_repair_all(dq, oio) == 2
    assert np.array_equal(out, data)
    L.nanorq_free(dq)
    oio.contents.destroy(oio)


def test_objects_on_two_threads():
    """Distinct nanorq objects are independent (reference: no globals); here they share one GPU context, guarded by a
    lock: two threads, each encoding and decoding its own objects, must not disturb each other."""
    import threading
    errs = []

    def work(seed):
        try:
            for rnd in range(3):
                K, T = 150 + 37 * seed, 64 + 32 * seed
                data = payload(3 * K * T - 5, seed=seed * 10 + rnd)
                c, s, pk = encode_object_batched(data, T, K=K, loss=0.1, overhead=3, seed=seed + rnd)
                ok, out = decode_object_batched(c, s, pk, len(data))
                assert ok and np.array_equal(out, data)
                c2, s2, pk2 = encode_object(data, T, K=K, loss=0.1, overhead=3, seed=seed + rnd)
                assert (c2, s2) == (c, s) and pk2 == pk
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in (1, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_surplus_repair_symbols_are_held_in_reserve():
    """A receiver that collected far more repair symbols than gaps: only gaps + 2 become constraint rows up front, the
    rest is taken one at a time when the system is rank deficient -- so the plan stays small and M never leaves the
    planners' 16-bit slot range however many symbols arrived (here: 3 gaps, 4000 repair symbols, K = 2000)."""
    L = api()
    K, T = 2000, 32
    data = payload(K * T, seed=44)
    rq = L.nanorq_encoder_new_ex(K * T, T, K, 0, 8)
    io = mem_io(data)
    buf = np.zeros((4000, T), np.uint8)
    assert L.nanorq_encode_range(rq, buf.ctypes.data_as(C.c_void_p), K, 4000, 0, io) == 4000 * T
    oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
    L.nanorq_free(rq)
    io.contents.destroy(io)
    dq = L.nanorq_decoder_new(*oti)
    assert L.nanorq_set_max_esi(dq, K + 4500)
    out = np.zeros(K * T, np.uint8)
    oio = mem_io(out)
    for e in range(K):
        if e not in (5, 700, 1999):
            L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(data[e * T:(e + 1) * T].tobytes()), L.nanorq_tag(0, e), oio)
    for j in range(4000):
        L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * T).from_buffer_copy(buf[j].tobytes()), L.nanorq_tag(0, K + j), oio)
    assert L.nanorq_num_repair(dq, 0) == 4000 and L.nanorq_repair_block(dq, oio, 0)
    assert np.array_equal(out, data)
    L.nanorq_free(dq)
    oio.contents.destroy(oio)


# ------------------------------------------------------------------ RFC 6330 options (include/nanorq_ext.h) ----
def _ext_roundtrip(F, T, K, N, flags, seed, loss=0.1, oh=3):
    from capi import SYM_ERR
    L = api()
    data = payload(F, seed=seed)
    rq = L.nanorq_encoder_new_ext(F, T, K, 0, N, 8, flags)
    assert rq
    io = mem_io(data)
    rng = np.random.default_rng(seed)
    Tsz = L.nanorq_symbol_size(rq)
    pk = []
    buf = (C.c_uint8 * Tsz)()
    for sbn in range(L.nanorq_blocks(rq)):
        nk = L.nanorq_block_symbols(rq, sbn)
        drop = [e for e in range(nk) if rng.random() < loss]
        for e in [e for e in range(nk) if e not in drop] + list(range(nk, nk + len(drop) + oh)):
            assert L.nanorq_encode(rq, buf, e, sbn, io) == Tsz
            pk.append((L.nanorq_tag(sbn, e), bytes(buf)))
    oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
    info = (L.nanorq_sub_blocks(rq), [L.nanorq_block_symbols(rq, b) for b in range(L.nanorq_blocks(rq))],
            [L.nanorq_block_kprime(rq, b) for b in range(L.nanorq_blocks(rq))])
    L.nanorq_free(rq)
    io.contents.destroy(io)
    dq = L.nanorq_decoder_new_ext(oti[0], oti[1], flags)
    assert dq and L.nanorq_ext_flags(dq) == flags and L.nanorq_sub_blocks(dq) == info[0]
    out = np.zeros(F, np.uint8)
    oio = mem_io(out)
    for t, p in pk:
        assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * len(p)).from_buffer_copy(p), t, oio) != SYM_ERR
    ok = all(L.nanorq_repair_block(dq, oio, b) for b in range(L.nanorq_blocks(dq)))
    L.nanorq_free(dq)
    oio.contents.destroy(oio)
    return ok, np.array_equal(out, data), oti, info, pk, data


def test_rfc_oti_packing():
    """RFC 6330 section 3.3.2 / 3.3.3: common = F | reserved | T, scheme-specific = Z | N | Al, values as they are (the
    nanorq packing stores T-1, Z-1, N-1: reference lib/nanorq.c:309-324)."""
    from capi import EXT_RFC_OTI
    ok, same, oti, info, _, _ = _ext_roundtrip(5 * 100 * 64 - 9, 64, 100, 1, EXT_RFC_OTI, seed=5)
    assert ok and same
    F = 5 * 100 * 64 - 9
    assert oti[0] == (F << 24) | 64 and oti[1] == (5 << 24) | (1 << 8) | 8
    L = api()
    rq = L.nanorq_encoder_new_ex(F, 64, 100, 0, 8)
    assert L.nanorq_oti_common(rq) == (F << 24) | 63 and L.nanorq_oti_scheme_specific(rq) == (4 << 24) | 8   # nanorq's own packing
    L.nanorq_free(rq)
    assert not L.nanorq_decoder_new_ext((F << 24) | 64, (0 << 24) | (1 << 8) | 8, EXT_RFC_OTI)   # Z = 0 is not an object


def test_per_block_kprime(orc):
    """RFC 6330 section 5.3.1.2: a short block is coded with the table row of its own K.  203 symbols in blocks of 102
    and 101: rows 114 and 101 (nanorq: 114 for both, test_short_last_block_is_coded_with_block_zeros_table_row)."""
    from capi import EXT_PER_BLOCK_KP
    T = 48
    ok, same, oti, info, pk, data = _ext_roundtrip(203 * T, T, 102, 1, EXT_PER_BLOCK_KP, seed=6)
    assert ok and same and info[1] == [102, 101] and info[2] == [114, 101]
    src1 = data[102 * T:].reshape(101, T)
    rep1 = [(t & 0xffffff, p) for t, p in pk if (t >> 24) == 1 and (t & 0xffffff) >= 101]
    want, _, _ = orc.encode_block(src1, 101, T, [e for e, _ in rep1])          # the oracle with the block's OWN row
    assert [w.tobytes() for w in want] == [p for _, p in rep1]


@pytest.mark.parametrize("N,T,F", [(2, 64, 3 * 50 * 64), (4, 96, 2 * 40 * 96 - 13), (8, 64, 50 * 64 - 1)])
def test_sub_blocking(orc, N, T, F):
    """RFC 6330 section 4.4.1.2 with N > 1: every symbol is N sub-symbols that lie in N different regions of the block
    (the transfer_symbol branch nanorq never reaches: lib/nanorq.c:78 forces N = 1).  Round trip, and the first repair
    symbol against the oracle on symbols gathered the RFC's way."""
    from capi import EXT_SUBBLOCKS
    K = 50 if N != 4 else 40
    ok, same, oti, info, pk, data = _ext_roundtrip(F, T, K, N, EXT_SUBBLOCKS, seed=7 + N)
    assert ok and same and info[0] == N
    # block 0 gathered by hand: Al = 8, T/Al units split into N sub-symbols (TL/TS units, NL/NS of each)
    unit = T // 8
    TL, TS = -(-unit // N), unit // N
    NL = unit - TS * N
    k0 = info[1][0]
    padded = np.zeros(sum(info[1]) * T + T, np.uint8)
    padded[:F] = data
    sym = np.zeros((k0, T), np.uint8)
    off, col = 0, 0
    for j in range(N):
        w = (TL if j < NL else TS) * 8
        for e in range(k0):
            sym[e, col:col + w] = padded[off + e * w: off + (e + 1) * w]
        off += k0 * w
        col += w
    rep = [(t & 0xffffff, p) for t, p in pk if (t >> 24) == 0 and (t & 0xffffff) >= k0]
    want, _, _ = orc.encode_block(sym, k0, T, [rep[0][0]])
    assert want[0].tobytes() == rep[0][1]
    src = [(t & 0xffffff, p) for t, p in pk if (t >> 24) == 0 and (t & 0xffffff) < k0]
    assert src[0][1] == sym[src[0][0]].tobytes()


def _devices_run(env_devices, K, T, Z, loss):
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("NANORQ_HIP_DEVICES", None)
    if env_devices:
        env["NANORQ_HIP_DEVICES"] = env_devices
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devices_worker.py"),
                        str(K), str(T), str(Z), str(loss)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().split("\n")[-1])


def test_object_spread_over_two_contexts_gives_the_same_packets():
    """NANORQ_HIP_DEVICES names the GPUs; block sbn lives on device sbn mod N and the batched calls run one host
    thread per device.  Two contexts on GPU 0 (the box has one GPU) against the single-context run: same packets, same
    recovered object (reference: blocks share nothing, lib/nanorq.c:97-112)."""
    one = _devices_run("", 600, 320, 7, 0.1)
    two = _devices_run("0,0", 600, 320, 7, 0.1)
    three = _devices_run("0,0,0", 600, 320, 7, 0.1)
    assert one["devices"] == 1 and two["devices"] == 2 and three["devices"] == 3
    assert one["ok"] and two["ok"] and three["ok"]
    assert one["packets"] == two["packets"] == three["packets"]
    assert one["object"] == two["object"] == three["object"]


def test_one_batch_with_many_repair_symbols_per_block():
    """A single nanorq_decoder_add_symbols batch that carries more repair symbols for a block than its device rows hold at
    first (max(K/8, 64)): the rows are grown once, after the batch's bookkeeping, and every symbol of the batch lands in the
    grown buffer (was: the earlier ones were scattered into the outgrown buffer and the block decoded from garbage)."""
    r = _devices_run("", 1000, 256, 3, 0.2)
    assert r["max_repair_per_block"] > 1000 // 8
    assert r["ok"], r


@pytest.mark.parametrize("K,T", [(300, 64), (2000, 1280)])
def test_deferred_generate_symbols_in_a_reset_loop(orc, K, T):
    """The loop of reference benchmark.c:101-109 -- nanorq_generate_symbols + nanorq_encoder_reset over and over, each call
    only enqueueing its solve (and, for a memory context of >= 1 MiB, reading the caller's memory by DMA after page-locking
    it in place) -- with DIFFERENT bytes in the caller's buffer every time: the bytes are consumed when the call returns
    (the buffer is scribbled over right after), and the repair symbols fetched at the end are the oracle's for the LAST
    contents."""
    L = api()
    data = np.zeros(K * T, np.uint8)
    rq = L.nanorq_encoder_new_ex(K * T, T, K, 0, 8)
    io = mem_io(data)
    last = None
    for it in range(5):
        last = payload(K * T, seed=77 + it)
        data[:] = last
        assert L.nanorq_generate_symbols(rq, 0, io)
        data[:] = 0xEE                      # the call has returned: the library may not read these bytes any more
        if it < 4:
            L.nanorq_encoder_reset(rq, 0)
    esis = np.arange(K, K + 40, dtype=np.uint32)
    want, _, _ = orc.encode_block(last.reshape(K, T), K, T, esis)
    buf = (C.c_uint8 * T)()
    for q, esi in enumerate(esis):
        assert L.nanorq_encode(rq, buf, int(esi), 0, io) == T
        assert bytes(buf) == want[q].tobytes(), (it, esi)
    # a source symbol after the solve: the reference regenerates it from the intermediate symbols; here the row is
    # read from the caller's context on first use -- which now holds the scribble, so put the block back first
    data[:] = last
    assert L.nanorq_encode(rq, buf, 3, 0, io) == T and bytes(buf) == last[3 * T:4 * T].tobytes()
    L.nanorq_free(rq)
    io.contents.destroy(io)


def test_page_locking_on_first_use_gives_the_same_packets():
    """NANORQ_HIP_AUTOPIN=0 NANORQ_HIP_LAZY=1 (no page lock on the caller's memory, nothing prepared by the constructors):
    same packets, same recovered object -- compared across two processes of the per-block API"""
    import subprocess
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, 'tests'); "
            "from capi import encode_object, decode_object; from util import payload; "
            "d = payload(1200 * 1280, seed=5); c, s, p = encode_object(d, 1280, K=1200, loss=0.06, overhead=3, seed=2); "
            "ok, out = decode_object(c, s, p, len(d)); assert ok and np.array_equal(out, d); "
            "print(hashlib.sha256(b''.join(x for _, x in p)).hexdigest(), c, s)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"NANORQ_HIP_AUTOPIN": "0", "NANORQ_HIP_LAZY": "1"}):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **env), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        outs.append(r.stdout.decode().split())
    assert outs[0] == outs[1]


def test_object_spread_over_distinct_gpus_gives_the_same_packets():
    """The same on PHYSICALLY different devices (skipped on a one-GPU box): NANORQ_HIP_DEVICES=0,1 against the one-device run --
    same packets, same recovered object; and tools/bench_one_object.py (what bench.py --one-object runs) decodes its object."""
    import json
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    one = _devices_run("", 600, 320, 7, 0.1)
    two = _devices_run("0,1", 600, 320, 7, 0.1)
    assert one["devices"] == 1 and two["devices"] == 2 and one["ok"] and two["ok"]
    assert one["packets"] == two["packets"] and one["object"] == two["object"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_one_object.py"), "--devices", "0,1", "--K", "2000", "--T", "256",
                        "--blocks", "6", "--loss", "0.1", "--reps", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    rec = json.loads(r.stdout.strip().split("\n")[-1])
    assert rec["ok"] and rec["devices"] == 2 and rec["blocks"] == 6


def test_bench_one_object_tool_on_one_gpu():
    """tools/bench_one_object.py with two contexts on GPU 0: the placement bench.py reports as `one_object`"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_one_object.py"), "--devices", "0,0", "--K", "1500", "--T", "256",
                        "--blocks", "5", "--loss", "0.1", "--reps", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    rec = json.loads(r.stdout.strip().split("\n")[-1])
    assert rec["ok"] and rec["devices"] == 2 and rec["blocks"] == 5 and rec["value"] > 0


@pytest.fixture
def book_threads(request):
    """threads that book a packet batch (nanorq_api.c book_worker: thread t the blocks sbn mod P == t), from the first symbol on"""
    L = api()
    L.nanorq_hip_option.restype = C.c_int
    L.nanorq_hip_option.argtypes = [C.c_size_t, C.c_char_p, C.c_longlong]
    assert L.nanorq_hip_option(0, b"book_threads", request.param) == 0 and L.nanorq_hip_option(0, b"book_min", 1) == 0
    yield request.param
    L.nanorq_hip_option(0, b"book_threads", 0)
    L.nanorq_hip_option(0, b"book_min", 0)


@pytest.mark.parametrize("book_threads", [1, 4], indirect=True)
@pytest.mark.parametrize("shuffle", [False, True])
def test_deferred_ingestion_gives_the_same_object(shuffle, book_threads):
    """nanorq_decoder_add_symbols_async: result codes and counts final on return, the bytes travel afterwards in pieces;
    nanorq_repair_all lets every chunk of blocks wait for its last piece only.  Against the waiting call: same codes, same
    recovered object -- packets in block order and shuffled (then every block's last piece is the last one), in TWO batches,
    a block completed by per-symbol calls in between, one block left undecodable (flushed as received).  With the batch booked
    by one thread and by four (a symbol's code depends on the earlier symbols of its own block only: duplicates, symbols for
    a block that is complete, repair indices must come out the same)."""
    from capi import SYM_ADDED, pinned_array, pinned_io
    L = api()
    K, T, Z = 900, 1280, 40           # 46 MB of packets: the deferred path moves them in pieces of 48 MB / several batches
    F = K * T * Z - 333
    data = payload(F, seed=91)
    c, s, packets = encode_object_batched(data, T, K=K, loss=0.08, overhead=3, seed=4)
    # block 7 loses all its repair symbols: it stays incomplete
    packets = [(t, p) for t, p in packets if not ((t >> 24) == 7 and (t & 0xffffff) >= K)]
    if shuffle:
        rng = np.random.default_rng(5)
        packets = [packets[i] for i in rng.permutation(len(packets))]
    Tsz = T
    half = len(packets) // 2
    outs = []
    for mode in ("sync", "async"):
        dq = L.nanorq_decoder_new(c, s)
        oio, out = pinned_io(F)
        out[:] = 0
        addrs, codes = [], []
        for lo, hi in ((0, half), (half, len(packets))):
            tags = np.array([t for t, _ in packets[lo:hi]], np.uint32)
            addr, blob = pinned_array(len(tags) * Tsz)
            blob[:] = np.frombuffer(b"".join(p for _, p in packets[lo:hi]), np.uint8)
            res = np.zeros(len(tags), np.int32)
            add = L.nanorq_decoder_add_symbols_async if mode == "async" else L.nanorq_decoder_add_symbols
            n = add(dq, C.c_void_p(addr), tags.ctypes.data_as(C.POINTER(C.c_uint32)), len(tags), res.ctypes.data_as(C.POINTER(C.c_int)), oio)
            assert n == int((res == SYM_ADDED).sum())
            codes.append(res.copy()); addrs.append(addr)
            if lo == 0:   # between the batches: a per-symbol call on a device-resident block (a duplicate) and a per-block repair
                t0, p0 = packets[0]
                assert L.nanorq_decoder_add_symbol(dq, (C.c_uint8 * Tsz).from_buffer_copy(p0), t0, oio) in (1, 2)
                L.nanorq_repair_block(dq, oio, 3)   # may or may not have enough symbols yet; must not disturb the batch in flight
        done = L.nanorq_repair_all(dq, oio)
        assert L.nanorq_decoder_flush(dq, oio) >= 0
        outs.append((done, [x.tolist() for x in codes], out.copy(), [L.nanorq_num_missing(dq, b) for b in range(Z)]))
        L.nanorq_free(dq)
        oio.contents.destroy(oio)
        for a in addrs:
            L.nanorq_pinned_free(a)
    assert outs[0][0] == outs[1][0] == Z - 1 and outs[0][1] == outs[1][1] and outs[0][3] == outs[1][3]
    assert np.array_equal(outs[0][2], outs[1][2])
    got = outs[1][2].reshape(-1)
    per = K * T
    for b in range(Z):
        if b == 7:
            continue
        assert np.array_equal(got[b * per:min(F, (b + 1) * per)], data[b * per:min(F, (b + 1) * per)]), b
