"""-m gpu tier for the drop-in object API (include/nanorq.h): what the reference's encode.c /
decode.c / benchmark.c do, through the same calls, with the solve on the GPU."""
import ctypes as C
import hashlib

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
    """The N>1 path of bench.py through libnanorq_hip.so: two ranks (gloo, both on GPU 0) shard 16 source blocks --
    block b on rank b mod 2, no data-path collective -- and must produce, block for block, the repair symbols a
    single rank produces for the same 16 blocks (per-block SHA-256); each rank's decode is verified inside bench.py."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--K", "1024", "--T", "256", "--steps", "1", "--warmup", "1", "--cpu-sample", "0",
            "--pmc", "off", "--no-e2e", "--force-device", "0", "--dist-backend", "gloo"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen(base + ["--gpus", "2", "--blocks", "8", "--digest-out", str(tmp_path / ("r%d.json" % r))], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["check"]["oracle"]
    e = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run(base + ["--gpus", "1", "--blocks", "16", "--digest-out", str(tmp_path / "solo.json")], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()
    two = dict(json.load(open(tmp_path / "r0.json")), **json.load(open(tmp_path / "r1.json")))
    solo = json.load(open(tmp_path / "solo.json"))
    assert sorted(two, key=int) == [str(b) for b in range(16)] and two == solo
