"""CPU tier for the drop-in boundary (include/nanorq.h, include/io.h, include/nanorq_hip.h): the
library loads, exports every declared symbol, and the host-side object logic (OTI, partitioning,
parameter normalisation, symbol bookkeeping, ioctx back-ends) behaves like the reference
(SURVEY.md section 8(b)).  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nanorq_amd
from capi import SYM_ADDED, SYM_DUP, SYM_ERR, SYM_IGN, api, mem_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:nanorq|nrq|ioctx)_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("header", ["nanorq.h", "io.h", "nanorq_hip.h", "nanorq_batch.h", "nanorq_ext.h"])
def test_library_exports_every_declared_symbol(header):
    L = nanorq_amd.lib()
    names = _declared(header)
    assert len(names) >= 3
    for n in names:
        assert hasattr(L, n), "missing export %s" % n


def test_reference_api_surface_is_complete():
    # the 22 functions of the reference's include/nanorq.h:19-83 and the 3 constructors of io.h:18-20
    want = """nanorq_encoder_new nanorq_encoder_new_ex nanorq_generate_symbols nanorq_free nanorq_oti_common
    nanorq_oti_scheme_specific nanorq_transfer_length nanorq_symbol_size nanorq_blocks nanorq_block_symbols
    nanorq_tag nanorq_max_blocks nanorq_precalculate nanorq_encode nanorq_encoder_cleanup nanorq_encoder_reset
    nanorq_decoder_new nanorq_set_max_esi nanorq_decoder_add_symbol nanorq_num_missing nanorq_num_repair
    nanorq_repair_block ioctx_from_file ioctx_mmap_file ioctx_from_mem""".split()
    have = set(_declared("nanorq.h") + _declared("io.h"))
    assert set(want) <= have


def test_no_gpu_means_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nanorq_amd.NrqError):
        nanorq_amd.Context(0)
    L = api()
    data = np.arange(80, dtype=np.uint8)
    rq = L.nanorq_encoder_new_ex(80, 8, 10, 0, 8)
    io = mem_io(data)
    assert not L.nanorq_generate_symbols(rq, 0, io)   # the solve needs the GPU
    buf = (C.c_uint8 * 8)()
    assert L.nanorq_encode(rq, buf, 3, 0, io) == 8 and bytes(buf) == data[24:32].tobytes()  # plain copy works
    assert L.nanorq_encode(rq, buf, 10, 0, io) == 0  # repair symbol: needs the GPU
    # the batched / multi-device layer (nanorq_batch.h): no context could be opened -- nothing is solved, nothing crashes
    assert L.nanorq_devices() == 0
    assert L.nanorq_generate_symbols_all(rq, io) == 0
    rep = np.zeros(8 * 3, np.uint8)
    assert L.nanorq_encode_range_all(rq, rep.ctypes.data_as(C.c_void_p), 10, 3, io) == 0
    L.nanorq_trim()
    L.nanorq_free(rq)
    io.contents.destroy(io)


def test_oti_known_answers():
    L = api()
    rq = L.nanorq_encoder_new_ex(80, 8, 10, 0, 8)  # SURVEY 8(b) KAT
    assert L.nanorq_oti_common(rq) == 0x0000000050000007 and L.nanorq_oti_scheme_specific(rq) == 0x00000008
    L.nanorq_free(rq)
    rq = L.nanorq_encoder_new_ex(8192 * 1280, 1280, 8192, 0, 8)
    assert L.nanorq_oti_common(rq) == 0x0000a000000004ff
    c, s = L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq)
    dq = L.nanorq_decoder_new(c, s)
    assert L.nanorq_transfer_length(dq) == 8192 * 1280 and L.nanorq_symbol_size(dq) == 1280
    assert L.nanorq_blocks(dq) == 1 and L.nanorq_block_symbols(dq, 0) == 8192 and L.nanorq_block_symbols(dq, 1) == 0
    L.nanorq_free(rq); L.nanorq_free(dq)
    assert L.nanorq_tag(3, 0x1234567) == (3 << 24) | 0x234567 and L.nanorq_max_blocks(None) == 256


def test_parameter_normalisation_and_partitioning():
    L = api()
    # Al snaps down to {1,2,4,8}; T down to a multiple of Al; T below Al becomes Al
    rq = L.nanorq_encoder_new_ex(1000, 13, 0, 4, 7)
    assert L.nanorq_oti_scheme_specific(rq) & 0xff == 4 and L.nanorq_symbol_size(rq) == 12
    # Kt = ceil(1000/12) = 84 over Z=4 -> 4 blocks of 21
    assert L.nanorq_blocks(rq) == 4 and [L.nanorq_block_symbols(rq, b) for b in range(5)] == [21, 21, 21, 21, 0]
    L.nanorq_free(rq)
    rq = L.nanorq_encoder_new_ex(100, 3, 0, 0, 200)
    assert L.nanorq_symbol_size(rq) == 8
    L.nanorq_free(rq)
    # unequal blocks: Kt = 103 symbols, K = 25 -> Z = 5: 3 blocks of 21 and 2 of 20
    rq = L.nanorq_encoder_new_ex(103 * 16, 16, 25, 0, 8)
    assert L.nanorq_blocks(rq) == 5
    assert [L.nanorq_block_symbols(rq, b) for b in range(5)] == [21, 21, 21, 20, 20]
    L.nanorq_free(rq)
    # default: at least 16 blocks
    rq = L.nanorq_encoder_new(16 * 1000 * 64, 64, 8)
    assert L.nanorq_blocks(rq) == 16 and L.nanorq_block_symbols(rq, 0) == 1000
    L.nanorq_free(rq)
    # rejections: too many blocks, too many symbols per block, oversize object
    assert not L.nanorq_encoder_new_ex(300 * 10 * 8, 8, 10, 0, 8)
    assert not L.nanorq_encoder_new_ex(60000 * 8, 8, 60000, 0, 8)
    assert not L.nanorq_encoder_new_ex(946270874880 + 1, 65535, 0, 0, 1)
    assert not L.nanorq_decoder_new((100 << 24) | (7 - 1), (0 << 24) | 4)  # T=7 not a multiple of Al=4


def test_decoder_symbol_bookkeeping():
    L = api()
    K, T = 20, 8
    src = np.arange(K * T, dtype=np.uint8)
    out = np.zeros(K * T, np.uint8)
    enc = L.nanorq_encoder_new_ex(K * T, T, K, 0, 8)
    dq = L.nanorq_decoder_new(L.nanorq_oti_common(enc), L.nanorq_oti_scheme_specific(enc))
    io = mem_io(out)
    sym = (C.c_uint8 * T)()
    kp = nanorq_amd.params(K)["Kp"]
    assert L.nanorq_num_missing(dq, 0) == K and L.nanorq_num_repair(dq, 0) == 0
    for esi in range(K - 2):
        C.memmove(sym, src[esi * T:].ctypes.data, T)
        assert L.nanorq_decoder_add_symbol(dq, sym, L.nanorq_tag(0, esi), io) == SYM_ADDED
    assert L.nanorq_decoder_add_symbol(dq, sym, L.nanorq_tag(0, 3), io) == SYM_DUP
    assert L.nanorq_decoder_add_symbol(dq, sym, L.nanorq_tag(0, 2 * kp + 1), io) == SYM_ERR  # above max_esi
    assert L.nanorq_decoder_add_symbol(dq, sym, L.nanorq_tag(0, 2 * kp), io) == SYM_ADDED    # max_esi itself is fine
    assert L.nanorq_num_missing(dq, 0) == 2 and L.nanorq_num_repair(dq, 0) == 1
    assert np.array_equal(out[:(K - 2) * T], src[:(K - 2) * T])  # source symbols are written through
    assert not L.nanorq_repair_block(dq, io, 0)                  # fewer repair symbols than gaps
    assert not L.nanorq_set_max_esi(dq, kp - 1) and not L.nanorq_set_max_esi(dq, 1 << 24)
    assert L.nanorq_set_max_esi(dq, 5 * kp)
    for esi in (K - 2, K - 1):
        C.memmove(sym, src[esi * T:].ctypes.data, T)
        assert L.nanorq_decoder_add_symbol(dq, sym, L.nanorq_tag(0, esi), io) == SYM_ADDED
    assert L.nanorq_decoder_add_symbol(dq, sym, L.nanorq_tag(0, K + 5), io) == SYM_IGN  # nothing missing any more
    assert L.nanorq_repair_block(dq, io, 0) and np.array_equal(out, src)
    L.nanorq_encoder_reset(dq, 0)
    assert L.nanorq_num_missing(dq, 0) == K and L.nanorq_num_repair(dq, 0) == 0
    L.nanorq_free(dq); L.nanorq_free(enc)
    io.contents.destroy(io)


def test_tail_symbol_is_truncated_at_transfer_length():
    L = api()
    F, T = 37, 8  # 5 symbols, the last one holds 5 bytes
    src = np.arange(F, dtype=np.uint8) + 1
    enc = L.nanorq_encoder_new_ex(F, T, 5, 0, 8)
    io = mem_io(src)
    buf = (C.c_uint8 * T)()
    assert L.nanorq_encode(enc, buf, 4, 0, io) == T
    assert bytes(buf) == src[32:].tobytes() + b"\0\0\0"  # zero padded beyond F
    out = np.full(F, 0xAA, np.uint8)
    oio = mem_io(out)
    dq = L.nanorq_decoder_new(L.nanorq_oti_common(enc), L.nanorq_oti_scheme_specific(enc))
    assert L.nanorq_decoder_add_symbol(dq, buf, L.nanorq_tag(0, 4), oio) == SYM_ADDED
    assert np.array_equal(out[32:], src[32:]) and (out[:32] == 0xAA).all()
    L.nanorq_free(enc); L.nanorq_free(dq)
    io.contents.destroy(io); oio.contents.destroy(oio)


@pytest.mark.parametrize("ctor", ["ioctx_from_file", "ioctx_mmap_file"])
def test_file_ioctx_backends(tmp_path, ctor):
    L = api()
    fn = str(tmp_path / "obj.bin").encode()
    w = getattr(L, ctor)(fn, 0)
    assert w and w.contents.writable and w.contents.seekable
    blob = (np.arange(200000, dtype=np.uint32) * 2654435761 >> 7).astype(np.uint8)
    assert w.contents.seek(w, 100000)
    assert w.contents.write(w, blob[100000:].ctypes.data_as(C.POINTER(C.c_uint8)), 100000) == 100000
    assert w.contents.seek(w, 0)
    assert w.contents.write(w, blob.ctypes.data_as(C.POINTER(C.c_uint8)), 100000) == 100000
    assert w.contents.tell(w) == 100000
    w.contents.destroy(w)
    assert os.path.getsize(fn) == 200000
    r = getattr(L, ctor)(fn, 1)
    assert r and not r.contents.writable and r.contents.size(r) == 200000
    back = np.zeros(200000, np.uint8)
    assert r.contents.seek(r, 150000)
    assert r.contents.read(r, back[150000:].ctypes.data_as(C.POINTER(C.c_uint8)), 60000) == 50000  # clipped at EOF
    assert r.contents.seek(r, 0) and r.contents.read(r, back.ctypes.data_as(C.POINTER(C.c_uint8)), 150000) == 150000
    assert np.array_equal(back, blob)
    if ctor == "ioctx_mmap_file":
        assert not r.contents.seek(r, 200000)
    r.contents.destroy(r)
    assert not getattr(L, ctor)(str(tmp_path / "missing").encode(), 1)


def test_mem_ioctx_clips():
    L = api()
    buf = np.zeros(10, np.uint8)
    io = mem_io(buf)
    src = np.arange(16, dtype=np.uint8)
    assert io.contents.seek(io, 6) and not io.contents.seek(io, 10)
    assert io.contents.write(io, src.ctypes.data_as(C.POINTER(C.c_uint8)), 16) == 4
    assert io.contents.tell(io) == 10 and io.contents.size(io) == 10
    assert list(buf[6:]) == [0, 1, 2, 3]
    io.contents.destroy(io)


def test_rfc_options_host_logic():
    """include/nanorq_ext.h on the CPU tier (no solve involved): OTI words packed the RFC 6330 way and read back, the
    table row of a short block, and the object <-> symbol mapping with N > 1 sub-blocks -- source symbols come out of
    nanorq_encode (a plain gather before the solve) the way RFC 6330 section 4.4.1.2 lays them out."""
    from capi import EXT_PER_BLOCK_KP, EXT_RFC_OTI, EXT_SUBBLOCKS
    L = api()
    F, T = 203 * 48, 48
    rq = L.nanorq_encoder_new_ext(F, T, 102, 0, 1, 8, EXT_RFC_OTI | EXT_PER_BLOCK_KP)
    assert L.nanorq_oti_common(rq) == (F << 24) | T and L.nanorq_oti_scheme_specific(rq) == (2 << 24) | (1 << 8) | 8
    assert [L.nanorq_block_kprime(rq, b) for b in range(2)] == [114, 101] and L.nanorq_ext_flags(rq) == 3
    dq = L.nanorq_decoder_new_ext(L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq), EXT_RFC_OTI | EXT_PER_BLOCK_KP)
    assert dq and L.nanorq_blocks(dq) == 2 and L.nanorq_symbol_size(dq) == T and L.nanorq_transfer_length(dq) == F
    assert [L.nanorq_block_symbols(dq, b) for b in range(2)] == [102, 101]
    L.nanorq_free(dq)
    L.nanorq_free(rq)
    # the same words are not a valid nanorq-packed OTI of the same object (T - 1, Z - 1, N - 1 there)
    dq = L.nanorq_decoder_new((F << 24) | T, (2 << 24) | (1 << 8) | 8)   # read the nanorq way: T = 49, not a multiple of Al
    assert not dq
    assert not L.nanorq_encoder_new_ext(F, T, 102, 0, 7, 8, EXT_SUBBLOCKS)        # N > T / Al
    assert L.nanorq_sub_blocks(L.nanorq_encoder_new_ext(F, T, 102, 0, 3, 8, 0)) == 1  # N ignored without the flag
    # sub-blocking: T = 64, Al = 8 -> 8 units; N = 3 -> sub-symbols of 3, 3, 2 units; K = 10 symbols, one block
    K, T, N = 10, 64, 3
    data = np.arange(K * T, dtype=np.uint32).astype(np.uint8) ^ (np.arange(K * T) // 251).astype(np.uint8)
    rq = L.nanorq_encoder_new_ext(K * T, T, K, 0, N, 8, EXT_SUBBLOCKS)
    io = mem_io(data)
    buf = (C.c_uint8 * T)()
    widths = [24, 24, 16]
    for esi in (0, 3, 9):
        assert L.nanorq_encode(rq, buf, esi, 0, io) == T
        want, off = b"", 0
        for w in widths:   # sub-block j holds the j-th sub-symbol of every symbol, symbol after symbol
            want += data[off + esi * w: off + (esi + 1) * w].tobytes()
            off += K * w
        assert bytes(buf) == want, esi
    L.nanorq_free(rq)
    io.contents.destroy(io)


def test_batched_add_calls_without_a_gpu_are_the_per_symbol_call():
    """nanorq_decoder_add_symbols and its enqueue-only form (nanorq_decoder_add_symbols_async) on ordinary memory -- and on any
    machine without a GPU -- are nanorq_decoder_add_symbol in a loop: same result codes, source symbols written through,
    a block without missing symbols complete (reference lib/nanorq.c:478-509, :605-606)."""
    L = api()
    K, T = 30, 16
    src = np.arange(K * T, dtype=np.uint8)
    enc = L.nanorq_encoder_new_ex(K * T, T, K, 0, 8)
    oti = (L.nanorq_oti_common(enc), L.nanorq_oti_scheme_specific(enc))
    tags = np.array([L.nanorq_tag(0, e) for e in list(range(K)) + [3, K + 2]], np.uint32)     # all source symbols, a duplicate, a late repair symbol
    blob = np.concatenate([src, src[3 * T:4 * T], np.zeros(T, np.uint8)])
    for name in ("nanorq_decoder_add_symbols", "nanorq_decoder_add_symbols_async"):
        dq = L.nanorq_decoder_new(*oti)
        out = np.zeros(K * T, np.uint8)
        io = mem_io(out)
        res = np.full(len(tags), 99, np.int32)
        n = getattr(L, name)(dq, blob.ctypes.data_as(C.c_void_p), tags.ctypes.data_as(C.POINTER(C.c_uint32)), len(tags),
                             res.ctypes.data_as(C.POINTER(C.c_int)), io)
        assert n == K and list(res[:K]) == [SYM_ADDED] * K and list(res[K:]) == [1, 1], (name, list(res))   # complete block: IGN, IGN
        assert np.array_equal(out, src) and L.nanorq_num_missing(dq, 0) == 0
        assert L.nanorq_repair_block(dq, io, 0)            # nothing missing: true without a solve
        L.nanorq_free(dq)
        io.contents.destroy(io)
    L.nanorq_free(enc)
