"""ctypes view of the drop-in C API (include/nanorq.h, include/io.h) for tests: the same calls the
reference's encode.c / decode.c / benchmark.c make."""
import ctypes as C

import numpy as np

import nanorq_amd

SYM_DUP, SYM_IGN, SYM_ADDED, SYM_ERR = 2, 1, 0, -1


class IoCtx(C.Structure):
    pass


IoCtx._fields_ = [("read", C.CFUNCTYPE(C.c_size_t, C.POINTER(IoCtx), C.POINTER(C.c_uint8), C.c_size_t)),
                  ("write", C.CFUNCTYPE(C.c_size_t, C.POINTER(IoCtx), C.POINTER(C.c_uint8), C.c_size_t)),
                  ("seek", C.CFUNCTYPE(C.c_bool, C.POINTER(IoCtx), C.c_size_t)),
                  ("size", C.CFUNCTYPE(C.c_size_t, C.POINTER(IoCtx))),
                  ("tell", C.CFUNCTYPE(C.c_long, C.POINTER(IoCtx))),
                  ("destroy", C.CFUNCTYPE(None, C.POINTER(IoCtx))),
                  ("seekable", C.c_bool), ("writable", C.c_bool)]

_L = None


def api():
    global _L
    if _L is None:
        L = nanorq_amd.lib()
        vp, iop = C.c_void_p, C.POINTER(IoCtx)
        L.nanorq_encoder_new.restype = vp
        L.nanorq_encoder_new.argtypes = [C.c_size_t, C.c_uint16, C.c_uint8]
        L.nanorq_encoder_new_ex.restype = vp
        L.nanorq_encoder_new_ex.argtypes = [C.c_size_t, C.c_uint16, C.c_uint16, C.c_uint16, C.c_uint8]
        L.nanorq_decoder_new.restype = vp
        L.nanorq_decoder_new.argtypes = [C.c_uint64, C.c_uint32]
        L.nanorq_free.argtypes = [vp]
        L.nanorq_free.restype = None
        L.nanorq_oti_common.restype = C.c_uint64
        L.nanorq_oti_common.argtypes = [vp]
        L.nanorq_oti_scheme_specific.restype = C.c_uint32
        L.nanorq_oti_scheme_specific.argtypes = [vp]
        for f in ("nanorq_transfer_length", "nanorq_symbol_size", "nanorq_blocks", "nanorq_max_blocks"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [vp]
        L.nanorq_block_symbols.restype = C.c_size_t
        L.nanorq_block_symbols.argtypes = [vp, C.c_uint8]
        L.nanorq_tag.restype = C.c_uint32
        L.nanorq_tag.argtypes = [C.c_uint8, C.c_uint32]
        L.nanorq_precalculate.restype = C.c_bool
        L.nanorq_precalculate.argtypes = [vp]
        L.nanorq_generate_symbols.restype = C.c_bool
        L.nanorq_generate_symbols.argtypes = [vp, C.c_uint8, iop]
        L.nanorq_encode.restype = C.c_size_t
        L.nanorq_encode.argtypes = [vp, vp, C.c_uint32, C.c_uint8, iop]
        L.nanorq_encoder_cleanup.argtypes = [vp, C.c_uint8]
        L.nanorq_encoder_cleanup.restype = None
        L.nanorq_encoder_reset.argtypes = [vp, C.c_uint8]
        L.nanorq_encoder_reset.restype = None
        L.nanorq_set_max_esi.restype = C.c_bool
        L.nanorq_set_max_esi.argtypes = [vp, C.c_uint32]
        L.nanorq_decoder_add_symbol.restype = C.c_int
        L.nanorq_decoder_add_symbol.argtypes = [vp, vp, C.c_uint32, iop]
        L.nanorq_num_missing.restype = C.c_size_t
        L.nanorq_num_missing.argtypes = [vp, C.c_uint8]
        L.nanorq_num_repair.restype = C.c_size_t
        L.nanorq_num_repair.argtypes = [vp, C.c_uint8]
        L.nanorq_repair_block.restype = C.c_bool
        L.nanorq_repair_block.argtypes = [vp, iop, C.c_uint8]
        # include/nanorq_batch.h
        L.nanorq_generate_symbols_all.restype = C.c_size_t
        L.nanorq_generate_symbols_all.argtypes = [vp, iop]
        L.nanorq_encode_range.restype = C.c_size_t
        L.nanorq_encode_range.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint8, iop]
        L.nanorq_decoder_add_symbols.restype = C.c_size_t
        L.nanorq_decoder_add_symbols.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int), iop]
        L.nanorq_decoder_add_symbols_async.restype = C.c_size_t
        L.nanorq_decoder_add_symbols_async.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int), iop]
        L.nanorq_repair_all.restype = C.c_size_t
        L.nanorq_repair_all.argtypes = [vp, iop]
        # include/nanorq_batch.h: page-locked memory; include/nanorq_ext.h: RFC 6330 options
        L.ioctx_from_pinned_mem.restype = iop
        L.ioctx_from_pinned_mem.argtypes = [C.c_size_t]
        L.ioctx_from_registered_mem.restype = iop
        L.ioctx_from_registered_mem.argtypes = [vp, C.c_size_t]
        L.ioctx_mem_base.restype = vp
        L.ioctx_mem_base.argtypes = [iop]
        L.nanorq_pinned_alloc.restype = vp
        L.nanorq_pinned_alloc.argtypes = [C.c_size_t]
        L.nanorq_pinned_free.restype = None
        L.nanorq_pinned_free.argtypes = [vp]
        L.nanorq_decoder_flush.restype = C.c_size_t
        L.nanorq_decoder_flush.argtypes = [vp, iop]
        L.nanorq_encode_range_all.restype = C.c_size_t
        L.nanorq_encode_range_all.argtypes = [vp, vp, C.c_uint32, C.c_uint32, iop]
        L.nanorq_devices.restype = C.c_size_t
        L.nanorq_devices.argtypes = []
        L.nanorq_trim.restype = None
        L.nanorq_trim.argtypes = []
        L.nanorq_encoder_new_ext.restype = vp
        L.nanorq_encoder_new_ext.argtypes = [C.c_size_t, C.c_uint16, C.c_uint16, C.c_uint16, C.c_uint16, C.c_uint8, C.c_uint32]
        L.nanorq_decoder_new_ext.restype = vp
        L.nanorq_decoder_new_ext.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.nanorq_ext_flags.restype = C.c_uint32
        L.nanorq_ext_flags.argtypes = [vp]
        L.nanorq_sub_blocks.restype = C.c_size_t
        L.nanorq_sub_blocks.argtypes = [vp]
        L.nanorq_block_kprime.restype = C.c_size_t
        L.nanorq_block_kprime.argtypes = [vp, C.c_uint8]
        L.ioctx_from_mem.restype = iop
        L.ioctx_from_mem.argtypes = [vp, C.c_size_t]
        L.ioctx_from_file.restype = iop
        L.ioctx_from_file.argtypes = [C.c_char_p, C.c_int]
        L.ioctx_mmap_file.restype = iop
        L.ioctx_mmap_file.argtypes = [C.c_char_p, C.c_int]
        _L = L
    return _L


def mem_io(arr):
    return api().ioctx_from_mem(arr.ctypes.data_as(C.c_void_p), arr.nbytes)


EXT_RFC_OTI, EXT_PER_BLOCK_KP, EXT_SUBBLOCKS = 1, 2, 4


def pinned_io(nbytes):
    """(ioctx over page-locked memory it owns, numpy view of that memory)"""
    L = api()
    io = L.ioctx_from_pinned_mem(nbytes)
    assert io, "ioctx_from_pinned_mem failed"
    return io, np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(L.ioctx_mem_base(io)))


def pinned_array(nbytes):
    """(address, numpy view) of a page-locked packet buffer; free with api().nanorq_pinned_free(address)"""
    L = api()
    p = L.nanorq_pinned_alloc(max(1, nbytes))
    assert p, "nanorq_pinned_alloc failed"
    return p, np.ctypeslib.as_array((C.c_uint8 * max(1, nbytes)).from_address(p))[:nbytes]


def encode_object(data, T, K=0, Z=0, Al=8, loss=0.06, overhead=0, seed=1, precalc=False):
    """The shape of reference benchmark.c:52-116 / encode.c: returns (oti_common, oti_scheme, packets)
    where packets = [(tag, bytes)], each source ESI dropped with probability `loss` and replaced by a
    repair symbol, plus `overhead` extra repair symbols per block."""
    L = api()
    data = np.ascontiguousarray(data, np.uint8)
    rq = L.nanorq_encoder_new_ex(data.nbytes, T, K, Z, Al)
    assert rq, "encoder_new_ex failed"
    io = mem_io(data)
    if precalc:
        assert L.nanorq_precalculate(rq)
    rng = np.random.default_rng(seed)
    Tsz = L.nanorq_symbol_size(rq)
    packets = []
    buf = (C.c_uint8 * Tsz)()
    for sbn in range(L.nanorq_blocks(rq)):
        assert L.nanorq_generate_symbols(rq, sbn, io)
        nk = L.nanorq_block_symbols(rq, sbn)
        dropped = 0
        for esi in range(nk):
            if rng.random() < loss:
                dropped += 1
                continue
            assert L.nanorq_encode(rq, buf, esi, sbn, io) == Tsz
            packets.append((L.nanorq_tag(sbn, esi), bytes(buf)))
        for esi in range(nk, nk + dropped + overhead):
            assert L.nanorq_encode(rq, buf, esi, sbn, io) == Tsz
            packets.append((L.nanorq_tag(sbn, esi), bytes(buf)))
        L.nanorq_encoder_cleanup(rq, sbn)
    oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
    L.nanorq_free(rq)
    io.contents.destroy(io)
    return oti[0], oti[1], packets


def decode_object(oti_common, oti_scheme, packets, nbytes):
    """reference benchmark.c:118-160 / decode.c: returns (ok, data)."""
    L = api()
    rq = L.nanorq_decoder_new(oti_common, oti_scheme)
    assert rq
    out = np.zeros(nbytes, np.uint8)
    io = mem_io(out)
    for tag, payload in packets:
        b = (C.c_uint8 * len(payload)).from_buffer_copy(payload)
        assert L.nanorq_decoder_add_symbol(rq, b, tag, io) != SYM_ERR
    ok = True
    for sbn in range(L.nanorq_blocks(rq)):
        ok = L.nanorq_repair_block(rq, io, sbn) and ok
    L.nanorq_free(rq)
    io.contents.destroy(io)
    return ok, out


def encode_object_batched(data, T, K=0, Z=0, Al=8, loss=0.06, overhead=0, seed=1):
    """encode_object through include/nanorq_batch.h: all blocks solved by one call, symbols fetched in ranges.
    Same packets (same rng consumption) as encode_object."""
    L = api()
    data = np.ascontiguousarray(data, np.uint8)
    rq = L.nanorq_encoder_new_ex(data.nbytes, T, K, Z, Al)
    assert rq, "encoder_new_ex failed"
    io = mem_io(data)
    nblk = L.nanorq_blocks(rq)
    assert L.nanorq_generate_symbols_all(rq, io) == nblk
    rng = np.random.default_rng(seed)
    Tsz = L.nanorq_symbol_size(rq)
    packets = []
    for sbn in range(nblk):
        nk = L.nanorq_block_symbols(rq, sbn)
        keep = [esi for esi in range(nk) if not rng.random() < loss]
        nrep = nk - len(keep) + overhead
        buf = np.zeros((nk + nrep, Tsz), np.uint8)
        assert L.nanorq_encode_range(rq, buf.ctypes.data_as(C.c_void_p), 0, nk + nrep, sbn, io) == (nk + nrep) * Tsz
        for esi in keep + list(range(nk, nk + nrep)):
            packets.append((L.nanorq_tag(sbn, esi), buf[esi].tobytes()))
        L.nanorq_encoder_cleanup(rq, sbn)
    oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
    L.nanorq_free(rq)
    io.contents.destroy(io)
    return oti[0], oti[1], packets


def decode_object_batched(oti_common, oti_scheme, packets, nbytes):
    """decode_object through include/nanorq_batch.h: one add call, one repair call."""
    L = api()
    rq = L.nanorq_decoder_new(oti_common, oti_scheme)
    assert rq
    out = np.zeros(nbytes, np.uint8)
    io = mem_io(out)
    Tsz = L.nanorq_symbol_size(rq)
    blob = np.frombuffer(b"".join(p for _, p in packets), np.uint8).copy()
    tags = np.array([t for t, _ in packets], np.uint32)
    res = np.zeros(len(packets), np.int32)
    added = L.nanorq_decoder_add_symbols(rq, blob.ctypes.data_as(C.c_void_p), tags.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         len(packets), res.ctypes.data_as(C.POINTER(C.c_int)), io)
    assert added == int((res == SYM_ADDED).sum()) and not (res == SYM_ERR).any() and blob.nbytes == len(packets) * Tsz
    nblk = L.nanorq_blocks(rq)
    complete = L.nanorq_repair_all(rq, io)
    L.nanorq_free(rq)
    io.contents.destroy(io)
    return complete == nblk, out
