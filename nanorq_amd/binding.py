"""ctypes binding of libnanorq_hip.so (include/nanorq_hip.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class NrqError(RuntimeError):
    pass


class CallStats(C.Structure):
    _fields_ = [("plan_ms", C.c_double), ("host_ms", C.c_double), ("strip_bytes", C.c_uint32),
                ("lds_bytes", C.c_uint32), ("grid", C.c_uint32), ("planner", C.c_uint32),
                ("plan_bytes", C.c_uint64), ("xor_ops", C.c_uint64), ("npiv", C.c_uint32), ("u", C.c_uint32),
                ("nlev", C.c_uint32), ("nfree", C.c_uint32), ("wg_threads", C.c_uint32), ("strips_per_slot", C.c_uint32),
                ("wg_waves_per_simd", C.c_uint32), ("host_planned", C.c_uint32), ("movers_aligned", C.c_uint32),
                ("plan_ahead", C.c_uint32), ("strip_bytes_b", C.c_uint32), ("blocks_b", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib_path():
    return os.path.join(_HERE, "libnanorq_hip.so")


def lib():
    """Load the native library (building it first if the sources are newer). No fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    from . import build
    path = os.environ.get("NANORQ_HIP_LIB") or build.build_lib()  # the override loads a tuning variant of the same library
    L = C.CDLL(path)
    vp, u32p, ip = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)
    sz = C.c_size_t
    L.nrq_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.nrq_ctx_destroy.argtypes = [vp]
    L.nrq_ctx_destroy.restype = None
    L.nrq_ctx_set_stream.argtypes = [vp, vp]
    L.nrq_ctx_error.argtypes = [vp]
    L.nrq_ctx_error.restype = C.c_char_p
    L.nrq_ctx_sync.argtypes = [vp]
    L.nrq_ctx_last_stats.argtypes = [vp, C.POINTER(CallStats)]
    L.nrq_ctx_last_stats.restype = None
    L.nrq_ctx_set_threads.argtypes = [vp, C.c_int]
    L.nrq_ctx_set_planner.argtypes = [vp, C.c_int]
    L.nrq_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    L.nrq_params.argtypes = [C.c_uint32, u32p]
    L.nrq_precalculate.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.nrq_plan_cache_clear.argtypes = [vp]
    L.nrq_plan_cache_clear.restype = None
    L.nrq_encode_blocks.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, sz, vp, sz, C.c_uint32, u32p, vp, sz]
    L.nrq_decode_blocks.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, sz, u32p, u32p, C.c_uint32, u32p,
                                    u32p, C.c_uint32, vp, sz, vp, sz, ip]
    L.nrq_decode_blocks_lazy.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, sz, u32p, u32p, C.c_uint32,
                                         u32p, u32p, u32p, C.c_uint32, vp, sz, vp, sz, ip, u32p]
    L.nrq_decode_plan_ahead.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, sz, u32p, u32p, C.c_uint32,
                                        u32p, u32p, u32p, C.c_uint32, vp, sz, vp, sz]
    L.nrq_gen_symbols.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, sz, C.c_uint32, u32p, vp, sz]
    L.nrq_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.nrq_dev_free.argtypes = [vp, vp]
    L.nrq_dev_upload.argtypes = [vp, vp, vp, sz]
    L.nrq_dev_download.argtypes = [vp, vp, vp, sz]
    L.nrq_dev_memset.argtypes = [vp, vp, C.c_int, sz]
    L.nrq_ktime_enable.argtypes = [vp, C.c_int]
    L.nrq_ktime_read.argtypes = [vp, C.POINTER(C.c_float), C.c_uint32, u32p]
    L.nrq_ptime_read.argtypes = [vp, C.POINTER(C.c_float), C.c_uint32, u32p]
    L.nrq_ktime_read_intervals.argtypes = [vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, u32p]
    L.nrq_timer_start.argtypes = [vp]
    L.nrq_timer_stop_ms.argtypes = [vp, C.POINTER(C.c_float)]
    u8pp = C.POINTER(C.POINTER(C.c_uint8))
    L.nrq_host_kconst_build.argtypes = [C.c_uint32, u8pp, u32p]
    L.nrq_host_plan_build.argtypes = [C.c_uint32, C.c_uint32, u32p, C.POINTER(C.c_uint8), u8pp, u32p]
    L.nrq_host_free.argtypes = [vp]
    L.nrq_host_free.restype = None
    _LIB = L
    return L


PARAM_NAMES = ("Kp", "J", "S", "H", "W", "L", "P", "P1", "U", "B")
PLAN_FIELDS = ("magic status K Kp J S H W L P P1 B M npiv u nlow r2 nfree nlev nrows pipe wpr "
               "npiv_pad n_xor_ops off_ops off_pivslot off_pivcol off_wt off_lowslot off_pivx off_fbits "
               "off_mh off_freex off_hinv off_colslot off_pivof off_uslot total_bytes reserved0 reserved1 fail_site "
               "off_augt lpr aug_stride").split()


def params(K):
    out = (C.c_uint32 * 10)()
    if lib().nrq_params(K, out) != 0:
        raise ValueError("K out of range: %r" % (K,))
    return dict(zip(PARAM_NAMES, (int(x) for x in out)))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def host_kconst(K):
    """Per-K' HDPC constants arena (bytes)."""
    out = C.POINTER(C.c_uint8)()
    n = C.c_uint32()
    if lib().nrq_host_kconst_build(K, C.byref(out), C.byref(n)) != 0:
        raise NrqError("kconst build failed")
    buf = bytes(C.string_at(out, n.value))
    lib().nrq_host_free(out)
    return buf


def host_plan(K, isis, kconst):
    """Host planner: returns the plan arena as bytes (header status tells solvable/singular)."""
    isis = np.ascontiguousarray(isis, dtype=np.uint32)
    kc = (C.c_uint8 * len(kconst)).from_buffer_copy(kconst)
    out = C.POINTER(C.c_uint8)()
    n = C.c_uint32()
    rc = lib().nrq_host_plan_build(K, len(isis), _u32(isis), kc, C.byref(out), C.byref(n))
    if rc != 0:
        raise NrqError("plan build failed: %d" % rc)
    buf = bytes(C.string_at(out, n.value))
    lib().nrq_host_free(out)
    return buf


def plan_header(plan):
    h = np.frombuffer(plan, dtype=np.uint32, count=len(PLAN_FIELDS))
    return dict(zip(PLAN_FIELDS, (int(x) for x in h)))


class Context:
    """One per GPU.  Device buffers are plain integers (device addresses): pass torch tensor
    data_ptr()s, or use alloc()/upload()/download()."""

    def __init__(self, device=0, stream=None):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.nrq_ctx_create(device, C.c_void_p(stream or 0), C.byref(h))
        if rc != 0:
            raise NrqError("nrq_ctx_create failed (%d): no usable HIP device -- the HIP path has no CPU fallback" % rc)
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.nrq_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise NrqError("%s (rc=%d)" % (self._L.nrq_ctx_error(self._h).decode(), rc))

    def set_stream(self, stream):
        self._chk(self._L.nrq_ctx_set_stream(self._h, C.c_void_p(stream or 0)))

    def set_planner(self, device=True):
        self._chk(self._L.nrq_ctx_set_planner(self._h, int(bool(device))))

    def set_option(self, name, value):
        self._chk(self._L.nrq_ctx_set_option(self._h, name.encode(), int(value)))

    def set_threads(self, n):
        self._chk(self._L.nrq_ctx_set_threads(self._h, n))

    def sync(self):
        self._chk(self._L.nrq_ctx_sync(self._h))

    def stats(self):
        s = CallStats()
        self._L.nrq_ctx_last_stats(self._h, C.byref(s))
        return s.as_dict()

    def precalculate(self, K, Kp=0):
        self._chk(self._L.nrq_precalculate(self._h, K, Kp))

    def clear_plan_cache(self):
        self._L.nrq_plan_cache_clear(self._h)

    # -- raw device memory -------------------------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self._L.nrq_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        self._chk(self._L.nrq_dev_free(self._h, C.c_void_p(ptr)))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self._L.nrq_dev_upload(self._h, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, dptr, nbytes):
        out = np.empty(nbytes, np.uint8)
        self._chk(self._L.nrq_dev_download(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), nbytes))
        return out

    def memset(self, dptr, value, nbytes):
        self._chk(self._L.nrq_dev_memset(self._h, C.c_void_p(dptr), value, nbytes))

    # -- hot path ----------------------------------------------------------------------------
    def encode_blocks(self, K, T, nblk, d_src, src_stride, d_rep, rep_stride, esis, d_inter=0, inter_stride=0, Kp=0):
        esis = np.ascontiguousarray(esis, dtype=np.uint32)
        self._chk(self._L.nrq_encode_blocks(self._h, K, Kp, T, nblk, C.c_void_p(d_src), src_stride,
                                            C.c_void_p(d_inter or 0), inter_stride, len(esis),
                                            _u32(esis) if len(esis) else None, C.c_void_p(d_rep or 0), rep_stride))

    def decode_blocks(self, K, T, nblk, d_src, src_stride, lost, nlost, rep_esi, nrep, d_rep, rep_stride,
                      d_inter=0, inter_stride=0, Kp=0):
        """lost: [nblk, lost_cap] uint32, rep_esi: [nblk, rep_cap] uint32. Returns status int array."""
        lost = np.ascontiguousarray(lost, dtype=np.uint32).reshape(nblk, -1)
        rep_esi = np.ascontiguousarray(rep_esi, dtype=np.uint32).reshape(nblk, -1)
        nlost = np.ascontiguousarray(nlost, dtype=np.uint32)
        nrep = np.ascontiguousarray(nrep, dtype=np.uint32)
        status = np.zeros(nblk, dtype=np.int32)
        self._chk(self._L.nrq_decode_blocks(self._h, K, Kp, T, nblk, C.c_void_p(d_src), src_stride, _u32(lost),
                                            _u32(nlost), lost.shape[1], _u32(rep_esi), _u32(nrep), rep_esi.shape[1],
                                            C.c_void_p(d_rep or 0), rep_stride, C.c_void_p(d_inter or 0),
                                            inter_stride, status.ctypes.data_as(C.POINTER(C.c_int))))
        return status

    def decode_blocks_lazy(self, K, T, nblk, d_src, src_stride, lost, nlost, rep_esi, nrep, nrep_avail, d_rep, rep_stride,
                           d_inter=0, inter_stride=0, Kp=0):
        """Like decode_blocks, but block b starts from its first nrep[b] repair symbols and takes more (up to
        nrep_avail[b]) only if its system is rank deficient.  Returns (status, used)."""
        lost = np.ascontiguousarray(lost, dtype=np.uint32).reshape(nblk, -1)
        rep_esi = np.ascontiguousarray(rep_esi, dtype=np.uint32).reshape(nblk, -1)
        nlost = np.ascontiguousarray(nlost, dtype=np.uint32)
        nrep = np.ascontiguousarray(nrep, dtype=np.uint32)
        nrep_avail = np.ascontiguousarray(nrep_avail, dtype=np.uint32)
        status = np.zeros(nblk, dtype=np.int32)
        used = np.zeros(nblk, dtype=np.uint32)
        self._chk(self._L.nrq_decode_blocks_lazy(self._h, K, Kp, T, nblk, C.c_void_p(d_src), src_stride, _u32(lost),
                                                 _u32(nlost), lost.shape[1], _u32(rep_esi), _u32(nrep), _u32(nrep_avail),
                                                 rep_esi.shape[1], C.c_void_p(d_rep or 0), rep_stride,
                                                 C.c_void_p(d_inter or 0), inter_stride,
                                                 status.ctypes.data_as(C.POINTER(C.c_int)), _u32(used)))
        return status, used

    def decode_plan_ahead(self, K, T, nblk, d_src, src_stride, lost, nlost, rep_esi, nrep, nrep_avail, d_rep, rep_stride,
                          d_inter=0, inter_stride=0, Kp=0):
        """Issue the planner run of the decode_blocks_lazy (nrep_avail given) / decode_blocks (None) call with the same
        arguments now; that call then only waits for it."""
        lost = np.ascontiguousarray(lost, dtype=np.uint32).reshape(nblk, -1)
        rep_esi = np.ascontiguousarray(rep_esi, dtype=np.uint32).reshape(nblk, -1)
        nlost = np.ascontiguousarray(nlost, dtype=np.uint32)
        nrep = np.ascontiguousarray(nrep, dtype=np.uint32)
        av = None if nrep_avail is None else np.ascontiguousarray(nrep_avail, dtype=np.uint32)
        self._chk(self._L.nrq_decode_plan_ahead(self._h, K, Kp, T, nblk, C.c_void_p(d_src), src_stride, _u32(lost), _u32(nlost),
                                                lost.shape[1], _u32(rep_esi), _u32(nrep), None if av is None else _u32(av),
                                                rep_esi.shape[1], C.c_void_p(d_rep or 0), rep_stride, C.c_void_p(d_inter or 0),
                                                inter_stride))

    def gen_symbols(self, K, T, nblk, d_inter, inter_stride, isis, d_out, out_stride, Kp=0):
        isis = np.ascontiguousarray(isis, dtype=np.uint32)
        self._chk(self._L.nrq_gen_symbols(self._h, K, Kp, T, nblk, C.c_void_p(d_inter), inter_stride, len(isis),
                                          _u32(isis), C.c_void_p(d_out), out_stride))

    def ktime_enable(self, on=True):
        self._chk(self._L.nrq_ktime_enable(self._h, int(on)))

    def ktime_read(self, cap=65536):
        buf = (C.c_float * cap)()
        n = C.c_uint32()
        self._chk(self._L.nrq_ktime_read(self._h, buf, cap, C.byref(n)))
        return [float(buf[k]) for k in range(min(cap, n.value))]

    def ptime_read(self, cap=65536):
        """durations (ms) of the decode planner runs since ktime_enable"""
        buf = (C.c_float * cap)()
        n = C.c_uint32()
        self._chk(self._L.nrq_ptime_read(self._h, buf, cap, C.byref(n)))
        return [float(buf[k]) for k in range(min(cap, n.value))]

    def ktime_read_intervals(self, ref=None, cap=65536):
        """[(start_ms, dur_ms)] of the solve-kernel launches since ktime_enable, on `ref`'s time axis."""
        a = (C.c_float * cap)()
        b = (C.c_float * cap)()
        n = C.c_uint32()
        self._chk(self._L.nrq_ktime_read_intervals(self._h, (ref or self)._h, a, b, cap, C.byref(n)))
        return [(float(a[k]), float(b[k])) for k in range(min(cap, n.value))]

    def timer_start(self):
        self._chk(self._L.nrq_timer_start(self._h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._chk(self._L.nrq_timer_stop_ms(self._h, C.byref(ms)))
        return float(ms.value)


def plan_ops(plan, header=None):
    """The op stream of a plan as an array [row, lane] (a copy).  plan.h: the stream is stored quad-interleaved -- the words
    of rows 4g..4g+3 of a lane lie next to each other (NRQ_OP_INDEX)."""
    import numpy as np
    h = header or plan_header(bytes(plan))
    rows = (h["nrows"] + 3) & ~3
    a = np.frombuffer(bytes(plan), dtype=np.uint32, offset=h["off_ops"], count=rows * 64)
    return a.reshape(rows // 4, 64, 4).transpose(0, 2, 1).reshape(rows, 64)[:h["nrows"]].copy()


def plan_ops_store(plan, ops, header=None):
    """Write an array [row, lane] (all nrows rows) back into the bytearray `plan` (tests that tamper with a stream)."""
    import numpy as np
    h = header or plan_header(bytes(plan))
    rows = (h["nrows"] + 3) & ~3
    assert ops.shape == (h["nrows"], 64)
    full = np.frombuffer(plan, dtype=np.uint32, offset=h["off_ops"], count=rows * 64).reshape(rows // 4, 64, 4)
    for r in range(h["nrows"]):
        full[r // 4, :, r % 4] = ops[r]
