"""ctypes binding of oracle/liborc.so (built by oracle/Makefile with gcc)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Stats(C.Structure):
    _fields_ = [("i", C.c_uint32), ("u", C.c_uint32), ("recorded_ops", C.c_uint64),
                ("n1", C.c_uint64), ("nB", C.c_uint64), ("n0", C.c_uint64),
                ("gaps", C.c_uint64), ("overhead", C.c_uint64), ("gen_rows", C.c_uint64),
                ("ns_plan", C.c_uint64), ("ns_replay", C.c_uint64), ("ns_gen", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("rq_oracle.c", "plan_exec_ref.c")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        L.orc_params.argtypes = [C.c_uint32, u32p]
        L.orc_tuple.argtypes = [C.c_uint32, C.c_uint32, u32p]
        L.orc_lt_columns.argtypes = [C.c_uint32, C.c_uint32, u32p]
        L.orc_gf_tables.argtypes = [u8p, u8p, u8p]
        L.orc_hdpc.argtypes = [C.c_uint32, u8p]
        L.orc_row_axpy.argtypes = [u8p, u8p, C.c_size_t, C.c_uint8]
        L.orc_row_scal.argtypes = [u8p, C.c_size_t, C.c_uint8]
        L.orc_encode_block.argtypes = [C.c_uint32, C.c_uint32, u8p, u8p, C.c_uint32, u32p, u8p,
                                       C.POINTER(Stats)]
        L.orc_decode_block.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u8p, u8p,
                                       C.POINTER(Stats)]
        L.orc_encode_block_kp.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u8p, u8p, C.c_uint32, u32p, u8p,
                                          C.POINTER(Stats)]
        L.orc_decode_block_kp.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u8p, u8p,
                                          C.POINTER(Stats)]
        L.orc_plan_probe.argtypes = [C.c_uint32, C.c_uint32, u32p, C.POINTER(Stats)]
        L.orc_encode_block_cached.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint32, u32p, u8p]
        _LIB = L
    return _LIB


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


PARAM_NAMES = ("Kp", "J", "S", "H", "W", "L", "P", "P1", "U", "B")


def params(K):
    out = np.zeros(10, dtype=np.uint32)
    if not lib().orc_params(K, _u32(out)):
        raise ValueError("K out of range: %r" % (K,))
    return dict(zip(PARAM_NAMES, (int(x) for x in out)))


def tuple_of(K, isi):
    out = np.zeros(6, dtype=np.uint32)
    lib().orc_tuple(K, isi, _u32(out))
    return tuple(int(x) for x in out)


def lt_columns(K, isi):
    out = np.zeros(64, dtype=np.uint32)
    n = lib().orc_lt_columns(K, isi, _u32(out))
    return [int(x) for x in out[:n]]


def gf_tables():
    e = np.zeros(510, np.uint8); l = np.zeros(256, np.uint8); i = np.zeros(256, np.uint8)
    lib().orc_gf_tables(_u8(e), _u8(l), _u8(i))
    return e, l, i


def hdpc(K):
    p = params(K)
    out = np.zeros((p["H"], p["Kp"] + p["S"]), np.uint8)
    lib().orc_hdpc(K, _u8(out))
    return out


def set_simd(on):
    lib().orc_set_simd(int(on))


def has_avx2():
    return bool(lib().orc_has_avx2())


def has_gfni():
    """AVX-512 + GFNI on this host: set_simd(2) then runs the row kernels as vgf2p8affineqb (one instruction per 64 bytes)."""
    return bool(lib().orc_has_gfni())


def row_axpy(dst, src, beta):
    lib().orc_row_axpy(_u8(dst), _u8(src), dst.size, beta)


def row_scal(dst, beta):
    lib().orc_row_scal(_u8(dst), dst.size, beta)


def encode_block(src, K, T, repair_esis=(), want_inter=False, Kp=0):
    """src: uint8 array of K*T bytes. Returns (repair[nrep,T], inter[L,T] or None, stats).
    Kp: Table 2 row the block is coded with (0 = the row of K; the reference uses block 0's row for every block)."""
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(K, T)
    esis = np.ascontiguousarray(repair_esis, dtype=np.uint32)
    rep = np.zeros((max(len(esis), 1), T), np.uint8)
    inter = np.zeros((params(Kp or K)["L"], T), np.uint8) if want_inter else None
    st = Stats()
    ok = lib().orc_encode_block_kp(K, Kp, T, _u8(src), _u8(inter) if want_inter else None, len(esis),
                                   _u32(esis) if len(esis) else None, _u8(rep), C.byref(st))
    if not ok:
        raise RuntimeError("oracle encode failed")
    return rep[:len(esis)], inter, st.as_dict()


def encode_block_cached(src, K, T, repair_esis=(), Kp=0):
    """encode_block with the schedule kept from the previous call of the same (K, K') -- the reference's precalculated
    encoder (nanorq_precalculate).  Returns repair[nrep, T]."""
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(K, T)
    esis = np.ascontiguousarray(repair_esis, dtype=np.uint32)
    rep = np.zeros((max(len(esis), 1), T), np.uint8)
    if not lib().orc_encode_block_cached(K, Kp, T, _u8(src), len(esis), _u32(esis) if len(esis) else None, _u8(rep)):
        raise RuntimeError("oracle encode failed")
    return rep[:len(esis)]


def decode_block(esis, syms, K, T, Kp=0):
    """esis[n] / syms[n,T] in arrival order. Returns (ok, out[K,T], stats)."""
    esis = np.ascontiguousarray(esis, dtype=np.uint32)
    syms = np.ascontiguousarray(syms, dtype=np.uint8).reshape(len(esis), T)
    out = np.zeros((K, T), np.uint8)
    st = Stats()
    ok = lib().orc_decode_block_kp(K, Kp, T, len(esis), _u32(esis), _u8(syms), _u8(out), C.byref(st))
    return bool(ok), out, st.as_dict()


def plan_probe(K, isis):
    isis = np.ascontiguousarray(isis, dtype=np.uint32)
    st = Stats()
    r = lib().orc_plan_probe(K, len(isis), _u32(isis), C.byref(st))
    return r, st.as_dict()
