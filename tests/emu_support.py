"""Test support: drive the CPU emulation of the solve workgroup (tests/emu/solve_emu.cpp) and the
oracle's reference plan executor with host-built plans."""
import ctypes as C

import numpy as np

import nanorq_amd
from nanorq_amd import build as nbuild


class Job(C.Structure):
    _fields_ = [("plan", C.c_uint64), ("rowsrc", C.c_uint64), ("src", C.c_uint64), ("rep", C.c_uint64),
                ("inter", C.c_uint64), ("out", C.c_uint64), ("out_cptr", C.c_uint64), ("out_slots", C.c_uint64),
                ("out_row", C.c_uint64), ("nout", C.c_uint32), ("pad", C.c_uint32)]


_EMU = None
ROW_ZERO = 0xFFFFFFFF
ROW_REP = 0x80000000


def emu():
    global _EMU
    if _EMU is None:
        L = C.CDLL(nbuild.build_emu())
        L.emu_solve.argtypes = [C.POINTER(Job), C.c_uint32, C.c_uint32, C.c_void_p]
        L.emu_lds_bytes.argtypes = [C.c_void_p, C.c_uint32]
        L.emu_lds_bytes.restype = C.c_uint32
        L.emu_set_dense_shared_min_nt.argtypes = [C.c_uint32]
        L.emu_set_hdpc_regs.argtypes = [C.c_int]
        _EMU = L
    return _EMU


def decode_setup(orc, K, lost, rep_esis):
    """Row/ISI layout of a decode (reference nanorq.c:527-565): returns (isis, rowsrc)."""
    p = orc.params(K)
    pad = p["Kp"] - K
    nl = len(lost)
    overhead = len(rep_esis) - nl
    isis = list(range(p["Kp"]))
    rowsrc = np.full(p["L"] + overhead, ROW_ZERO, np.uint32)
    rowsrc[p["S"] + p["H"]: p["S"] + p["H"] + K] = np.arange(K, dtype=np.uint32)
    for g, e in enumerate(lost):
        isis[int(e)] = int(rep_esis[g]) + pad
        rowsrc[p["S"] + p["H"] + int(e)] = ROW_REP | g
    for x in range(overhead):
        isis.append(int(rep_esis[nl + x]) + pad)
        rowsrc[p["L"] + x] = ROW_REP | (nl + x)
    return np.array(isis, np.uint32), rowsrc


def lt_lists(orc, K, isis, plan):
    """LT neighbour lists translated to slots through the plan's colslot[] (what ph_store reads)."""
    hdr = nanorq_amd.plan_header(plan)
    colslot = np.frombuffer(plan, np.uint16, count=hdr["L"], offset=hdr["off_colslot"])
    cptr = [0]
    cols = []
    for x in isis:
        cols += [int(colslot[c]) for c in orc.lt_columns(K, int(x))]
        cptr.append(len(cols))
    return np.array(cptr, np.uint32), np.array(cols if cols else [0], np.uint16)


def emu_solve(plan, kconst, rowsrc, src, rep, T, L, out_isis_lists, out_rows, out_buf, wb):
    """Runs all strips. Returns (status, inter[L,T]); out_buf is modified in place."""
    L_ = emu()
    planb = (C.c_uint8 * len(plan)).from_buffer_copy(plan)
    kcb = (C.c_uint8 * len(kconst)).from_buffer_copy(kconst)
    inter = np.zeros((L, T), np.uint8)
    cptr, cols = out_isis_lists
    out_rows = np.ascontiguousarray(out_rows, np.uint32)
    rep = np.ascontiguousarray(rep if rep is not None and len(rep) else np.zeros((1, T), np.uint8))
    j = Job()
    j.plan = C.addressof(planb)
    j.rowsrc = rowsrc.ctypes.data
    j.src = src.ctypes.data
    j.rep = rep.ctypes.data
    j.inter = inter.ctypes.data
    j.out = out_buf.ctypes.data
    j.out_cptr = cptr.ctypes.data
    j.out_slots = cols.ctypes.data
    j.out_row = out_rows.ctypes.data if len(out_rows) else 0
    j.nout = len(out_rows)
    r = L_.emu_solve(C.byref(j), T, wb, C.addressof(kcb))
    return r, inter


_PEMU = None


def pemu():
    global _PEMU
    if _PEMU is None:
        L = C.CDLL(nbuild.build_planner_emu())
        u32p = C.POINTER(C.c_uint32)
        L.emu_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, u32p, C.c_uint32, u32p, C.c_uint32, C.c_uint32,
                               C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Job)]
        L.emu_plan_arena_bound.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        L.emu_plan_arena_bound.restype = C.c_uint32
        _PEMU = L
    return _PEMU


def emu_device_plan(K, kconst, lost, rep_esis, lds_bytes=140 * 1024, Kp=0, use=None, encode=False, split=False, caps=None,
                    compact=True, wentry=False):
    """Run the GPU planner's phase code on the CPU. Returns (plan arena bytes, header).  `use` = number of repair
    symbols to use up front (default all); the rest may be taken one at a time if the system is rank deficient."""
    L_ = pemu()
    kcb = (C.c_uint8 * len(kconst)).from_buffer_copy(kconst)
    lost = np.ascontiguousarray(lost, np.uint32)
    rep_esis = np.ascontiguousarray(rep_esis, np.uint32)
    cap = L_.emu_plan_arena_bound(K, C.addressof(kcb), max(0, len(rep_esis) - len(lost)) + 24, len(lost) + 8)
    use = len(rep_esis) if use is None else use
    arena = np.zeros(cap, np.uint8)
    job = Job()
    L_.emu_plan_set_caps(*(caps or (0, 0)))  # capacities of the arrays behind pl_shared (0 = the big-block ones)
    # encode plan: a job without missing symbols; split: the phase sequence in its two parts; compact=False: a peeling state
    # that does not fit the LDS lives in the workspace ALONE (no compact copy of counts and flags in LDS)
    # wentry (with split): part 1 cut once more, the entry pass over the matrix by two "workgroups" in between (nrq_wentry_kernel)
    L_.emu_plan_set_mode((1 if encode else 0) | (0x100 if split else 0) | (0 if compact else 0x200) | (0x400 if wentry else 0))
    try:
        rc = L_.emu_plan(K, Kp, C.addressof(kcb), lost.ctypes.data_as(C.POINTER(C.c_uint32)), len(lost),
                         rep_esis.ctypes.data_as(C.POINTER(C.c_uint32)), use, len(rep_esis), arena.ctypes.data, cap, lds_bytes,
                         C.byref(job))
    finally:
        L_.emu_plan_set_mode(0)
        L_.emu_plan_set_caps(0, 0)
    assert rc == 0
    hdr = nanorq_amd.plan_header(arena.tobytes()[:256])
    return arena.tobytes()[:max(hdr["total_bytes"], 256)], hdr
