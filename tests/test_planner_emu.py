"""CPU tier: host planner + CPU emulation of the HIP solve workgroup versus the oracle.

These run the SAME per-thread phase code the gfx950 kernel runs (nanorq_amd/csrc/solve_body.h),
sequentially on the CPU (tests/emu/solve_emu.cpp), so plan format, strip indexing and the packed
GF(256) arithmetic are verified without a GPU.  The -m gpu tests then repeat the comparisons on
the real kernel through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import nanorq_amd
from emu_support import ROW_ZERO, decode_setup, emu, emu_device_plan, emu_solve, lt_lists
from util import loss_pattern, payload, received_set


def _encode_case(orc, K, T, wb, nrep=7):
    p = orc.params(K)
    src = payload(K * T, seed=11).reshape(K, T)
    kc = nanorq_amd.host_kconst(K)
    plan = nanorq_amd.host_plan(K, np.arange(p["Kp"], dtype=np.uint32), kc)
    hdr = nanorq_amd.plan_header(plan)
    assert hdr["status"] == 0 and hdr["npiv"] + hdr["u"] == p["L"]
    rowsrc = np.full(p["L"], ROW_ZERO, np.uint32)
    rowsrc[p["S"] + p["H"]: p["S"] + p["H"] + K] = np.arange(K, dtype=np.uint32)
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    lists = lt_lists(orc, K, esis + (p["Kp"] - K), plan)
    out = np.zeros((nrep, T), np.uint8)
    r, inter = emu_solve(plan, kc, rowsrc, src, None, T, p["L"], lists, np.arange(nrep), out, wb)
    ref_rep, ref_inter, _ = orc.encode_block(src, K, T, esis, want_inter=True)
    assert r == 1
    assert np.array_equal(inter, ref_inter)
    assert np.array_equal(out, ref_rep)


@pytest.mark.parametrize("K,T,wb", [(10, 8, 16), (10, 8, 4), (10, 24, 16), (100, 64, 16), (100, 36, 8), (100, 10, 4),
                                    (100, 6, 2), (1024, 32, 16), (1024, 20, 8), (8192, 16, 16),
                                    # T a multiple of 16: narrow strips gathered as 16-byte chunks of their line group (pf_gather_chunks)
                                    (100, 32, 8), (100, 16, 2), (1024, 48, 4), (300, 160, 2),
                                    # 12-byte strips: T a multiple of 12, of 4 only (the last strip holds 4 or 8 bytes), of neither
                                    (10, 24, 12), (100, 36, 12), (100, 64, 12), (100, 104, 12), (1024, 20, 12), (100, 13, 12), (8192, 16, 12)])
def test_encode_emulated_matches_oracle(orc, K, T, wb):
    _encode_case(orc, K, T, wb)


@pytest.mark.parametrize("K", [300, 2000])
def test_encoder_plan_by_components(orc, K, monkeypatch):
    """The encoder's plan takes the rows of an inactivation event from the largest components of the two-column rows' graph (RFC
    6330 section 5.4.2.2; planner_host.cpp nrq_host_plan_build: one row an event from K'=1500 on, six below).  Whatever the
    rule -- NRQ_HOST_WAY forces first-found (0) or n rows an event -- the encode is the oracle's, byte for byte (the reference
    takes any row with two columns left, precode.c:115-126: the choice is the plan's business, not the result's); and the
    rule must not end at more inactive columns than first-found does."""
    p = orc.params(K)
    kc = nanorq_amd.host_kconst(K)
    u = {}
    for way in ("0", "6", "1", None):
        if way is None:
            monkeypatch.delenv("NRQ_HOST_WAY", raising=False)
        else:
            monkeypatch.setenv("NRQ_HOST_WAY", way)
        u[way] = nanorq_amd.plan_header(nanorq_amd.host_plan(K, np.arange(p["Kp"], dtype=np.uint32), kc))["u"]
        _encode_case(orc, K, 16, 16, nrep=3)
    assert u[None] == u["1" if p["Kp"] >= 1500 else "6"]
    assert u[None] <= u["0"], u


@pytest.mark.parametrize("K,T,wb,p,oh", [(10, 16, 16, 0.3, 0), (100, 32, 16, 0.06, 0), (100, 32, 8, 0.06, 2),
                                         (100, 8, 2, 0.5, 30), (1024, 16, 16, 0.05, 0), (1024, 16, 4, 0.06, 52),
                                         (8192, 16, 16, 0.1, 0), (8192, 16, 16, 0.1, 2),
                                         (100, 32, 12, 0.06, 2), (1024, 28, 12, 0.05, 0), (100, 11, 12, 0.3, 3)])
def test_decode_emulated_matches_oracle(orc, K, T, wb, p, oh):
    prm = orc.params(K)
    src = payload(K * T, seed=5).reshape(K, T)
    kc = nanorq_amd.host_kconst(K)
    done = 0
    for seed in range(1, 4):
        lost = loss_pattern(K, p, seed)
        esis = received_set(K, lost, oh)
        rep_esis = esis[esis >= K]
        rep, _, _ = orc.encode_block(src, K, T, rep_esis)
        syms = np.concatenate([src[esis[esis < K]], rep])
        ok, ref_out, _ = orc.decode_block(esis, syms, K, T)
        isis, rowsrc = decode_setup(orc, K, lost, rep_esis)
        plan = nanorq_amd.host_plan(K, isis, kc)
        hdr = nanorq_amd.plan_header(plan)
        assert (hdr["status"] == 0) == ok  # failure parity: decodable iff rank(A) == L
        if not ok:
            continue
        work = src.copy()
        work[lost] = 0xEE  # missing rows hold garbage
        lists = lt_lists(orc, K, lost, plan)
        r, inter = emu_solve(plan, kc, rowsrc, work, rep, T, prm["L"], lists, lost, work, wb)
        assert r == 1
        assert np.array_equal(work, src)
        done += 1
    assert done >= 1


@pytest.mark.parametrize("K,T,wb", [(100, 64, 16), (1024, 20, 8), (1024, 12, 4), (8192, 16, 16)])
def test_dense_fold_with_shared_multiples(orc, K, T, wb):
    """The big-workgroup variant of the dense fold (one thread per scratch row, products through the accumulator copies
    and a second ph_hdpc_reduce) on the emulator's 256-thread workgroup."""
    from emu_support import emu
    emu().emu_set_dense_shared_min_nt(1)
    try:
        _encode_case(orc, K, T, wb)
    finally:
        emu().emu_set_dense_shared_min_nt(512)


@pytest.mark.parametrize("K,T,wb", [(1024, 32, 16), (1024, 20, 8), (1024, 12, 4), (2000, 6, 2), (8192, 16, 16)])
def test_hdpc_with_register_accumulators(orc, K, T, wb):
    """The big-workgroup form of the HDPC phase (hdpc_chunk_regs: the H sums in registers over a thread's columns, one LDS
    update per thread) on the emulator's 256-thread workgroup, with the dense fold of that workgroup."""
    from emu_support import emu
    emu().emu_set_hdpc_regs(1)
    emu().emu_set_dense_shared_min_nt(1)
    try:
        _encode_case(orc, K, T, wb)
    finally:
        emu().emu_set_hdpc_regs(0)
        emu().emu_set_dense_shared_min_nt(512)


def test_failure_parity_small_blocks(orc):
    """rank(A) < L verdicts must agree with the reference algorithm (~1 % at +0)."""
    K = 12
    kc = nanorq_amd.host_kconst(K)
    nfail = 0
    rng = np.random.default_rng(7)
    for trial in range(400):
        nl = int(rng.integers(1, 7))
        lost = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
        rep_esis = (K + rng.choice(60, nl, replace=False)).astype(np.uint32)
        isis, _ = decode_setup(orc, K, lost, rep_esis)
        r, st = orc.plan_probe(K, isis)
        hdr = nanorq_amd.plan_header(nanorq_amd.host_plan(K, isis, kc))
        assert (hdr["status"] == 0) == (r == 1), (trial, lost, rep_esis)
        nfail += (r == 0)
    assert nfail > 0  # the sweep really contains singular systems


def test_emulated_forward_pass_is_sensitive_to_row_spacing(orc):
    """The emulator issues the op stream in the kernel's pipelined order (row q is read before rows q-1 and q-2
    are applied), so a plan that puts dependent levels too close must NOT come out right -- otherwise the CPU
    tier could not see a scheduling bug of either planner.  Here: the spacer rows are squeezed out."""
    K, T = 1500, 32
    p = orc.params(K)
    src = payload(K * T, seed=3).reshape(K, T)
    kc = nanorq_amd.host_kconst(K)
    plan = bytearray(nanorq_amd.host_plan(K, np.arange(p["Kp"], dtype=np.uint32), kc))
    h = nanorq_amd.plan_header(bytes(plan))
    assert h["pipe"] >= 2
    ops = nanorq_amd.plan_ops(plan, h)
    real = ops[((ops & 0xFFFF) >= 64).any(axis=1)]          # rows with at least one real op, in order
    assert len(real) <= h["nrows"] - 8                      # there were spacer rows
    squeezed = ops.copy()
    lane_nop = (np.arange(64, dtype=np.uint32) * 0x10001).astype(np.uint32)
    squeezed[:] = lane_nop
    squeezed[:len(real)] = real
    nanorq_amd.plan_ops_store(plan, squeezed, h)
    rowsrc = np.full(p["L"], ROW_ZERO, np.uint32)
    rowsrc[p["S"] + p["H"]: p["S"] + p["H"] + K] = np.arange(K, dtype=np.uint32)
    esis = np.arange(K, K + 3, dtype=np.uint32)
    out = np.zeros((3, T), np.uint8)
    r, inter = emu_solve(bytes(plan), kc, rowsrc, src, None, T, p["L"], lt_lists(orc, K, esis + (p["Kp"] - K), bytes(plan)),
                         np.arange(3), out, 16)
    ref_rep, ref_inter, _ = orc.encode_block(src, K, T, esis, want_inter=True)
    assert r != 1 or not np.array_equal(inter, ref_inter)


def _stream_hazards(plan):
    """Rows are issued as: apply row q-P, read row q.  So a slot may be read only P or more rows after its last write,
    and never written again once something read it (it must be final)."""
    h = nanorq_amd.plan_header(plan)
    ops = nanorq_amd.plan_ops(plan, h)
    P = h["pipe"]
    row = np.repeat(np.arange(len(ops)), 64)
    flat = ops.ravel()
    real = (flat & 0xFFFF) >= 64
    dst, src, row = (flat[real] & 0xFFFF).astype(np.int64), (flat[real] >> 16).astype(np.int64), row[real]
    n = int(max(dst.max(), src.max())) + 1
    last_write = np.full(n, -10 ** 6)
    np.maximum.at(last_write, dst, row)
    first_read = np.full(n, 10 ** 6)
    np.minimum.at(first_read, src, row)
    return int((first_read - last_write < P).sum())  # slots whose first read comes less than P rows after their last write


@pytest.mark.parametrize("K,p,oh", [(100, 0.3, 3), (1024, 0.05, 0), (1024, 0.2, 20), (8192, 0.1, 0)])
def test_op_stream_respects_the_row_pipeline(orc, K, p, oh):
    kc = nanorq_amd.host_kconst(K)
    for seed in (1, 2, 3):
        lost = loss_pattern(K, p, seed)
        esis = received_set(K, lost, oh)
        rep_esis = esis[esis >= K]
        host = nanorq_amd.host_plan(K, decode_setup(orc, K, lost, rep_esis)[0], kc)
        assert _stream_hazards(host) == 0
        dev, hdr = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=(140 if seed != 2 else 48) * 1024)
        if hdr["status"] == 0:
            assert _stream_hazards(dev) == 0


def test_lds_budget_of_headline_config(orc):
    """K=8192: the 16-byte strip image must fit the 160 KiB LDS of one gfx950 workgroup."""
    K = 8192
    p = orc.params(K)
    kc = nanorq_amd.host_kconst(K)
    plan = nanorq_amd.host_plan(K, np.arange(p["Kp"], dtype=np.uint32), kc)
    buf = (C.c_uint8 * len(plan)).from_buffer_copy(plan)
    assert emu().emu_lds_bytes(C.addressof(buf), 16) <= 163840
    hdr = nanorq_amd.plan_header(plan)
    assert hdr["nlev"] < 600  # breadth-first peeling keeps the dependency depth low


@pytest.mark.parametrize("K,T,wb,p,oh,lds", [(10, 16, 16, 0.3, 0, 140), (10, 16, 16, 0.3, 4, 140), (100, 32, 16, 0.06, 0, 140),
                                             (100, 8, 8, 0.4, 25, 140), (1024, 16, 16, 0.05, 0, 140),
                                             (1024, 16, 16, 0.06, 52, 24), (8192, 16, 16, 0.1, 0, 140),
                                             (8192, 16, 16, 0.1, 2, 60),
                                             (8192, 16, 16, 0.55, 3, 60)])   # > 4096 missing symbols: the list offsets by the scan phases
def test_device_planner_emulated_matches_oracle(orc, K, T, wb, p, oh, lds):
    """The GPU planner's phase code (planner_body.h), emulated, then the emulated solve: decoded data must be
    the oracle's; `lds` (KiB) small enough forces the peeling state out of LDS into the HBM workspace.  Every case
    also runs SEGMENTED, the way big blocks run on the GPU (planner_seq.h): part 1, the HDPC fold by two "workgroups"
    (nrq_mh_kernel), part 2 with pl_shared restored from the workspace, W transposed afterwards (nrq_wt_kernel) --
    and must produce the identical plan."""
    prm = orc.params(K)
    src = payload(K * T, seed=15).reshape(K, T)
    kc = nanorq_amd.host_kconst(K)
    done = 0
    for seed in range(1, 4):
        lost = loss_pattern(K, p, seed)
        if len(lost) == 0:
            continue
        esis = received_set(K, lost, oh)
        rep_esis = esis[esis >= K]
        rep, _, _ = orc.encode_block(src, K, T, rep_esis)
        syms = np.concatenate([src[esis[esis < K]], rep])
        ok, ref_out, _ = orc.decode_block(esis, syms, K, T)
        plan, hdr = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=lds * 1024)
        assert (hdr["status"] == 0) == ok
        plan2, hdr2 = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=lds * 1024, split=True)
        assert hdr2["status"] == hdr["status"]
        if not ok:
            continue
        assert plan2 == plan, "the segmented run built a different plan"
        if lds < 140:   # the peeling state is out of LDS: by default its counts and flags are mirrored there (compact state,
            # big blocks on the GPU); without the mirror the same plan must come out
            plan4, hdr4 = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=lds * 1024, compact=False)
            assert plan4 == plan, "the compact peeling state built a different plan"
        if K <= 1024:   # small blocks run with small queues / claim lists / Gauss-Jordan flags behind pl_shared
            plan3, hdr3 = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=lds * 1024, caps=(512, 384))
            assert plan3 == plan
        _, rowsrc = decode_setup(orc, K, lost, rep_esis)
        work = src.copy()
        work[lost] = 0x77
        lists = lt_lists(orc, K, lost, plan)
        r, inter = emu_solve(plan, kc, rowsrc, work, rep, T, prm["L"], lists, lost, work, wb)
        assert r == 1 and np.array_equal(work, src)
        # the segmented run with the entry pass over the matrix dealt out over two "workgroups" (nrq_wentry_kernel: counters in
        # the workspace, column levels in HBM): ops land in other lanes of their groups, so the plan differs byte for byte --
        # it must solve the block all the same
        plan5, hdr5 = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=lds * 1024, split=True, wentry=True)
        assert hdr5["status"] == 0 and [hdr5[k] for k in ("npiv", "u", "nlev", "n_xor_ops")] == [hdr[k] for k in ("npiv", "u", "nlev", "n_xor_ops")]
        work5 = src.copy()
        work5[lost] = 0x55
        r5, _ = emu_solve(plan5, kc, rowsrc, work5, rep, T, prm["L"], lt_lists(orc, K, lost, plan5), lost, work5, wb)
        assert r5 == 1 and np.array_equal(work5, src)
        done += 1
    assert done >= 1


def test_device_planner_failure_parity(orc):
    K = 12
    kc = nanorq_amd.host_kconst(K)
    rng = np.random.default_rng(11)
    nfail = 0
    for trial in range(300):
        nl = int(rng.integers(1, 7))
        lost = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
        rep_esis = (K + rng.choice(60, nl, replace=False)).astype(np.uint32)
        isis, _ = decode_setup(orc, K, lost, rep_esis)
        r, _ = orc.plan_probe(K, isis)
        _, hdr = emu_device_plan(K, kc, lost, rep_esis)
        assert (hdr["status"] == 0) == (r == 1), (trial, lost, rep_esis)
        nfail += (r == 0)
    assert nfail > 0


def test_device_planner_rejects_bad_input(orc):
    K = 100
    kc = nanorq_amd.host_kconst(K)
    assert emu_device_plan(K, kc, [5, 3], [100, 101])[1]["status"] == 1      # not ascending
    assert emu_device_plan(K, kc, [5, 7], [100])[1]["status"] == 1           # fewer repair symbols than gaps
    assert emu_device_plan(K, kc, [5, 200], [100, 101])[1]["status"] == 1    # ESI outside the block
    assert emu_device_plan(K, kc, [5, 7], [100, 50])[1]["status"] == 1       # repair ESI below K


def test_device_planner_takes_more_symbols_when_rank_deficient(orc):
    """Lazy use of spare symbols: planned with exactly K symbols; where that system is rank deficient the planner
    adds the next repair symbol(s) as extra rows without re-peeling.  The decoded block must equal the source."""
    K, T = 12, 16
    prm = orc.params(K)
    kc = nanorq_amd.host_kconst(K)
    src = payload(K * T, seed=21).reshape(K, T)
    rng = np.random.default_rng(5)
    took_extra = 0
    for trial in range(900):
        nl = int(rng.integers(1, 7))
        lost = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
        rep_esis = (K + rng.choice(60, nl + 4, replace=False)).astype(np.uint32)
        isis0, _ = decode_setup(orc, K, lost, rep_esis[:nl])
        r0, _ = orc.plan_probe(K, isis0)
        plan, hdr = emu_device_plan(K, kc, lost, rep_esis, use=nl)
        if hdr["status"] != 0:
            # even all nl+4 symbols must then be insufficient for the reference algorithm
            isis_all, _ = decode_setup(orc, K, lost, rep_esis)
            assert orc.plan_probe(K, isis_all)[0] == 0
            continue
        nextra = hdr["M"] - prm["L"]
        assert (nextra == 0) == (r0 == 1)
        took_extra += nextra > 0
        rep, _, _ = orc.encode_block(src, K, T, rep_esis[:nl + nextra])
        _, rowsrc = decode_setup(orc, K, lost, rep_esis[:nl + nextra])
        work = src.copy()
        work[lost] = 0x11
        lists = lt_lists(orc, K, lost, plan)
        r, _ = emu_solve(plan, kc, rowsrc, work, rep, T, prm["L"], lists, lost, work, 16)
        assert r == 1 and np.array_equal(work, src), trial
    assert took_extra > 0


@pytest.mark.parametrize("K,T,wb,lds", [(10, 8, 16, 140), (100, 36, 8, 140), (1024, 20, 16, 140), (1024, 12, 4, 24), (8192, 16, 16, 140)])
def test_device_planner_builds_encode_plans(orc, K, T, wb, lds):
    """Encode plans of big blocks are built by the GPU planner (a job without missing symbols, nrq_planjob::mode = 1,
    what nrq_precalculate enqueues for K' from 12000 up): its phase code on the CPU, the plan through the emulated
    solve workgroup, against the oracle -- intermediate and repair symbols."""
    p = orc.params(K)
    src = payload(K * T, seed=12).reshape(K, T)
    kc = nanorq_amd.host_kconst(K)
    plan, hdr = emu_device_plan(K, kc, [], [], lds_bytes=lds * 1024, encode=True)
    assert hdr["status"] == 0 and hdr["npiv"] + hdr["u"] == p["L"] and hdr["M"] == p["L"]
    rowsrc = np.full(p["L"], ROW_ZERO, np.uint32)
    rowsrc[p["S"] + p["H"]: p["S"] + p["H"] + K] = np.arange(K, dtype=np.uint32)
    esis = np.arange(K, K + 5, dtype=np.uint32)
    lists = lt_lists(orc, K, esis + (p["Kp"] - K), plan)
    out = np.zeros((5, T), np.uint8)
    r, inter = emu_solve(plan, kc, rowsrc, src, None, T, p["L"], lists, np.arange(5), out, wb)
    ref_rep, ref_inter, _ = orc.encode_block(src, K, T, esis, want_inter=True)
    assert r == 1 and np.array_equal(inter, ref_inter) and np.array_equal(out, ref_rep)
    # without the mode flag a job without missing symbols stays "nothing to do"
    assert emu_device_plan(K, kc, [], [])[1]["status"] == 1


def test_dense_stage_over_the_dead_rowstate_image_gives_the_same_plan(orc):
    """Blocks of ~8500 to ~11000 symbols: the peeling state fits the LDS only if the dense stage takes over the rowstate
    image once peeling is done (planner_body.h pl_state_in_lds == 2).  Same plan as with an LDS region that holds state and
    dense-stage reserve side by side (the state in the workspace peels with another search order: its plan may differ), and
    the plan solves the block."""
    import ctypes as C
    K, T = 9400, 16
    prm = orc.params(K)
    kc = nanorq_amd.host_kconst(K)
    src = payload(K * T, seed=33).reshape(K, T)
    lost = loss_pattern(K, 0.1, 7)
    esis = received_set(K, lost, 0)
    rep_esis = esis[esis >= K]
    rep, _, _ = orc.encode_block(src, K, T, rep_esis)
    ok, _, _ = orc.decode_block(esis, np.concatenate([src[esis[esis < K]], rep]), K, T)
    plan_side, hdr_side = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=160 * 1024)
    plan_ovl, hdr_ovl = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=134 * 1024)
    assert (hdr_ovl["status"] == 0) == ok and hdr_side["status"] == hdr_ovl["status"]
    assert plan_ovl == plan_side
    if ok:
        _, rowsrc = decode_setup(orc, K, lost, rep_esis)
        work = src.copy()
        work[lost] = 0x77
        r, _ = emu_solve(plan_ovl, kc, rowsrc, work, rep, T, prm["L"], lt_lists(orc, K, lost, plan_ovl), lost, work, 8)
        assert r == 1 and np.array_equal(work, src)


@pytest.mark.parametrize("K,p,oh", [(300, 0.3, 0), (1024, 0.12, 0), (1024, 0.12, 2), (4000, 0.2, 0)])
def test_blocked_gauss_jordan_gives_the_same_plan(K, p, oh):
    """Big matrices run the GF(2) Gauss-Jordan a panel of 32 columns at a time (planner_body.h pl_gjp_*): same pivot rule, so
    the plan must be the one the column-at-a-time loop produces, byte for byte -- forced here for sizes the CPU tier affords
    (rank-deficient blocks included: they take spare symbols through the single-column steps afterwards)."""
    from emu_support import pemu
    kc = nanorq_amd.host_kconst(K)
    for seed in range(3):
        lost = loss_pattern(K, p, seed + 40)
        rep_esis = np.arange(K, K + len(lost) + oh + 3, dtype=np.uint32)
        pemu().emu_plan_set_gj_block_min(1 << 30)
        pemu().emu_plan_set_gj_wave(0)
        try:
            ref, ref_hdr = emu_device_plan(K, kc, lost, rep_esis, use=len(lost) + oh)          # a column at a time
            pemu().emu_plan_set_gj_block_min(0)
            blk, blk_hdr = emu_device_plan(K, kc, lost, rep_esis, use=len(lost) + oh)          # panels, two phases per column
            pemu().emu_plan_set_gj_wave(1)
            wav, wav_hdr = emu_device_plan(K, kc, lost, rep_esis, use=len(lost) + oh)          # panels, a panel's columns in one phase
        finally:
            pemu().emu_plan_set_gj_block_min(4)
            pemu().emu_plan_set_gj_wave(1)
        assert ref_hdr["status"] == blk_hdr["status"] == wav_hdr["status"]
        assert ref == blk
        assert ref == wav


def _atomic_cycles_per_row(plan):
    """LDS cycles of a row's 64-lane atomic under the bank model of plan.h "lane placement": four blocks of 16 contiguous lanes,
    a block costs as many cycles as its busiest class (target slot mod 8) has different slots."""
    ops = nanorq_amd.plan_ops(plan)
    dst = (ops & 0xFFFF).astype(np.int64)
    live = ((ops & 0xFFFF) >= 64).any(axis=1)
    cost = []
    for row in dst[live]:
        c = 0
        for g in range(4):
            s = np.unique(row[16 * g:16 * g + 16])
            c += np.bincount(s % 8, minlength=8).max()
        cost.append(c)
    return float(np.mean(cost))


def test_lanes_are_placed_by_bank_class():
    """Both planners place the ops of a level group two per bank class and 16-lane block (8 cycles per row at best; lanes in
    arrival order cost ~16): host plan and emulated device plan of a decode."""
    K = 3000
    kc = nanorq_amd.host_kconst(K)
    p = nanorq_amd.params(K)
    host = nanorq_amd.host_plan(K, np.arange(p["Kp"], dtype=np.uint32), kc)
    assert _atomic_cycles_per_row(host) < 11.5   # (the thin level groups: the wide, well-filled group of the GF(2)
    lost = loss_pattern(K, 0.1, 77)              #  combinations left the stream for the table phase)
    rep_esis = np.arange(K, K + len(lost) + 3, dtype=np.uint32)
    dev, hdr = emu_device_plan(K, kc, lost, rep_esis, use=len(lost))
    assert hdr["status"] == 0
    assert _atomic_cycles_per_row(dev) < 11.5


def test_small_planner_state_reports_overflow(orc):
    """Capacities of the arrays behind pl_shared are the launch's choice; a block that does not fit them must come back
    as a capacity failure (the host planner then takes it), never as a wrong plan."""
    K = 1024
    kc = nanorq_amd.host_kconst(K)
    lost = loss_pattern(K, 0.3, 5)
    esis = received_set(K, lost, 0)
    rep_esis = esis[esis >= K]
    _, ok_hdr = emu_device_plan(K, kc, lost, rep_esis)
    _, hdr = emu_device_plan(K, kc, lost, rep_esis, caps=(64, 16))
    assert ok_hdr["status"] == 0 and hdr["status"] == 1


@pytest.mark.parametrize("kind", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("split", [False, True])
def test_plan_check_finds_damaged_books(orc, kind, split):
    """The device-only peeling forms (chained claims, batched inactivation events, pivot sort) are race-based by design, and a
    wrong plan would decode to silently wrong bytes: behind the peel every planner instance checks that its books describe a
    permutation (planner_body.h pl_check_a / _b / _c, and the two rules of the entry pass).  Here the emulator damages the
    books the way a lost race would -- two pivots on one row, pivot columns swapped, a row claimed before its sources were
    final, a column left in V, two inactive columns on one W bit, pivots trading rows -- and every kind must come back as a
    capacity failure (status 1 with reason PL_FAIL_CAPACITY = 2: the host planner then takes the block), the undamaged run as
    a plan."""
    from emu_support import pemu
    K = 1024
    kc = nanorq_amd.host_kconst(K)
    lost = loss_pattern(K, 0.2, 7)
    esis = received_set(K, lost, 0)
    rep_esis = esis[esis >= K]
    _, ok_hdr = emu_device_plan(K, kc, lost, rep_esis, split=split, wentry=split)
    assert ok_hdr["status"] == 0
    pemu().emu_plan_set_sabotage(kind)
    try:
        _, hdr = emu_device_plan(K, kc, lost, rep_esis, split=split, wentry=split)
    finally:
        pemu().emu_plan_set_sabotage(0)
    assert hdr["status"] == 1 and hdr["reserved0"] == 2 and hdr["fail_site"] != 0, (kind, hdr["status"], hdr["reserved0"], hdr["fail_site"])


def test_host_planner_fills_the_rows_of_the_stream():
    """plan.h "early ops by release": the host planner deals the early ops with wide windows out in the order of their release,
    so that a group takes what its rows have lanes for -- the stream's rows are nearly full (by hash alone a quarter of it was
    padding: ~50 real ops per row of 64), and the forward passes still come out right (the emulated pipeline is sensitive to
    row spacing: test_emulated_forward_pass_is_sensitive_to_row_spacing)."""
    K = 8192   # (55.6 k ops on 343 levels: 870 rows are needed; small blocks are bound by two rows per level instead)
    kc = nanorq_amd.host_kconst(K)
    p = nanorq_amd.params(K)
    plan = nanorq_amd.host_plan(K, np.arange(p["Kp"], dtype=np.uint32), kc)
    ops = nanorq_amd.plan_ops(plan)
    real = ((ops & 0xFFFF) >= 64).sum(1)
    h = nanorq_amd.plan_header(plan)
    rows = real[:h["nrows"]]
    assert rows.sum() / max(1, (rows > 0).sum()) > 60.0, "stream rows are not filled"
    assert h["nrows"] < 950
