"""Pin the CPU oracle against the reference-derived known answers of SURVEY.md section 8(c)
(the reference ships no test vectors of its own: SURVEY.md section 4) and RFC 6330 invariants."""
import hashlib
import json
import os

import numpy as np
import pytest

from util import kat_payload, payload, loss_pattern, received_set

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# SURVEY.md section 8(c) "PROBE KATs": reference lib/*.c output, payload = kat_payload, one block, Al=8
KAT_SMALL = {10: "1bad540de9c8d3c2", 11: "7db6a373c0213e10", 12: "78e7c05eb6bfe9aa"}
KAT_SHA = [
    (100, 1024, 100, 110, "6a95935839af8cba9d0921efd4a08d7cd103ec95e7318044667162dc5e59f79a"),
    (1024, 1280, 1024, 1076, "835df9f9100883e0de8308e15d809ca4593c50da043713486762c0ad2bfe79e3"),
    (8192, 1280, 8192, 8208, "35721ecf72a443010134a95fcec8920b550d47b30f41a75ba0f96179b72f99b0"),
]
# SURVEY.md section 8 schedule statistics of the reference planner (encode rows): (K, i, u, recorded, n1, nB, n0)
SCHED = [(100, 102, 26, 2257, 1744, 1244, 9), (1024, 1031, 70, 22473, 21681, 10926, 10),
         (8192, 8209, 207, 186913, 189238, 91868, 11)]
# SURVEY.md section 8 parameter table
PARAMS = {100: (101, 562, 17, 10, 113, 128, 15, 17), 1024: (1032, 824, 59, 10, 1051, 1101, 50, 53),
          8192: (8194, 212, 211, 11, 8273, 8416, 143, 149), 27000: (27111, 21, 503, 13, 27367, 27627, 260, 263),
          56403: (56403, 471, 907, 16, 56951, 57326, 375, 379)}


def test_params_table(orc):
    for K, (kp, j, s, h, w, l, p, p1) in PARAMS.items():
        q = orc.params(K)
        assert (q["Kp"], q["J"], q["S"], q["H"], q["W"], q["L"], q["P"], q["P1"]) == (kp, j, s, h, w, l, p, p1)
    with pytest.raises(ValueError):
        orc.params(56404)


def test_gf256_field(orc):
    e, l, inv = orc.gf_tables()
    assert e[0] == 1 and e[1] == 2 and e[8] == 0x1D and e[255] == 1  # x^8 = x^4+x^3+x^2+1 (RFC 6330 5.7)
    assert sorted(int(x) for x in e[:255]) == list(range(1, 256))
    for v in range(1, 256):
        assert e[l[v]] == v and e[(int(l[v]) + int(l[inv[v]])) % 255] == 1


def test_kat_small_symbols(orc):
    rep, _, st = orc.encode_block(kat_payload(80), 10, 8, [10, 11, 12])
    assert {10 + k: rep[k].tobytes().hex() for k in range(3)} == KAT_SMALL


@pytest.mark.parametrize("K,T,lo,hi,sha", KAT_SHA)
def test_kat_sha(orc, K, T, lo, hi, sha):
    rep, inter, st = orc.encode_block(kat_payload(K * T), K, T, list(range(lo, hi)), want_inter=True)
    assert hashlib.sha256(rep.tobytes()).hexdigest() == sha
    # systematic property: LT(C, esi) == source[esi]
    src = kat_payload(K * T).reshape(K, T)
    for esi in (0, 1, K // 2, K - 1):
        acc = np.zeros(T, np.uint8)
        for c in orc.lt_columns(K, esi):
            acc ^= inter[c]
        assert np.array_equal(acc, src[esi])


@pytest.mark.parametrize("K,i,u,rec,n1,nB,n0", SCHED)
def test_schedule_statistics_match_reference(orc, K, i, u, rec, n1, nB, n0):
    p = orc.params(K)
    r, st = orc.plan_probe(K, np.arange(p["Kp"], dtype=np.uint32))
    assert r == 1
    assert (st["i"], st["u"], st["recorded_ops"], st["n1"], st["nB"], st["n0"]) == (i, u, rec, n1, nB, n0)


def test_simd_rows_equal_scalar(orc):
    rng = np.random.default_rng(3)
    for n in (1, 31, 32, 33, 1280, 4099):
        src = rng.integers(0, 256, n, dtype=np.uint8)
        for beta in (1, 2, 0x1D, 0xFF, 0x53):
            d0 = rng.integers(0, 256, n, dtype=np.uint8)
            d1 = d0.copy()
            orc.set_simd(0); orc.row_axpy(d0, src, beta)
            orc.set_simd(1); orc.row_axpy(d1, src, beta)
            assert np.array_equal(d0, d1)
            s0, s1 = src.copy(), src.copy()
            orc.set_simd(0); orc.row_scal(s0, beta)
            orc.set_simd(1); orc.row_scal(s1, beta)
            assert np.array_equal(s0, s1)
    orc.set_simd(1)


def test_gfni_rows_equal_scalar(orc):
    """The AVX-512 + GFNI row kernels (vgf2p8affineqb with an 8x8 bit matrix per constant: GFNI's own multiply is fixed to the AES
    polynomial, RFC 6330's field is 0x11D) against the scalar log/antilog form, every constant; and a whole encode + decode with
    them gives the bytes of the AVX2 run.  Without the ISA set_simd(2) falls back, and the comparison is trivially true."""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, 1280 + 37, dtype=np.uint8)
    for beta in range(256):
        d0 = rng.integers(0, 256, len(src), dtype=np.uint8)
        d2 = d0.copy()
        orc.set_simd(0); orc.row_axpy(d0, src, beta)
        orc.set_simd(2); orc.row_axpy(d2, src, beta)
        assert np.array_equal(d0, d2), beta
        if beta:
            s0, s2 = src.copy(), src.copy()
            orc.set_simd(0); orc.row_scal(s0, beta)
            orc.set_simd(2); orc.row_scal(s2, beta)
            assert np.array_equal(s0, s2), beta
    K, T = 300, 192
    blk = rng.integers(0, 256, (K, T), dtype=np.uint8)
    esis = np.arange(K, K + 20, dtype=np.uint32)
    try:
        orc.set_simd(1); r1, i1, _ = orc.encode_block(blk, K, T, esis, want_inter=True)
        orc.set_simd(2); r2, i2, _ = orc.encode_block(blk, K, T, esis, want_inter=True)
    finally:
        orc.set_simd(1)
    assert np.array_equal(r1, r2) and np.array_equal(i1, i2)


@pytest.mark.parametrize("K,T,p,oh", [(10, 8, 0.3, 0), (100, 1024, 0.06, 0), (100, 1024, 0.06, 2),
                                      (1024, 1280, 0.06, 52), (1024, 1280, 0.05, 0), (1024, 64, 0.5, 3)])
def test_oracle_roundtrip(orc, K, T, p, oh):
    src = payload(K * T, seed=1).reshape(K, T)
    ok_any = False
    for seed in range(1, 4):
        lost = loss_pattern(K, p, seed)
        esis = received_set(K, lost, oh)
        rep_esis = esis[esis >= K]
        rep, _, _ = orc.encode_block(src, K, T, rep_esis)
        syms = np.concatenate([src[esis[esis < K]], rep]) if len(rep) else src[esis[esis < K]]
        ok, out, st = orc.decode_block(esis, syms, K, T)
        if ok:
            ok_any = True
            assert np.array_equal(out, src)
            assert st["gaps"] == len(lost) and st["overhead"] == oh
    assert ok_any


def test_oracle_decode_needs_enough_symbols(orc):
    K, T = 100, 16
    src = payload(K * T).reshape(K, T)
    lost = np.array([3, 50, 77], np.uint32)
    rep, _, _ = orc.encode_block(src, K, T, [100, 101])
    esis = np.concatenate([np.setdiff1d(np.arange(K, dtype=np.uint32), lost), [100, 101]]).astype(np.uint32)
    syms = np.concatenate([src[esis[:-2]], rep])
    ok, out, st = orc.decode_block(esis, syms, K, T)
    assert not ok and st["gaps"] == 3


def test_oracle_add_symbol_semantics(orc):
    # duplicates ignored, ESI > 2K' rejected, arrival order of repair symbols defines row placement
    K, T = 50, 8
    src = payload(K * T, seed=5).reshape(K, T)
    rep, _, _ = orc.encode_block(src, K, T, [50, 51, 52, 60])
    keep = [e for e in range(K) if e not in (0, 49)]
    esis = np.array(keep + [60, 60, 500000, 52, 50], np.uint32)
    syms = np.concatenate([src[keep], rep[[3, 3]], np.zeros((1, T), np.uint8), rep[[2, 0]]])
    ok, out, st = orc.decode_block(esis, syms, K, T)
    assert ok and np.array_equal(out, src) and st["overhead"] == 1
