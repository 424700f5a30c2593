"""-m gpu tier: the HIP path (through the C ABI of include/nanorq_hip.h) against the CPU oracle,
the SURVEY section 8(c) known answers and, at BASELINE.json's full sizes, size-independent properties
(encode -> erase -> decode round trips, the systematic property).  Bit-exact everywhere."""
import hashlib

import numpy as np
import pytest

import nanorq_amd
from test_oracle_kat import KAT_SHA, KAT_SMALL
from util import kat_payload, loss_pattern, payload, received_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_support
    gpu_support.ctx()
    return gpu_support


def test_kat_small(G):
    rep, _ = G.gpu_encode(kat_payload(80).reshape(1, 10, 8), 10, 8, [10, 11, 12])
    assert {10 + k: rep[0, k].tobytes().hex() for k in range(3)} == KAT_SMALL


@pytest.mark.parametrize("K,T,lo,hi,sha", KAT_SHA)
def test_kat_sha(G, K, T, lo, hi, sha):
    rep, _ = G.gpu_encode(kat_payload(K * T).reshape(1, K, T), K, T, list(range(lo, hi)))
    assert hashlib.sha256(rep[0].tobytes()).hexdigest() == sha


@pytest.mark.parametrize("K,T", [(10, 8), (10, 40), (55, 4), (100, 1024), (101, 20), (500, 72), (1024, 1280),
                                 (1033, 16), (4000, 48)])
def test_encode_matches_oracle(G, orc, K, T):
    nblk = 3
    src = np.stack([payload(K * T, seed=3, block=b).reshape(K, T) for b in range(nblk)])
    esis = np.array([K, K + 1, K + 5, K + 1000, (1 << 24) - 1], np.uint32)
    rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
    for b in range(nblk):
        r_rep, r_int, _ = orc.encode_block(src[b], K, T, esis, want_inter=True)
        assert np.array_equal(inter[b], r_int), "intermediate symbols differ (block %d)" % b
        assert np.array_equal(rep[b], r_rep)


@pytest.mark.parametrize("K,T,p,oh", [(10, 16, 0.3, 0), (100, 1024, 0.06, 0), (100, 64, 0.06, 3), (100, 8, 0.5, 40),
                                      (1024, 1280, 0.05, 0), (1024, 1280, 0.06, 52), (1024, 24, 0.3, 1),
                                      (8192, 32, 0.1, 0), (8192, 32, 0.1, 2), (8192, 32, 0.1, 11)])
def test_decode_matches_oracle(G, orc, K, T, p, oh):
    nblk = 4
    src = np.stack([payload(K * T, seed=9, block=b).reshape(K, T) for b in range(nblk)])
    lost = [loss_pattern(K, p, seed=21, block=b) for b in range(nblk)]
    lost[0] = lost[0][:0] if K > 10 else lost[0]  # one block with nothing missing
    rep_esis = [np.arange(K, K + len(l) + (oh if len(l) else 0), dtype=np.uint32) for l in lost]
    reps, work, exp_ok = [], src.copy(), []
    for b in range(nblk):
        r, _, _ = orc.encode_block(src[b], K, T, rep_esis[b])
        reps.append(r)
        work[b][lost[b]] = 0x5A
        esis = received_set(K, lost[b], oh if len(lost[b]) else 0)
        syms = np.concatenate([src[b][esis[esis < K]], r]) if len(r) else src[b][esis[esis < K]]
        ok, out, _ = orc.decode_block(esis, syms, K, T)
        exp_ok.append(ok)
    st, out, _ = G.gpu_decode(work, K, T, lost, rep_esis, reps)
    for b in range(nblk):
        assert bool(st[b]) == exp_ok[b]
        if exp_ok[b]:
            assert np.array_equal(out[b], src[b]), "block %d" % b
        else:
            assert np.array_equal(out[b], work[b])  # an undecodable block is left untouched


def test_decode_too_few_symbols_and_retry(G, orc):
    K, T = 100, 32
    src = payload(K * T, seed=2).reshape(1, K, T)
    lost = [np.array([3, 50, 77], np.uint32)]
    rep, _ = G.gpu_encode(src, K, T, [100, 101, 102])
    work = src.copy(); work[0][lost[0]] = 0
    st, out, _ = G.gpu_decode(work, K, T, lost, [np.array([100, 101], np.uint32)], [rep[0][:2]])
    assert st[0] == 0
    st, out, _ = G.gpu_decode(work, K, T, lost, [np.array([100, 101, 102], np.uint32)], [rep[0]])
    assert st[0] == 1 and np.array_equal(out[0], src[0])


def test_failure_parity_sweep(G, orc):
    """rank-deficient systems must be reported exactly where the reference algorithm reports them."""
    from emu_support import decode_setup
    K, T = 12, 8
    src = payload(K * T, seed=4).reshape(K, T)
    rng = np.random.default_rng(7)
    lost_l, resi_l, reps_l, expect = [], [], [], []
    all_rep, _ = G.gpu_encode(src.reshape(1, K, T), K, T, np.arange(K, K + 60, dtype=np.uint32))
    for trial in range(200):
        nl = int(rng.integers(1, 7))
        lost = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
        resi = (K + rng.choice(60, nl, replace=False)).astype(np.uint32)
        isis, _ = decode_setup(orc, K, lost, resi)
        r, _ = orc.plan_probe(K, isis)
        lost_l.append(lost); resi_l.append(resi); reps_l.append(all_rep[0][resi - K]); expect.append(r == 1)
    work = np.repeat(src.reshape(1, K, T), 200, axis=0).copy()
    for b in range(200):
        work[b][lost_l[b]] = 0xFF
    st, out, _ = G.gpu_decode(work, K, T, lost_l, resi_l, reps_l)
    assert [bool(x) for x in st] == expect and not all(expect)
    for b in range(200):
        if expect[b]:
            assert np.array_equal(out[b], src)


def test_gen_symbols_from_hbm(G, orc):
    K, T, nblk = 300, 96, 2
    p = orc.params(K)
    src = np.stack([payload(K * T, seed=6, block=b).reshape(K, T) for b in range(nblk)])
    c = G.ctx()
    L = p["L"]
    d_src = c.alloc(nblk * K * T); d_int = c.alloc(nblk * L * T)
    isis = np.array([0, 5, K - 1, p["Kp"], p["Kp"] + 7, p["Kp"] + 100000], np.uint32)
    d_out = c.alloc(nblk * len(isis) * T)
    c.upload(d_src, src)
    c.encode_blocks(K, T, nblk, d_src, K * T, 0, 0, [], d_int, L * T)
    c.gen_symbols(K, T, nblk, d_int, L * T, isis, d_out, len(isis) * T)
    got = c.download(d_out, nblk * len(isis) * T).reshape(nblk, len(isis), T)
    for b in range(nblk):
        assert np.array_equal(got[b, 0], src[b, 0]) and np.array_equal(got[b, 2], src[b, K - 1])  # systematic
        rep, _, _ = orc.encode_block(src[b], K, T, isis[3:] - (p["Kp"] - K))
        assert np.array_equal(got[b, 3:], rep)
    for d in (d_src, d_int, d_out):
        c.free(d)


def _roundtrip(G, K, T, nblk, p, oh, seed):
    src = np.stack([payload(K * T, seed=seed, block=b).reshape(K, T) for b in range(nblk)])
    lost = [loss_pattern(K, p, seed=seed + 1, block=b) for b in range(nblk)]
    nrep = max(len(l) for l in lost) + oh
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep, _ = G.gpu_encode(src, K, T, esis)
    work = src.copy()
    for b in range(nblk):
        work[b][lost[b]] = 0
    st, out, _ = G.gpu_decode(work, K, T, lost, [esis[:len(l) + oh] for l in lost],
                              [rep[b][:len(lost[b]) + oh] for b in range(nblk)])
    return st, out, src


@pytest.mark.parametrize("K,T,nblk", [(100, 200, 70), (333, 1, 9), (64, 136, 130), (1024, 1288, 16)])
def test_launch_shapes(G, orc, K, T, nblk):
    """Work distribution of the persistent solve kernel: block counts that are / are not multiples of 8 (blocks are
    dealt to XCDs by octets), symbol sizes whose last 128-byte line group is partial, a single byte column; every
    block must decode, a few are compared with the oracle byte for byte."""
    st, out, src = _roundtrip(G, K, T, nblk, 0.12, 3, seed=61)
    assert st.all()
    assert np.array_equal(out, src)
    esis = np.arange(K, K + 5, dtype=np.uint32)
    rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
    for b in (0, nblk // 2, nblk - 1):
        r_rep, r_int, _ = orc.encode_block(src[b], K, T, esis, want_inter=True)
        assert np.array_equal(inter[b], r_int) and np.array_equal(rep[b], r_rep), "block %d" % b


@pytest.mark.parametrize("K,T,nblk", [(256, 80, 300), (700, 80, 90), (1500, 80, 40), (2048, 144, 24), (2300, 80, 24),
                                      (2600, 80, 24), (3000, 80, 16), (4500, 80, 16), (5200, 80, 16)])
def test_kernel_variant_boundaries(G, orc, K, T, nblk):
    """Block sizes either side of every switch of kernel shape: the 256-thread solve variants (register budget for 5 or 4
    workgroups per CU, chosen by how many LDS images fit), the 768-thread one, the 256- and 1024-thread planner.  All
    blocks round-trip; the first and the last are compared with the oracle byte for byte (intermediate symbols too)."""
    st, out, src = _roundtrip(G, K, T, nblk, 0.08, 4, seed=K)
    assert st.all()
    assert np.array_equal(out, src)
    esis = np.array([K, K + 3, K + 77], np.uint32)
    rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
    for b in (0, nblk - 1):
        r_rep, r_int, _ = orc.encode_block(src[b], K, T, esis, want_inter=True)
        assert np.array_equal(inter[b], r_int) and np.array_equal(rep[b], r_rep), "block %d" % b


def test_roundtrip_headline_config(G):
    """BASELINE configs[1]/[2] at full size: K=8192, T=1280, 10 % loss, +2 overhead, 8 blocks."""
    st, out, src = _roundtrip(G, 8192, 1280, 8, 0.10, 2, seed=31)
    assert st.sum() >= 7
    for b in range(8):
        if st[b]:
            assert np.array_equal(out[b], src[b])


def test_headline_block_vs_oracle_checksum(G, orc):
    """One full-size K=8192/T=1280 block: repair symbols and recovered data equal the oracle's."""
    K, T = 8192, 1280
    src = payload(K * T, seed=77).reshape(K, T)
    lost = loss_pattern(K, 0.10, seed=78)
    esis = np.arange(K, K + len(lost) + 2, dtype=np.uint32)
    rep, _ = G.gpu_encode(src.reshape(1, K, T), K, T, esis)
    r_rep, _, _ = orc.encode_block(src, K, T, esis)
    assert hashlib.sha256(rep[0].tobytes()).hexdigest() == hashlib.sha256(r_rep.tobytes()).hexdigest()
    work = src.copy(); work[lost] = 0
    st, out, _ = G.gpu_decode(work.reshape(1, K, T), K, T, [lost], [esis], [rep[0]])
    assert st[0] == 1 and np.array_equal(out[0], src)


def test_headline_reception_overhead0_full_width_vs_oracle(G, orc):
    """The reception shape bench.py's headline is quoted on -- K=8192, T=1280, 10 % loss, EXACTLY K symbols received (the
    GF(256)/HDPC path, reference precode.c:365-371) -- at full symbol width, two blocks: repair symbols, intermediate symbols
    (encode and decode side) and the recovered blocks against the oracle byte for byte; a block whose K-symbol system is rank
    deficient must be reported so by both, and decodes with one more symbol."""
    K, T, nblk = 8192, 1280, 2
    src = np.stack([payload(K * T, seed=91, block=b).reshape(K, T) for b in range(nblk)])
    lost = [loss_pattern(K, 0.10, seed=92, block=b) for b in range(nblk)]
    esis = np.arange(K, K + max(len(l) for l in lost) + 2, dtype=np.uint32)
    rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
    for b in range(nblk):
        r_rep, r_int, _ = orc.encode_block(src[b], K, T, esis, want_inter=True)
        assert np.array_equal(rep[b], r_rep) and np.array_equal(inter[b], r_int), "encode of block %d differs from the oracle" % b
    work = src.copy()
    for b in range(nblk):
        work[b][lost[b]] = 0xEE
    nuse = [len(l) for l in lost]
    for attempt in range(3):
        resi = [esis[:n] for n in nuse]
        st, out, dint = G.gpu_decode(work, K, T, lost, resi, [rep[b][:nuse[b]] for b in range(nblk)], want_inter=True)
        done = True
        for b in range(nblk):
            recv = np.concatenate([np.setdiff1d(np.arange(K, dtype=np.uint32), lost[b]), resi[b]])
            ok, o_out, _ = orc.decode_block(recv, np.concatenate([src[b][recv[recv < K]], rep[b][:nuse[b]]]), K, T)
            assert bool(st[b]) == bool(ok), "verdict of block %d differs from the oracle's" % b
            if ok:
                assert np.array_equal(out[b], src[b]) and np.array_equal(o_out, src[b])
                assert np.array_equal(dint[b], inter[b]), "intermediate symbols of the decode differ (block %d)" % b
            else:
                assert np.array_equal(out[b], work[b])
                nuse[b] += 1
                done = False
        if done:
            break
    assert done


def test_roundtrip_max_k(G):
    """BASELINE configs[4] shape: K'=56403 (RFC 6330 maximum), T=1280, 20 % loss, +16 (2-byte strips)."""
    st, out, src = _roundtrip(G, 56403, 1280, 1, 0.20, 16, seed=41)
    assert st[0] == 1 and np.array_equal(out[0], src[0])


def test_roundtrip_large_symbols(G):
    """BASELINE configs[3] shape at a reduced symbol size (K=27000, T=4096) through host buffers; the full-size
    block is test_cfg4_full_size_on_device."""
    st, out, src = _roundtrip(G, 27000, 4096, 1, 0.10, 2, seed=51)
    assert st[0] == 1 and np.array_equal(out[0], src[0])


def test_max_k_overhead0_vs_oracle(G, orc):
    """BASELINE configs[4] at full size on the GF(256)/HDPC path: K'=56403 (RFC 6330 maximum), T=1280, 20 % loss,
    decode from EXACTLY K symbols (reference precode.c:365-371, not the XOR-only branch :362-363).  Repair symbols and
    the recovered block are compared with the oracle's by SHA-256; the verdict must agree if the system is singular."""
    K, T = 56403, 1280
    src = payload(K * T, seed=43).reshape(K, T)
    lost = loss_pattern(K, 0.20, seed=44)
    esis = np.arange(K, K + len(lost), dtype=np.uint32)
    rep, _ = G.gpu_encode(src.reshape(1, K, T), K, T, esis)
    r_rep, _, _ = orc.encode_block(src, K, T, esis)
    assert hashlib.sha256(rep[0].tobytes()).hexdigest() == hashlib.sha256(r_rep.tobytes()).hexdigest()
    work = src.copy()
    work[lost] = 0x77
    st, out, _ = G.gpu_decode(work.reshape(1, K, T), K, T, [lost], [esis], [rep[0]])
    keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
    ok, r_out, stt = orc.decode_block(np.concatenate([keep, esis]), np.concatenate([src[keep], r_rep]), K, T)
    assert stt["overhead"] == 0 and bool(st[0]) == ok
    if ok:
        assert hashlib.sha256(out[0].tobytes()).hexdigest() == hashlib.sha256(r_out.tobytes()).hexdigest()
        assert np.array_equal(out[0], src)
    else:
        assert np.array_equal(out[0], work)


def test_cfg4_full_size_on_device(G, orc):
    """BASELINE configs[3] at FULL size: K=27000, T=65504, one 1.77 GB source block generated on the device, 10 % loss,
    overhead 0.  Checked on the device: decode(encode) == source, the systematic property of the intermediate symbols
    (LT(C, esi) == source symbol, through nrq_gen_symbols); against the oracle: one 4096-byte column slice of the
    repair and intermediate symbols (byte columns are independent, so a slice of the block is a block of its own)."""
    import torch
    K, T, W0, WS = 27000, 65504, 20480, 4096
    p = orc.params(K)
    L = p["L"]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    src = torch.randint(0, 256, (K, T), dtype=torch.uint8, device=dev, generator=g)
    lost = loss_pattern(K, 0.10, seed=52)
    nrep = len(lost) + 2
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep = torch.empty((nrep, T), dtype=torch.uint8, device=dev)
    inter = torch.empty((L, T), dtype=torch.uint8, device=dev)
    c = G.ctx()
    torch.cuda.synchronize()
    c.encode_blocks(K, T, 1, src.data_ptr(), K * T, rep.data_ptr(), nrep * T, esis, inter.data_ptr(), L * T)
    c.sync()
    # systematic property on sampled ESIs (first, last, around the loss positions)
    probe = np.unique(np.concatenate([[0, 1, K // 2, K - 1], lost[:20]])).astype(np.uint32)
    sysout = torch.empty((len(probe), T), dtype=torch.uint8, device=dev)
    c.gen_symbols(K, T, 1, inter.data_ptr(), L * T, probe, sysout.data_ptr(), len(probe) * T)
    c.sync()
    assert torch.equal(sysout, src[torch.from_numpy(probe.astype(np.int64)).to(dev)])
    # one column slice against the oracle
    src_sl = src[:, W0:W0 + WS].contiguous().cpu().numpy()
    r_rep, r_int, _ = orc.encode_block(src_sl, K, WS, esis[:64], want_inter=True)
    assert np.array_equal(rep[:64, W0:W0 + WS].cpu().numpy(), r_rep), "repair symbols, column slice"
    assert np.array_equal(inter[:, W0:W0 + WS].cpu().numpy(), r_int), "intermediate symbols, column slice"
    del inter, sysout
    # decode from exactly K symbols (GF(256)/HDPC path); verdict as the oracle's on the slice
    work = src.clone()
    work[torch.from_numpy(lost.astype(np.int64)).to(dev)] = 0xEE
    lost_arr = lost.reshape(1, -1)
    st = c.decode_blocks(K, T, 1, work.data_ptr(), K * T, lost_arr, [len(lost)], esis[:len(lost)].reshape(1, -1), [len(lost)],
                         rep.data_ptr(), nrep * T)
    c.sync()
    keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
    rep_sl = rep[:len(lost), W0:W0 + WS].cpu().numpy()
    ok, r_out, _ = orc.decode_block(np.concatenate([keep, esis[:len(lost)]]), np.concatenate([src_sl[keep], rep_sl]), K, WS)
    assert bool(st[0]) == ok
    if not ok:  # rank deficient at overhead 0 (about 1 % of receptions): one more symbol, as a receiver would
        st = c.decode_blocks(K, T, 1, work.data_ptr(), K * T, lost_arr, [len(lost)], esis[:len(lost) + 1].reshape(1, -1),
                             [len(lost) + 1], rep.data_ptr(), nrep * T)
        c.sync()
    assert st[0] == 1 and torch.equal(work, src)


@pytest.mark.parametrize("planner", ["device", "host"])
def test_lazy_decode_takes_spare_symbols_only_when_needed(G, orc, planner):
    """nrq_decode_blocks_lazy: start from exactly K symbols; rank-deficient blocks consume spare repair symbols
    one at a time (extra constraint rows) and still decode bit-exactly; the others use none."""
    from emu_support import decode_setup
    K, T, nblk = 12, 24, 300
    c = G.ctx()
    c.set_planner(planner == "device")
    try:
        src = payload(K * T, seed=33).reshape(K, T)
        all_rep, _ = G.gpu_encode(src.reshape(1, K, T), K, T, np.arange(K, K + 64, dtype=np.uint32))
        rng = np.random.default_rng(3)
        lost = np.zeros((nblk, 8), np.uint32); resi = np.zeros((nblk, 12), np.uint32)
        nlost = np.zeros(nblk, np.uint32); reps = np.zeros((nblk, 12, T), np.uint8)
        need_more = []
        for b in range(nblk):
            nl = int(rng.integers(1, 7))
            lo = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
            es = (K + rng.choice(64, nl + 4, replace=False)).astype(np.uint32)
            lost[b, :nl] = lo; resi[b, :nl + 4] = es; nlost[b] = nl; reps[b, :nl + 4] = all_rep[0][es - K]
            isis, _ = decode_setup(orc, K, lo, es[:nl])
            need_more.append(orc.plan_probe(K, isis)[0] == 0)
        work = np.repeat(src.reshape(1, K, T), nblk, axis=0).copy()
        for b in range(nblk):
            work[b][lost[b, :nlost[b]]] = 0x3C
        d_src = c.alloc(work.nbytes); d_rep = c.alloc(reps.nbytes)
        c.upload(d_src, work); c.upload(d_rep, reps)
        st, used = c.decode_blocks_lazy(K, T, nblk, d_src, K * T, lost, nlost, resi, nlost, nlost + 4, d_rep, 12 * T)
        c.sync()
        out = c.download(d_src, work.nbytes).reshape(nblk, K, T)
        c.free(d_src); c.free(d_rep)
        assert any(need_more)
        for b in range(nblk):
            assert st[b] == 1 and np.array_equal(out[b], src), b
            assert (used[b] > nlost[b]) == need_more[b], (b, used[b], nlost[b])
    finally:
        c.set_planner(True)


@pytest.mark.parametrize("wb,split", [(4, True), (2, True), (2, False), (8, False), (12, False)])
def test_narrow_strip_paths_at_small_sizes(G, orc, wb, split):
    """The kernels big blocks use -- 12/8/4/2-byte strips and, for 4 and 2, the split solve (nrq_backsub_kernel +
    nrq_collect_kernel finishing on full-width rows) -- forced at sizes the oracle checks in no time: intermediate,
    repair and recovered symbols byte for byte, ragged symbol sizes included (T = 1, 50, 1288; for the 12-byte strip
    1288 = 107 strips and 4 bytes, moved by the aligned form, 96 = whole strips, 50 and 1 the byte-wise form)."""
    c = G.ctx()
    c.set_option("max_wb", wb)
    c.set_option("no_split", 0 if split else 1)
    try:
        # (T = 1280, 32 and 16: rows 16-byte aligned -- the movers' aligned forms; for the 12-byte strip with a last strip of 8 / 8 / 4 bytes)
        for K, T, nblk, p, oh in [(300, 1288, 3, 0.1, 0), (1024, 50, 9, 0.05, 2), (64, 1, 5, 0.2, 1), (2000, 96, 2, 0.1, 0),
                                  (300, 1280, 3, 0.1, 0), (700, 32, 9, 0.1, 1), (100, 16, 17, 0.2, 0)]:
            src = np.stack([payload(K * T, seed=K + wb, block=b).reshape(K, T) for b in range(nblk)])
            esis = np.array([K, K + 1, K + 9, K + 500], np.uint32)
            rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
            assert c.stats()["strip_bytes"] == wb
            if T % 16 == 0 and wb >= 4:
                assert c.stats()["movers_aligned"] == 1, (K, T)
            for b in (0, nblk - 1):
                r_rep, r_int, _ = orc.encode_block(src[b], K, T, esis, want_inter=True)
                assert np.array_equal(inter[b], r_int) and np.array_equal(rep[b], r_rep), (K, T, b)
            st, out, src2 = _roundtrip(G, K, T, nblk, p, oh, seed=K + 7)
            for b in range(nblk):
                assert not st[b] or np.array_equal(out[b], src2[b]), (K, T, b)
            assert st.sum() >= nblk - 1
    finally:
        c.set_option("max_wb", 16)
        c.set_option("no_split", 0)


@pytest.mark.parametrize("K,T", [(10000, 40), (11500, 28), (9000, 1288)])
def test_twelve_byte_strips_where_sixteen_do_not_fit(G, orc, K, T):
    """K between ~8500 and ~12000: the 16-byte strip image exceeds the CU's LDS, the 12-byte one fits (three forward waves, a
    dword of the strip each; nrq_device.hip widest_fit).  Chosen on its own here, checked against the oracle; with "no_wb12"
    the launch falls back to 8 bytes and gives the same bytes."""
    c = G.ctx()
    nblk = 2
    src = np.stack([payload(K * T, seed=K + 12, block=b).reshape(K, T) for b in range(nblk)])
    esis = np.array([K, K + 3, K + 70000], np.uint32)
    rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
    assert c.stats()["strip_bytes"] == 12
    r_rep, r_int, _ = orc.encode_block(src[1], K, T, esis, want_inter=True)
    assert np.array_equal(inter[1], r_int) and np.array_equal(rep[1], r_rep)
    st, out, src2 = _roundtrip(G, K, T, 3, 0.06, 2, seed=K + 5)
    assert c.stats()["strip_bytes"] in (12, 8)   # (a decode plan with many inactive columns may need the narrower image)
    for b in range(3):
        assert not st[b] or np.array_equal(out[b], src2[b]), b
    assert st.sum() >= 2
    c.set_option("no_wb12", 1)
    try:
        rep8, inter8 = G.gpu_encode(src, K, T, esis, want_inter=True)
        assert c.stats()["strip_bytes"] == 8
        assert np.array_equal(rep8, rep) and np.array_equal(inter8, inter)
    finally:
        c.set_option("no_wb12", 0)


def test_device_built_encode_plans(G, orc):
    """Encode plans of big blocks come from the device planner, built asynchronously (nrq_precalculate enqueues, the
    encode that needs the plan waits; rebuilt plans alternate between two device buffers).  Forced for small K here:
    intermediate and repair symbols against the oracle, across cache clears and back-to-back rebuilds."""
    c = G.ctx()
    c.set_option("encplan_dev_min_l", 0)
    try:
        for K, T in [(10, 8), (100, 1024), (1033, 16), (4000, 48), (8192, 32)]:
            src = np.stack([payload(K * T, seed=K + 1, block=b).reshape(K, T) for b in range(2)])
            esis = np.array([K, K + 1, K + 5, (1 << 24) - 1], np.uint32)
            want = [orc.encode_block(src[b], K, T, esis, want_inter=True) for b in range(2)]
            for rebuild in range(3):
                c.clear_plan_cache()
                c.precalculate(K)          # enqueue only
                if rebuild == 2:
                    c.clear_plan_cache()   # dropped before use: the next encode starts another build
                rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
                for b in range(2):
                    assert np.array_equal(inter[b], want[b][1]) and np.array_equal(rep[b], want[b][0]), (K, rebuild, b)
        # a block coded with a larger table row than its own
        K, Kp, T = 95, 101, 64
        src = payload(K * T, seed=5).reshape(1, K, T)
        c.clear_plan_cache()
        rep, inter = G.gpu_encode(src, K, T, [95, 96, 300], want_inter=True, Kp=Kp)
        r_rep, r_int, _ = orc.encode_block(src[0], K, T, [95, 96, 300], want_inter=True, Kp=Kp)
        assert np.array_equal(inter[0], r_int) and np.array_equal(rep[0], r_rep)
    finally:
        c.set_option("encplan_dev_min_l", 12000)
        c.clear_plan_cache()


@pytest.mark.parametrize("g", [2, 4, 8])
def test_wide_strips(G, orc, g):
    """Wide strips (g lanes of 16 bytes per element, nrq_ctx_set_option "wide_g"): an option for small blocks that is not
    selected automatically (nrq_device.hip launch_wb has the measurements); forced here -- intermediate, repair and
    recovered symbols byte for byte, symbol sizes that end inside a strip (T = 50, 136, 1288)."""
    c = G.ctx()
    c.set_option("wide_g", g)
    try:
        for K, T, nblk, p, oh in [(100, 1024, 5, 0.06, 0), (300, 1288, 3, 0.1, 0), (1024, 136, 9, 0.05, 2), (64, 50 if g <= 2 else 160, 5, 0.2, 1)]:
            src = np.stack([payload(K * T, seed=K + g, block=b).reshape(K, T) for b in range(nblk)])
            esis = np.array([K, K + 1, K + 9, K + 500], np.uint32)
            rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
            assert c.stats()["strip_bytes"] == (16 * g if K <= 100 else c.stats()["strip_bytes"])   # (wide only where two images fit a CU)
            for b in (0, nblk - 1):
                r_rep, r_int, _ = orc.encode_block(src[b], K, T, esis, want_inter=True)
                assert np.array_equal(inter[b], r_int) and np.array_equal(rep[b], r_rep), (K, T, b)
            st, out, src2 = _roundtrip(G, K, T, nblk, p, oh, seed=K + 7)
            for b in range(nblk):
                assert not st[b] or np.array_equal(out[b], src2[b]), (K, T, b)
            assert st.sum() >= nblk - 1
    finally:
        c.set_option("wide_g", 0)


def test_planner_waves_stay_in_step(G):
    """The planner chooses its next phase from workgroup-shared values; a wave that reads such a value after another wave has
    already changed it (in the phase that follows) would take the other branch and the workgroup's barriers pair up across
    different phases (planner_seq.h PL_STEER_SYNC).  Sixteen waves on small blocks made that happen in every fifth plan: here
    600 plans with the 1024-thread planner forced -- every block planned on the device, decoded, equal to its source."""
    c = G.ctx()
    c.set_option("plan_big_wg", 1)
    try:
        K, T, nblk = 1024, 64, 600
        st, out, src = _roundtrip(G, K, T, nblk, 0.05, 3, seed=808)
        assert c.stats()["host_planned"] == 0
        assert st.all()
        assert np.array_equal(out, src)
    finally:
        c.set_option("plan_big_wg", 0)


def test_five_workgroups_per_cu_variant(G, orc):
    """The 256-thread solve workgroup also exists compiled for five per CU (96 registers per thread; option "small_waves4"
    = 0 selects it where five strip images fit): same bytes as the oracle."""
    c = G.ctx()
    c.set_option("small_waves4", 0)
    c.set_option("no_tiny", 1)   # (from seven strip images per CU on the single-wave workgroups would take over)
    try:
        for K, T, nblk, p in [(1000, 1280, 6, 0.06), (700, 333, 5, 0.1)]:
            src = np.stack([payload(K * T, seed=K + 3, block=b).reshape(K, T) for b in range(nblk)])
            esis = np.array([K, K + 1, K + 9, K + 500], np.uint32)
            rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
            assert c.stats()["wg_waves_per_simd"] == 5
            for b in (0, nblk - 1):
                r_rep, r_int, _ = orc.encode_block(src[b], K, T, esis, want_inter=True)
                assert np.array_equal(inter[b], r_int) and np.array_equal(rep[b], r_rep), (K, T, b)
            st, out, src2 = _roundtrip(G, K, T, nblk, p, 0, seed=K + 11)
            for b in range(nblk):
                assert not st[b] or np.array_equal(out[b], src2[b]), (K, T, b)
    finally:
        c.set_option("small_waves4", 1)
        c.set_option("no_tiny", 0)


def test_segmented_planner_at_small_sizes(G, orc):
    """Big blocks are planned in two kernel parts with helper kernels between and after (W pass on 2-byte strips, HDPC
    fold, W transposition; planner_seq.h).  Forced here for sizes the oracle checks quickly: decoded data against the
    source, verdicts of rank-deficient receptions against the reference algorithm's, device-built encode plans too."""
    from emu_support import decode_setup
    c = G.ctx()
    c.set_option("plan_split_force", 1)
    c.set_option("encplan_dev_min_l", 0)
    try:
        for K, T, nblk, p, oh in [(100, 64, 6, 0.1, 0), (1024, 48, 5, 0.06, 0), (1024, 48, 5, 0.2, 30), (8192, 32, 3, 0.1, 0), (20000, 16, 2, 0.1, 1)]:
            c.clear_plan_cache()
            src = np.stack([payload(K * T, seed=K + 3, block=b).reshape(K, T) for b in range(nblk)])
            esis = np.array([K, K + 7], np.uint32)
            rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
            r_rep, r_int, _ = orc.encode_block(src[0], K, T, esis, want_inter=True)
            assert np.array_equal(inter[0], r_int) and np.array_equal(rep[0], r_rep), K
            st, out, src2 = _roundtrip(G, K, T, nblk, p, oh, seed=K + 11)
            for b in range(nblk):
                assert not st[b] or np.array_equal(out[b], src2[b]), (K, b)
            assert st.sum() >= nblk - 1
        # failure parity
        K, T = 12, 8
        src = payload(K * T, seed=4).reshape(K, T)
        rng = np.random.default_rng(17)
        all_rep, _ = G.gpu_encode(src.reshape(1, K, T), K, T, np.arange(K, K + 60, dtype=np.uint32))
        lost_l, resi_l, reps_l, expect = [], [], [], []
        for trial in range(150):
            nl = int(rng.integers(1, 7))
            lost = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
            resi = (K + rng.choice(60, nl, replace=False)).astype(np.uint32)
            isis, _ = decode_setup(orc, K, lost, resi)
            lost_l.append(lost); resi_l.append(resi); reps_l.append(all_rep[0][resi - K]); expect.append(orc.plan_probe(K, isis)[0] == 1)
        work = np.repeat(src.reshape(1, K, T), 150, axis=0).copy()
        for b in range(150):
            work[b][lost_l[b]] = 0xFF
        st, out, _ = G.gpu_decode(work, K, T, lost_l, resi_l, reps_l)
        assert [bool(x) for x in st] == expect and not all(expect)
    finally:
        c.set_option("plan_split_force", 0)
        c.set_option("encplan_dev_min_l", 12000)
        c.clear_plan_cache()


@pytest.mark.parametrize("K,T,nblk,loss", [(1000, 64, 24, 0.3), (8192, 32, 6, 0.4)])
def test_capacity_fallback_replans_on_the_host(G, orc, K, T, nblk, loss):
    """A block whose plan exceeds a capacity of the device planner (here: inactive columns, lowered through "plan_ucap") is
    reported PL_FAIL_CAPACITY and re-planned by planner_host.cpp inside the same call (nrq_device.hip decode_device).  With
    the capacity set between the smallest and the largest u of the batch, device-planned and host-planned blocks are solved
    by ONE launch: the verdicts and the recovered bytes are the oracle's either way (reference precode.c:287-315: its
    solver has no such capacity)."""
    c = G.ctx()
    P = nanorq_amd.params(K)["P"]
    rng = np.random.default_rng(K)
    src = rng.integers(0, 256, (nblk, K, T), dtype=np.uint8)
    lost = [loss_pattern(K, loss * (0.5 + b / nblk), seed=3, block=b) for b in range(nblk)]   # heavier loss towards the end
    nrep = max(len(x) for x in lost) + 2
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep, _ = G.gpu_encode(src, K, T, esis)
    work = src.copy()
    for b in range(nblk):
        work[b][lost[b]] = 0xEE

    def decode():
        st, out, _ = G.gpu_decode(work, K, T, lost, [esis[:len(l) + 1] for l in lost], [rep[b][:len(lost[b]) + 1] for b in range(nblk)])
        return st, out, c.stats()["host_planned"]

    st0, out0, hp0 = decode()
    assert hp0 == 0
    u = []
    for b in range(nblk):   # the verdict of the reference algorithm, and how many inactive columns our planners end up with
        ok, ref, _ = orc.decode_block(np.concatenate([np.setdiff1d(np.arange(K, dtype=np.uint32), lost[b]), esis[:len(lost[b]) + 1]]),
                                      np.concatenate([src[b][np.setdiff1d(np.arange(K), lost[b])], rep[b][:len(lost[b]) + 1]]), K, T)
        assert bool(st0[b]) == ok, b
        if ok:
            assert np.array_equal(out0[b], ref) and np.array_equal(ref, src[b])
    mixed = False
    try:
        for extra in (40, 56, 72, 88, 104, 136, 168):
            c.set_option("plan_ucap", P + extra)
            st, out, hp = decode()
            assert np.array_equal(st, st0), extra
            assert np.array_equal(out[st0 != 0], out0[st0 != 0]), extra
            mixed = mixed or 0 < hp < nblk
            if extra == 40:
                assert hp > 0          # nothing fits so few inactive columns: every block went to the host planner
    finally:
        c.set_option("plan_ucap", 0)
    assert mixed, "no capacity setting split the batch between the device and the host planner"


def test_planner_instance_guard(G, orc):
    """Each planner kernel instance carries ONE form of the peeling phases (state in LDS | compact state beside arrays in HBM;
    planner_body.h PL_PEEL_DISPATCH3) and launch_plan_kernel picks the instance by pl_ctx_setup's rule.  "plan_wrong_instance"
    gives LDS-sized blocks to the other instance: pl_init_a must report them (PL_FAIL_CAPACITY) instead of peeling with the
    wrong form, and they come back from the host planner decoded like any other block."""
    K, T, nblk = 1024, 32, 8
    c = G.ctx()
    src = np.stack([payload(K * T, seed=70 + b).reshape(K, T) for b in range(nblk)])
    esis = np.arange(K, K + 120, dtype=np.uint32)
    rep, _ = G.gpu_encode(src, K, T, esis)
    lost = [loss_pattern(K, 0.08, seed=11, block=b) for b in range(nblk)]
    work = src.copy()
    for b in range(nblk):
        work[b][lost[b]] = 0x5A
    args = (work, K, T, lost, [esis[:len(l) + 2] for l in lost], [rep[b][:len(lost[b]) + 2] for b in range(nblk)])
    st0, out0, _ = G.gpu_decode(*args)
    assert c.stats()["host_planned"] == 0 and st0.all() and np.array_equal(out0, src)
    ok, ref, _ = orc.decode_block(np.concatenate([np.setdiff1d(np.arange(K, dtype=np.uint32), lost[0]), esis[:len(lost[0]) + 2]]),
                                  np.concatenate([src[0][np.setdiff1d(np.arange(K), lost[0])], rep[0][:len(lost[0]) + 2]]), K, T)
    assert ok and np.array_equal(ref, out0[0])
    try:
        c.set_option("plan_big_wg", 1)          # (1024-thread planner workgroups: the instance pair the guard tells apart)
        c.set_option("plan_wrong_instance", 1)
        st, out, _ = G.gpu_decode(*args)
        assert c.stats()["host_planned"] == nblk, "the guard did not send the blocks to the host planner"
        assert st.all() and np.array_equal(out, src)
    finally:
        c.set_option("plan_wrong_instance", 0)
        c.set_option("plan_big_wg", 0)
    st, out, _ = G.gpu_decode(*args)
    assert c.stats()["host_planned"] == 0 and np.array_equal(out, src)


def test_decode_plan_issued_ahead(G, orc):
    """nrq_decode_plan_ahead: the planner run of a decode call issued before the call (the symbolic stage needs the reception
    pattern only) -- the decode with the same arguments finds it (stats.plan_ahead), a decode with OTHER arguments discards it
    and plans for itself, and a pipeline of several batches with the next batch's plan always in flight (two arena sets in
    turn) decodes every batch like the plain call: bytes equal to the source, which the oracle's decode also returns."""
    K, T, nblk = 600, 64, 6
    c = G.ctx()
    src = np.stack([payload(K * T, seed=50 + b).reshape(K, T) for b in range(nblk)])
    esis = np.arange(K, K + 90, dtype=np.uint32)
    rep, _ = G.gpu_encode(src, K, T, esis)

    def batch(seed):
        lost = np.zeros((nblk, 80), np.uint32); nlost = np.zeros(nblk, np.uint32)
        work = src.copy()
        for b in range(nblk):
            lo = loss_pattern(K, 0.1, seed=seed, block=b)[:80]
            lost[b, :len(lo)] = lo; nlost[b] = len(lo)
            work[b][lo] = 0x77
        return lost, nlost, work

    resi = np.tile(esis, (nblk, 1))
    d_src = c.alloc(src.nbytes); d_rep = c.alloc(rep.nbytes)
    try:
        c.upload(d_rep, rep)
        # 1. same arguments: found
        lost, nlost, work = batch(1)
        c.upload(d_src, work)
        c.decode_plan_ahead(K, T, nblk, d_src, K * T, lost, nlost, resi, nlost + 2, nlost + 5, d_rep, 90 * T)
        st, used = c.decode_blocks_lazy(K, T, nblk, d_src, K * T, lost, nlost, resi, nlost + 2, nlost + 5, d_rep, 90 * T)
        assert c.stats()["plan_ahead"] == 1 and st.all()
        c.sync()
        assert np.array_equal(c.download(d_src, src.nbytes).reshape(src.shape), src)
        ok, ref, _ = orc.decode_block(np.concatenate([np.setdiff1d(np.arange(K, dtype=np.uint32), lost[0, :nlost[0]]), esis[:nlost[0] + 2]]),
                                      np.concatenate([src[0][np.setdiff1d(np.arange(K), lost[0, :nlost[0]])], rep[0][:nlost[0] + 2]]), K, T)
        assert ok and np.array_equal(ref, src[0])
        # 2. other arguments (another reception pattern): discarded, the call plans for itself
        lost2, nlost2, work2 = batch(2)
        c.upload(d_src, work2)
        c.decode_plan_ahead(K, T, nblk, d_src, K * T, lost, nlost, resi, nlost + 2, nlost + 5, d_rep, 90 * T)
        st, used = c.decode_blocks_lazy(K, T, nblk, d_src, K * T, lost2, nlost2, resi, nlost2 + 2, nlost2 + 5, d_rep, 90 * T)
        assert c.stats()["plan_ahead"] == 0 and st.all()
        c.sync()
        assert np.array_equal(c.download(d_src, src.nbytes).reshape(src.shape), src)
        # 3. a pipeline: batch i + 1's plan is issued right after batch i's decode call returned (its solve still queued)
        batches = [batch(10 + i) for i in range(5)]
        bufs = [c.alloc(src.nbytes) for _ in batches]
        for (lo_, nl_, wk_), d in zip(batches, bufs):
            c.upload(d, wk_)
        lo_, nl_, _ = batches[0]
        c.decode_plan_ahead(K, T, nblk, bufs[0], K * T, lo_, nl_, resi, nl_ + 2, nl_ + 5, d_rep, 90 * T)
        for i, ((lo_, nl_, _), d) in enumerate(zip(batches, bufs)):
            st, used = c.decode_blocks_lazy(K, T, nblk, d, K * T, lo_, nl_, resi, nl_ + 2, nl_ + 5, d_rep, 90 * T)
            assert c.stats()["plan_ahead"] == 1 and st.all(), i
            if i + 1 < len(batches):
                ln, nn, _ = batches[i + 1]
                c.decode_plan_ahead(K, T, nblk, bufs[i + 1], K * T, ln, nn, resi, nn + 2, nn + 5, d_rep, 90 * T)
        c.sync()
        for d in bufs:
            assert np.array_equal(c.download(d, src.nbytes).reshape(src.shape), src)
        # 4. the same with TWO runs in flight (they execute side by side: two planner streams, three arena sets); a third one
        # is refused while two are waiting
        for (lo_, nl_, wk_), d in zip(batches, bufs):
            c.upload(d, wk_)

        def ahead(i):
            ln, nn, _ = batches[i]
            c.decode_plan_ahead(K, T, nblk, bufs[i], K * T, ln, nn, resi, nn + 2, nn + 5, d_rep, 90 * T)
        ahead(0); ahead(1)
        with pytest.raises(nanorq_amd.NrqError):
            ahead(2)
        for i, ((lo_, nl_, _), d) in enumerate(zip(batches, bufs)):
            st, used = c.decode_blocks_lazy(K, T, nblk, d, K * T, lo_, nl_, resi, nl_ + 2, nl_ + 5, d_rep, 90 * T)
            assert c.stats()["plan_ahead"] == 1 and st.all(), i
            if i + 2 < len(batches):
                ahead(i + 2)
        c.sync()
        for d in bufs:
            assert np.array_equal(c.download(d, src.nbytes).reshape(src.shape), src)
            c.free(d)
    finally:
        c.free(d_src); c.free(d_rep)


def test_block_lists_per_launch(G, orc):
    """A launch runs at one strip width, and the widest width a block can have is set by ITS plan (inactive columns grow the LDS
    image).  A batch whose blocks do not all fit the widest width is solved as TWO lists (pick_and_launch): the blocks that fit,
    in place at the wide width -- the kernel leaves the others out -- and the others side by side at the widest width they fit.
    Here: K=8192 blocks with loss growing from 10 % to 60 % over the batch, so that some plans outgrow the 16-byte image.  Every
    block must come back bit-exact (and equal to what ONE launch at the narrow width gives: "no_lists"), one block per list is
    compared with the oracle, and the statistics must show both lists.  Reference: lib/precode.c:176-203 (u grows with the
    loss), :15-32 (the replay every width computes)."""
    c = G.ctx()
    K, T, nblk = 8192, 32, 32
    rng = np.random.default_rng(77)
    src = rng.integers(0, 256, (nblk, K, T), dtype=np.uint8)
    lost = [loss_pattern(K, 0.10 + 0.5 * b / nblk, seed=11, block=b) for b in range(nblk)]
    nrep = max(len(x) for x in lost) + 2
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep, _ = G.gpu_encode(src, K, T, esis)
    work = src.copy()
    for b in range(nblk):
        work[b][lost[b]] = 0xEE
    args = (work, K, T, lost, [esis[:len(l) + 2] for l in lost], [rep[b][:len(lost[b]) + 2] for b in range(nblk)])
    st, out, _ = G.gpu_decode(*args)
    s = c.stats()
    assert st.all() and np.array_equal(out, src) and s["strip_bytes"] == 16
    # (images of 153 .. 162 KB here: all fit the 160 KiB; the bound is lowered so that the heavier receptions' do not --
    # what one block in a few thousand does by itself at the real bound)
    thr = 0
    try:
        for kb in range(162, 150, -1):
            c.set_option("lds_max", kb * 1024)
            st, out, _ = G.gpu_decode(*args)
            s = c.stats()
            assert st.all() and np.array_equal(out, src), kb
            if s["blocks_b"]:
                thr = kb * 1024
                break
    finally:
        c.set_option("lds_max", 0)
    assert thr, "no bound split the batch"
    # (the second list runs at the next width down: 12 bytes since round 6)
    assert s["strip_bytes"] == 16 and s["strip_bytes_b"] == 12 and 0 < s["blocks_b"] < nblk, {k: s[k] for k in ("strip_bytes", "strip_bytes_b", "blocks_b", "lds_bytes", "u")}
    try:
        c.set_option("no_lists", 1)
        c.set_option("lds_max", thr)
        st1, out1, _ = G.gpu_decode(*args)
        s1 = c.stats()
    finally:
        c.set_option("no_lists", 0)
        c.set_option("lds_max", 0)
    # (one launch at the width every block fits: 12 bytes -- unless this run's plans, which differ from run to run in who claimed
    # which column, all happen to fit the lowered bound)
    assert s1["strip_bytes"] in (12, 16) and s1["blocks_b"] == 0 and st1.all() and np.array_equal(out1, out)
    try:   # ... and with the 12-byte strip switched off, the second list at 8 bytes as in round 5: same bytes
        c.set_option("no_wb12", 1)
        c.set_option("lds_max", thr)
        st2, out2, _ = G.gpu_decode(*args)
        s2 = c.stats()
    finally:
        c.set_option("no_wb12", 0)
        c.set_option("lds_max", 0)
    assert s2["strip_bytes_b"] in (0, 8) and st2.all() and np.array_equal(out2, out)
    for b in (0, nblk - 1):   # the oracle on the sparsest and the heaviest reception
        keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost[b])
        ok, ref, _ = orc.decode_block(np.concatenate([keep, esis[:len(lost[b]) + 2]]), np.concatenate([src[b][keep], rep[b][:len(lost[b]) + 2]]), K, T)
        assert ok and np.array_equal(out[b], ref)


def test_small_calls_take_the_host_planner(G, orc):
    """A decode call of one or two small blocks is planned on the host (200-460 us of planner kernel latency against ~0.45 us per
    source symbol on the CPU: nrq_decode_blocks_lazy; the reference's harness decodes one block per call) -- unless the option is
    off, the batch is larger, or the blocks are big.  Same bytes either way, and `host_planned` stays 0: it counts blocks the device
    planner gave up on.  Reference call site: benchmark.c (one nanorq_repair_block per block)."""
    c = G.ctx()

    def run(K, T, nblk):
        st, out, src = _roundtrip(G, K, T, nblk, 0.1, 2, seed=K + nblk)
        s = c.stats()
        assert st.all() and np.array_equal(out, src), (K, nblk)
        return s["planner"], s["host_planned"]

    try:
        c.set_option("host_plan_auto", 1)
        assert run(100, 64, 1) == (0, 0)
        assert run(1000, 32, 1) == (0, 0)
        assert run(300, 32, 2) == (0, 0)
        assert run(3000, 16, 1) == (1, 0)     # (a big block: the planner kernel is the faster one)
        assert run(100, 64, 8) == (1, 0)      # (a batch: one workgroup per block, all at once)
        c.set_option("host_plan_auto", 0)
        assert run(100, 64, 1) == (1, 0)
    finally:
        c.set_option("host_plan_auto", 0)


@pytest.mark.parametrize("K,T,nblk", [(300, 64, 5), (1000, 32, 64), (2500, 16, 9)])
def test_planner_workgroup_by_batch_size(G, orc, K, T, nblk):
    """A batch of at most one block per compute unit is planned by 1024-thread workgroups with the whole LDS each (nothing to
    share a CU with; launch_plan_kernel), larger batches of small blocks by 256- or 128-thread ones that share a CU -- "plan_pack"
    forces the latter, as the fixture does.  Same decoded bytes both ways, one block against the oracle."""
    c = G.ctx()
    res = []
    try:
        for pack in (0, 1):
            c.set_option("plan_pack", pack)
            st, out, src = _roundtrip(G, K, T, nblk, 0.1, 2, seed=K + 3)
            assert c.stats()["planner"] == 1 and c.stats()["host_planned"] == 0
            assert st.all() and np.array_equal(out, src)
            res.append(out)
    finally:
        c.set_option("plan_pack", 1)
    assert np.array_equal(res[0], res[1])


def test_rows_by_address_pair(G):
    """nrq_move_rows_dev: row k, T bytes, from address pairs[2k] to address pairs[2k+1], either side device memory or page-locked host
    memory -- the receiver's repaired symbols leave for their places in the caller's output this way (nanorq_repair_all).  Rows of a
    device buffer to scattered places of a page-locked host buffer and back, a zero pair skipped, T a multiple of 16 and not."""
    import ctypes as C
    c = G.ctx()
    L = nanorq_amd.lib()
    L.nrq_move_rows_dev.restype = C.c_int
    L.nrq_move_rows_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]
    L.nrq_host_device_address.restype = C.c_uint64
    L.nrq_host_device_address.argtypes = [C.c_void_p]
    L.nanorq_pinned_alloc.restype = C.c_void_p
    L.nanorq_pinned_alloc.argtypes = [C.c_size_t]
    L.nanorq_pinned_free.argtypes = [C.c_void_p]
    for T in (1280, 52):
        n, slots = 37, 64
        rows = np.frombuffer(payload(n * T, seed=T), np.uint8).reshape(n, T).copy()
        d_rows = c.alloc(n * T)
        c.upload(d_rows, rows)
        hp = L.nanorq_pinned_alloc(slots * T)
        host = np.ctypeslib.as_array((C.c_uint8 * (slots * T)).from_address(hp)).reshape(slots, T)
        host[:] = 0xA5
        hdev = L.nrq_host_device_address(C.c_void_p(hp))
        assert hdev != 0 and L.nrq_host_device_address(C.c_void_p(rows.ctypes.data)) == 0   # (pageable memory: no)
        place = np.random.default_rng(T).permutation(slots)[:n]
        pairs = np.zeros(2 * n, np.uint64)
        pairs[0::2] = d_rows + np.arange(n, dtype=np.uint64) * T
        pairs[1::2] = hdev + place.astype(np.uint64) * T
        pairs[2 * 5] = 0                                    # (skipped)
        d_pairs = c.alloc(pairs.nbytes)
        c.upload(d_pairs, pairs.view(np.uint8))
        assert L.nrq_move_rows_dev(c._h, 0, C.c_void_p(d_pairs), n, T) == 0
        c.sync()
        for k in range(n):
            assert np.array_equal(host[place[k]], rows[k] if k != 5 else np.full(T, 0xA5, np.uint8)), (T, k)
        untouched = np.setdiff1d(np.arange(slots), place)
        assert (host[untouched] == 0xA5).all()
        # and back: host rows to a second device buffer
        d_back = c.alloc(n * T)
        c.memset(d_back, 0, n * T)
        pairs[0::2] = hdev + place.astype(np.uint64) * T
        pairs[1::2] = d_back + np.arange(n, dtype=np.uint64) * T
        c.upload(d_pairs, pairs.view(np.uint8))
        assert L.nrq_move_rows_dev(c._h, 0, C.c_void_p(d_pairs), n, T) == 0
        c.sync()
        back = c.download(d_back, n * T).reshape(n, T)
        rows[5] = 0xA5
        assert np.array_equal(back, rows)
        for d in (d_rows, d_pairs, d_back):
            c.free(d)
        L.nanorq_pinned_free(C.c_void_p(hp))
