"""-m gpu tier: seeded random shapes through the C ABI -- block size, symbol size (incl. sizes whose last 128-byte line
group is partial and single bytes), block count, loss rate and overhead drawn at random; every decodable block must come
back bit-exact, undecodable ones untouched, and one block per case is compared with the CPU oracle byte for byte
(repair symbols, intermediate symbols, decode verdict)."""
import numpy as np
import pytest

from util import loss_pattern, payload, received_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_support
    gpu_support.ctx()
    return gpu_support


@pytest.mark.parametrize("seed", range(8))
def test_random_shapes(G, orc, seed):
    rng = np.random.default_rng(4000 + seed)
    for trial in range(30):
        K = int(rng.choice([1, 2, 7, 10, 11, 26, 55, 100, 101, 257, 400, 777, 1024, 1500, 2049, 2600, 3100]))
        T = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 40, 64, 100, 128, 129, 200, 272]))
        nblk = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 33]))
        if K * T * nblk > 24 << 20:
            nblk = max(1, (24 << 20) // (K * T))
        p = float(rng.choice([0.02, 0.1, 0.3, 0.6]))
        oh = int(rng.choice([0, 0, 1, 2, 5]))
        src = np.stack([payload(K * T, seed=seed * 100 + trial, block=b).reshape(K, T) for b in range(nblk)])
        lost = [loss_pattern(K, p, seed=seed * 977 + trial, block=b) for b in range(nblk)]
        nrep = max(len(l) for l in lost) + oh
        esis = np.arange(K, K + nrep, dtype=np.uint32)
        rep, inter = G.gpu_encode(src, K, T, esis, want_inter=True)
        b0 = int(rng.integers(nblk))
        r_rep, r_int, _ = orc.encode_block(src[b0], K, T, esis, want_inter=True)
        assert np.array_equal(rep[b0], r_rep) and np.array_equal(inter[b0], r_int), (K, T, nblk, "encode")
        work = src.copy()
        for b in range(nblk):
            work[b][lost[b]] = 0x77
        use = [len(l) + (oh if len(l) else 0) for l in lost]
        st, out, _ = G.gpu_decode(work, K, T, lost, [esis[:n] for n in use], [rep[b][:use[b]] for b in range(nblk)])
        for b in range(nblk):
            if st[b]:
                assert np.array_equal(out[b], src[b]), (K, T, nblk, p, oh, b)
            else:
                assert np.array_equal(out[b], work[b]), (K, T, nblk, p, oh, b, "undecodable block touched")
        # the verdict of one block against the reference algorithm
        rx = received_set(K, lost[b0], oh if len(lost[b0]) else 0)
        syms = np.concatenate([src[b0][rx[rx < K]], rep[b0][:use[b0]]]) if use[b0] else src[b0][rx[rx < K]]
        ok, _, _ = orc.decode_block(rx, syms, K, T)
        assert bool(st[b0]) == ok, (K, T, nblk, p, oh, b0, "verdict")
