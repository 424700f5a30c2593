import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, nanorq_amd
import gpu_support as G
from util import loss_pattern, payload
c = G.ctx()
print("K nblk  device_us  host_us   (decode call, median of 30, python binding incl.)")
for K in (100, 300, 600, 1000):
    for nblk in (1, 2, 3, 4, 8):
        T = 1280
        src = np.stack([payload(K * T, seed=1, block=b).reshape(K, T) for b in range(nblk)])
        lost = [loss_pattern(K, 0.06, seed=3, block=b) for b in range(nblk)]
        nrep = max(len(l) for l in lost) + 2
        esis = np.arange(K, K + nrep, dtype=np.uint32)
        rep, _ = G.gpu_encode(src, K, T, esis)
        work = src.copy()
        for b in range(nblk): work[b][lost[b]] = 0
        d_src = c.alloc(work.nbytes); d_rep = c.alloc(rep.nbytes)
        c.upload(d_rep, rep)
        ml = max(len(l) for l in lost)
        la = np.zeros((nblk, ml + 1), np.uint32); nl = np.array([len(l) for l in lost], np.uint32)
        for b in range(nblk): la[b, :len(lost[b])] = lost[b]
        res = {}
        for mode in (0, 1):
            c.set_option("host_plan_auto", 0)
            c.set_planner(mode == 0)
            ts = []
            for it in range(34):
                c.upload(d_src, work); c.sync()
                t0 = time.perf_counter()
                st, used = c.decode_blocks_lazy(K, T, nblk, d_src, K * T, la, nl, np.tile(esis, (nblk, 1)), nl, nl + 2, d_rep, nrep * T)
                c.sync()
                ts.append((time.perf_counter() - t0) * 1e6)
            res[mode] = float(np.median(ts[4:]))
            assert np.asarray(st).all()
        c.set_planner(True)
        c.free(d_src); c.free(d_rep)
        print("%5d %3d  %8.0f  %8.0f   rule: %s" % (K, nblk, res[0], res[1], "host" if 38 * nblk * K < 20000 + 28 * K else "device"))
