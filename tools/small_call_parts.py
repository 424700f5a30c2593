import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, nanorq_amd
import gpu_support as G
from util import loss_pattern, payload
c = G.ctx()
c.set_option("host_plan_auto", 1)
for K in (100, 500, 1000):
    T, nblk = 1280, 1
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
    tc, ts, pm, hm = [], [], [], []
    for it in range(40):
        c.upload(d_src, work); c.sync()
        t0 = time.perf_counter()
        st, used = c.decode_blocks_lazy(K, T, nblk, d_src, K * T, la, nl, np.tile(esis, (nblk, 1)), nl, nl + 2, d_rep, nrep * T)
        t1 = time.perf_counter()
        c.sync()
        t2 = time.perf_counter()
        s = c.stats()
        tc.append((t1 - t0) * 1e6); ts.append((t2 - t1) * 1e6); pm.append(s["plan_ms"] * 1e3); hm.append(s["host_ms"] * 1e3)
    med = lambda x: float(np.median(x[5:]))
    print("K=%d: call returns after %.0f us (host planner %.0f, decode_host in all %.0f), sync takes %.0f more; planner=%d" % (K, med(tc), med(pm), med(hm), med(ts), s["planner"]))
