import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, nanorq_amd
import gpu_support as G
from util import loss_pattern
c = G.ctx()
for K, loss in ((56403, 0.2), (56403, 0.45), (56403, 0.6), (20000, 0.45), (30000, 0.5)):
    T, nblk = 16, 2
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (nblk, K, T), dtype=np.uint8)
    lost = [loss_pattern(K, loss, seed=3, block=b) for b in range(nblk)]
    nrep = max(len(x) for x in lost) + 2
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep, _ = G.gpu_encode(src, K, T, esis)
    work = src.copy()
    for b in range(nblk):
        work[b][lost[b]] = 0
    st, out, _ = G.gpu_decode(work, K, T, lost, [esis[:len(l) + 2] for l in lost], [rep[b][:len(lost[b]) + 2] for b in range(nblk)])
    s = c.stats()
    print(K, loss, "status", st, "ok", [bool(np.array_equal(out[b], src[b])) for b in range(nblk)], "host_planned", s["host_planned"], "u", s["u"], "npiv", s["npiv"], "nlev", s["nlev"])
