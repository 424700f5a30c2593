"""Runs ON THE GPU BOX: wall time of encode / decode calls with a device synchronisation around each (what a caller sees),
next to the HIP-event duration of the solve kernel.   python tools/time_phases.py K T blocks loss"""
import os
import sys
import time

import numpy as np
import torch

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import nanorq_amd  # noqa: E402
from util import loss_pattern  # noqa: E402

K, T, NB, loss = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.empty(1, device=dev)
ctx = nanorq_amd.Context(0, torch.cuda.current_stream(dev).cuda_stream)
L = nanorq_amd.params(K)["L"]
src = torch.randint(0, 256, (NB, K, T), dtype=torch.uint8, device=dev)
lost = [loss_pattern(K, loss, seed=1000, block=b) for b in range(NB)]
ml = max(len(x) for x in lost)
nrep = ml + 3
esis = np.arange(K, K + nrep, dtype=np.uint32)
rep = torch.empty((NB, nrep, T), dtype=torch.uint8, device=dev)
inter = torch.empty((NB, L, T), dtype=torch.uint8, device=dev)
work = src.clone()
la = np.zeros((NB, ml + 1), np.uint32)
for b in range(NB):
    la[b, :len(lost[b])] = lost[b]
nl = np.array([len(x) for x in lost], np.uint32)
resi = np.tile(esis, (NB, 1))
ctx.ktime_enable(True)
te, td, tcall = [], [], []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.encode_blocks(K, T, NB, src.data_ptr(), K * T, rep.data_ptr(), nrep * T, esis, inter.data_ptr(), L * T)
    tc = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if it >= 2:
        tcall.append((tc - t0) * 1e3)
    st, used = ctx.decode_blocks_lazy(K, T, NB, work.data_ptr(), K * T, la, nl, resi, nl, nl + 3, rep.data_ptr(), nrep * T)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it >= 2:
        te.append((t1 - t0) * 1e3); td.append((t2 - t1) * 1e3)
kt = ctx.ktime_read()
print("host part of the encode call %.2f ms" % np.mean(tcall))
print("K=%d T=%d blocks=%d: encode call %.2f ms, decode call %.2f ms (wall, synchronised) | solve kernel by HIP events: enc %.2f dec %.2f ms | stats %s"
      % (K, T, NB, np.mean(te), np.mean(td), np.mean(kt[4::2]), np.mean(kt[5::2]), {k: ctx.stats()[k] for k in ("grid", "wg_threads", "lds_bytes", "strips_per_slot")}))
