"""Runs ON THE GPU BOX under rocprofv3 (tools/phase_counters.sh): `reps` encode + decode launches of the headline batch (K=8192,
T=1280, 256 blocks, 10 % loss) with NO verification -- for the phase-truncated builds of the solve kernel (-DNRQ_STOP_AFTER=p),
whose results are garbage by design.   python tools/phase_driver.py [K T blocks loss reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import nanorq_amd  # noqa: E402
from util import loss_pattern  # noqa: E402

K, T, NB, loss, reps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (8192, 1280, 256, 0.1, 4)
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
la = np.zeros((NB, ml + 1), np.uint32)
for b in range(NB):
    la[b, :len(lost[b])] = lost[b]
nl = np.array([len(x) for x in lost], np.uint32)
work = src.clone()
for it in range(reps):
    ctx.encode_blocks(K, T, NB, src.data_ptr(), K * T, rep.data_ptr(), nrep * T, esis, inter.data_ptr(), L * T)
    ctx.decode_blocks_lazy(K, T, NB, work.data_ptr(), K * T, la, nl, np.tile(esis, (NB, 1)), nl, nl + 3, rep.data_ptr(), nrep * T)
    torch.cuda.synchronize()
print("done", reps)
