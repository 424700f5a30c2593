"""Runs ON THE GPU BOX: decode batches with fresh loss patterns over and over; reports blocks that fail or come out wrong.
   python tools/stress_decode.py K T blocks loss iterations"""
import os
import sys

import numpy as np
import torch

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import nanorq_amd  # noqa: E402
from util import loss_pattern  # noqa: E402

K, T, NB, loss, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.empty(1, device=dev)
ctx = nanorq_amd.Context(0, torch.cuda.current_stream(dev).cuda_stream)
L = nanorq_amd.params(K)["L"]
src = torch.randint(0, 256, (NB, K, T), dtype=torch.uint8, device=dev)
bad = 0
for it in range(iters):
    lost = [loss_pattern(K, loss, seed=int(os.environ.get('STRESS_SEED', 5000 + it)), block=int(os.environ.get('STRESS_BLOCK', b))) for b in range(NB)]
    ml = max(len(x) for x in lost)
    nrep = ml + 3
    esis = np.arange(K, K + nrep, dtype=np.uint32)
    rep = torch.empty((NB, nrep, T), dtype=torch.uint8, device=dev)
    if it == 0 or nrep != last_nrep:
        ctx.encode_blocks(K, T, NB, src.data_ptr(), K * T, rep.data_ptr(), nrep * T, esis, 0, 0)
        last_nrep, last_rep = nrep, rep
    else:
        rep = last_rep
    work = src.clone()
    la = np.zeros((NB, ml + 1), np.uint32)
    for b in range(NB):
        la[b, :len(lost[b])] = lost[b]
        work[b, torch.from_numpy(lost[b].astype(np.int64)).to(dev)] = 0xEE
    nl = np.array([len(x) for x in lost], np.uint32)
    resi = np.tile(esis, (NB, 1))
    st, used = ctx.decode_blocks_lazy(K, T, NB, work.data_ptr(), K * T, la, nl, resi, nl, nl + 3, rep.data_ptr(), nrep * T)
    torch.cuda.synchronize()
    st = np.asarray(st)
    wrong = (~(work == src).flatten(1).all(1)).cpu().numpy()
    hp = ctx.stats().get("host_planned", 0)
    if hp:  # (a capacity the device planner ran out of, or its plan check: planner_body.h pl_check_*)
        print("iteration %d: host planner took %d blocks" % (it, hp), flush=True)
    for b in np.nonzero((st == 0) | wrong)[0]:
        bad += 1
        print("iteration %d block %d: status %d, data %s, lost %d, planner %s" % (it, b, st[b], "WRONG" if wrong[b] else "ok", nl[b], ctx.stats().get("host_planned")), flush=True)
print("done: %d iterations x %d blocks, %d bad" % (iters, NB, bad))
