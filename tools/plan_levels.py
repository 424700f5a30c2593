"""Print the shape of a plan's op stream: rows, ops per row, level widths.  Usage: plan_levels.py K [loss]"""
import sys

import numpy as np

from nanorq_amd import binding as b

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
loss = float(sys.argv[2]) if len(sys.argv) > 2 else 0
Kp = b.params(K)["Kp"]
kc = b.host_kconst(K)
isis = np.arange(Kp, dtype=np.uint32)
if loss > 0:
    rng = np.random.default_rng(1)
    lost = np.nonzero(rng.random(K) < loss)[0]
    isis = np.concatenate([np.setdiff1d(np.arange(Kp), lost), Kp + np.arange(len(lost))]).astype(np.uint32)
plan = b.host_plan(K, isis, kc)
h = b.plan_header(plan)
print({k: h[k] for k in "K Kp S H W L P M npiv u nlow r2 nfree nlev nrows pipe wpr n_xor_ops total_bytes".split()})
ops = b.plan_ops(plan, h)
real = ((ops & 0xFFFF) >= 64).sum(1)
print("rows", len(real), "empty rows", int((real == 0).sum()), "ops", int(real.sum()), "mean fill of non-empty rows %.1f" % real[real > 0].mean())
# level groups = runs of non-empty rows
lev, cur = [], 0
for n in real:
    if n:
        cur += n
    elif cur:
        lev.append(cur)
        cur = 0
lev = np.array(lev)
print("groups", len(lev), " <=64:", int((lev <= 64).sum()), " <=128:", int((lev <= 128).sum()), " <=256:", int((lev <= 256).sum()), "max", int(lev.max()))

# LDS cost model of the forward rows (MI355X_MICROARCH.md, LDS): a 64-lane ds_xor_b64 is served in four groups of 16
# contiguous lanes, bank = dword address mod 32 -> with 16-byte slots the class of a target is slot mod 8; ds_read_b64 in two
# groups of 32 lanes, bank = dword address mod 64 -> class of a source is slot mod 16.  A group takes as many cycles as the
# busiest class has different addresses.
dst = (ops & 0xFFFF).astype(np.int64)
srcs = (ops >> 16).astype(np.int64)
def cost(slots, lanes, classes):
    tot = 0
    rows = slots.reshape(len(slots), 64 // lanes, lanes)
    out = np.zeros(len(slots))
    for r in range(len(slots)):
        for g in range(64 // lanes):
            s = np.unique(rows[r, g])
            out[r] += np.bincount(s % classes, minlength=classes).max()
    return out
cd, cs = cost(dst, 16, 8), cost(srcs, 32, 16)
live = real > 0
print("model cycles per row: atomics %.2f (floor 8 at 2 per class)  reads %.2f (floor 4)   rows %d" % (cd[live].mean(), cs[live].mean(), int(live.sum())))
