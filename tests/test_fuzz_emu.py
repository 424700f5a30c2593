"""Randomised CPU-tier sweep: small random (K, T, loss, overhead, strip width) cases through the emulated device
planner + emulated solve workgroup against the oracle -- odd symbol sizes, K = 1, K at table-row boundaries, heavy
loss, large overhead, every strip width."""
import numpy as np
import pytest

import nanorq_amd
from emu_support import decode_setup, emu_device_plan, emu_solve, lt_lists
from util import payload


@pytest.mark.parametrize("seed", range(6))
def test_random_small_cases(orc, seed):
    rng = np.random.default_rng(100 + seed)
    done = 0
    for trial in range(40):
        K = int(rng.choice([1, 2, 9, 10, 11, 12, 13, 26, 27, 55, 56, 101, 102, 150, 257]))
        T = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 17, 24, 33, 40]))
        wb = int(rng.choice([2, 4, 8, 12, 16]))
        nl = int(rng.integers(1, K + 1)) if K > 1 else 1
        nl = min(nl, max(1, int(K * rng.choice([0.1, 0.3, 0.6, 1.0]))))
        prm = orc.params(K)
        room = 2 * prm["Kp"] - K + 1          # the reference decoder only accepts ESIs up to 2K' (nanorq.c:374, :485)
        nl = min(nl, room)
        oh = min(int(rng.choice([0, 0, 1, 2, 5, 20])), room - nl)
        kc = nanorq_amd.host_kconst(K)
        src = payload(K * T, seed=seed * 1000 + trial).reshape(K, T)
        lost = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
        rep_esis = (K + rng.choice(room, nl + oh, replace=False)).astype(np.uint32)
        rep, _, _ = orc.encode_block(src, K, T, rep_esis)
        keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
        ok, ref_out, _ = orc.decode_block(np.concatenate([keep, rep_esis]), np.concatenate([src[keep], rep]) if len(keep) else rep,
                                          K, T)
        plan, hdr = emu_device_plan(K, kc, lost, rep_esis, lds_bytes=int(rng.choice([24, 140])) * 1024)
        host_plan = nanorq_amd.host_plan(K, decode_setup(orc, K, lost, rep_esis)[0], kc)
        assert (hdr["status"] == 0) == ok == (nanorq_amd.plan_header(host_plan)["status"] == 0), (K, T, lost, rep_esis)
        if not ok:
            continue
        _, rowsrc = decode_setup(orc, K, lost, rep_esis)
        for pl in (plan, host_plan):
            work = src.copy()
            work[lost] = 0xA5
            r, _ = emu_solve(pl, kc, rowsrc, work, rep, T, prm["L"], lt_lists(orc, K, lost, pl), lost, work, wb)
            assert r == 1 and np.array_equal(work, src), (K, T, wb, lost, rep_esis)
        done += 1
    assert done > 10
