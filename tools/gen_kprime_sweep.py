#!/usr/bin/env python3
"""Writes tests/golden/kprime_sweep.json: one encode per row of RFC 6330 Table 2 (all 477 K' values).

Why: the reference itself cannot be built here (deps/oblas is an empty submodule), so bit-level parity rests on the oracle's
restatement of Rand / Deg / Tuple / LDPC / HDPC and the systematic index J(K').  RFC 6330 chose J(K') so that the constraint
matrix of K' source symbols is NONSINGULAR with exactly that generator; a mis-restated generator survives one K' with
p ~ 0.99 (a random matrix of this family over GF(2)/GF(256) is singular ~1 % of the time) and all 477 with p < 1 %.  So for
every K' the oracle must (a) find the matrix nonsingular, (b) reproduce the source symbols from the intermediate symbols
(systematic property, RFC 6330 section 5.3.3.4.2) -- both checked here and again by tests/test_golden.py -- and the first
four repair symbols (ESI K' .. K'+3, T = 8) are committed by SHA-256 so that the HIP encoder (tests/test_gpu_golden.py) and
later oracle edits are held to them.  This narrows what "parity unpinned" can hide; it does not lift the label: the
expected bytes are still the oracle's own.

    python tools/gen_kprime_sweep.py        # ~20 s on one core
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
from util import payload  # noqa: E402

T, NREP, SEED = 8, 4, 477


def systematic_sample(Kp, src, inter, n=48):
    """LT(intermediate) == source for the first, the last and n pseudo-random ESIs (oracle's own tuple code)."""
    rng = np.random.default_rng(Kp)
    for esi in sorted(set([0, Kp - 1] + [int(x) for x in rng.integers(0, Kp, n)])):
        acc = np.zeros(T, np.uint8)
        for c in oracle.lt_columns(Kp, esi):
            acc ^= inter[c]
        if not np.array_equal(acc, src[esi]):
            return False
    return True


def main():
    with open(os.path.join(ROOT, "tests", "golden", "rfc6330_tables.json")) as f:
        rows = json.load(f)["table2_rows"]
    out = []
    for Kp, J, S, H, W in rows:
        p = oracle.params(Kp)
        assert (p["Kp"], p["J"], p["S"], p["H"], p["W"]) == (Kp, J, S, H, W), Kp
        src = payload(Kp * T, seed=SEED, block=Kp).reshape(Kp, T)
        rep, inter, st = oracle.encode_block(src, Kp, T, np.arange(Kp, Kp + NREP, dtype=np.uint32), want_inter=True)
        assert rep is not None and systematic_sample(Kp, src, inter), "K'=%d: singular or not systematic" % Kp
        out.append({"Kp": Kp, "J": J, "i": int(st["i"]), "u": int(st["u"]), "sha256_repair": hashlib.sha256(rep.tobytes()).hexdigest(),
                    "sha256_intermediate": hashlib.sha256(inter.tobytes()).hexdigest()})
    doc = {"provenance": "oracle/rq_oracle.c (C restatement of the reference algorithm; the reference is unbuildable here: deps/oblas "
                         "absent).  One encode per row of RFC 6330 Table 2 (reference include/table2.h:6-211, lib/params.c:21-45, "
                         "lib/tuple.c:21-43): K = K', T = 8, payload = tests/util.py payload(K'*8, seed=477, block=K'); repair ESIs "
                         "K'..K'+3.  Every row was found nonsingular and systematic by tools/gen_kprime_sweep.py.",
           "T": T, "payload_seed": SEED, "repair_per_row": NREP, "rows": out}
    with open(os.path.join(ROOT, "tests", "golden", "kprime_sweep.json"), "w") as f:   # one table row per line
        head = {k: v for k, v in doc.items() if k != "rows"}
        f.write(json.dumps(head, separators=(",", ":"))[:-1] + ',"rows":[\n')
        f.write(",\n".join(json.dumps(r, separators=(",", ":")) for r in out))
        f.write("\n]}\n")
    print("wrote %d rows" % len(out))


if __name__ == "__main__":
    main()
