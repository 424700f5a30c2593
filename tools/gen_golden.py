#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/ (data only: inputs by formula, expected outputs as bytes or SHA-256).

  survey_kat.json       the known answers of SURVEY.md section 8(c) ("PROBE KATs": outputs of the reference's own
                        lib/*.c recorded during the survey; the reference ships no vectors), the
                        section 8 parameter table and the reference planner's schedule statistics.  Copied verbatim
                        from SURVEY.md; this script only re-checks that the oracle still reproduces them.
  oracle_vectors.json   further vectors -- encode (K, T, K', payload seed, ESI list -> SHA-256 of repair and of
                        intermediate symbols), decode (loss pattern seed, overhead, arrival order -> verdict + SHA-256
                        of the recovered block), rank-deficient receptions, blocks coded with a larger table row than
                        their own (nanorq.c:289) -- produced by oracle/rq_oracle.c, the C restatement of the reference
                        algorithm.  Provenance: the reference itself cannot be built here (deps/oblas is an empty
                        submodule), so these pin the HIP path and future oracle edits to TODAY's oracle, which is in turn
                        pinned by survey_kat.json.

Inputs are formulas of tests/util.py (kat_payload / payload / loss_pattern), so the files stay small.
    python tools/gen_golden.py          # rewrite both files
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
from util import kat_payload, loss_pattern, payload  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


SURVEY = {
    "provenance": "SURVEY.md section 8(c) PROBE KATs and section 8 tables: reference lib/*.c run at survey time; "
                  "payload byte i = ((uint32)i * 2654435761) >> 24 (tests/util.py kat_payload), one block, Al=8",
    "small": {"K": 10, "T": 8, "symbols": {"10": "1bad540de9c8d3c2", "11": "7db6a373c0213e10", "12": "78e7c05eb6bfe9aa"}},
    "sha256_of_repair": [
        {"K": 100, "T": 1024, "esi_lo": 100, "esi_hi": 110, "sha256": "6a95935839af8cba9d0921efd4a08d7cd103ec95e7318044667162dc5e59f79a"},
        {"K": 1024, "T": 1280, "esi_lo": 1024, "esi_hi": 1076, "sha256": "835df9f9100883e0de8308e15d809ca4593c50da043713486762c0ad2bfe79e3"},
        {"K": 8192, "T": 1280, "esi_lo": 8192, "esi_hi": 8208, "sha256": "35721ecf72a443010134a95fcec8920b550d47b30f41a75ba0f96179b72f99b0"},
    ],
    "oti": [{"K": 10, "T": 8, "common": "0x0000000050000007", "scheme": "0x00000008"},
            {"K": 8192, "T": 1280, "common": "0x0000a000000004ff"}],
    "params": {"100": [101, 562, 17, 10, 113, 128, 15, 17], "1024": [1032, 824, 59, 10, 1051, 1101, 50, 53],
               "8192": [8194, 212, 211, 11, 8273, 8416, 143, 149], "27000": [27111, 21, 503, 13, 27367, 27627, 260, 263],
               "56403": [56403, 471, 907, 16, 56951, 57326, 375, 379]},
    "params_order": ["Kp", "J", "S", "H", "W", "L", "P", "P1"],
    "schedule_encode": [{"K": 100, "i": 102, "u": 26, "recorded": 2257, "n1": 1744, "nB": 1244, "n0": 9},
                        {"K": 1024, "i": 1031, "u": 70, "recorded": 22473, "n1": 21681, "nB": 10926, "n0": 10},
                        {"K": 8192, "i": 8209, "u": 207, "recorded": 186913, "n1": 189238, "nB": 91868, "n0": 11}],
}

# (K, Kp, T, payload seed, repair ESIs)
ENCODE = [
    (1, 0, 8, 1, [1, 2, 3]), (2, 0, 4, 2, [2, 9]), (10, 0, 40, 3, [10, 11, 25]), (10, 18, 16, 3, [10, 11, 12]),
    (55, 0, 4, 4, [55, 56, 1000]), (100, 0, 1024, 5, list(range(100, 106))), (95, 101, 64, 5, [95, 96, 300]),
    (101, 0, 20, 6, [101, 102, (1 << 24) - 1]), (500, 0, 72, 7, [500, 777, 100000]), (1024, 0, 1280, 8, list(range(1024, 1030))),
    (1000, 1032, 24, 8, [1000, 1001, 2000]), (1033, 0, 16, 9, [1033, 1034]), (4000, 0, 48, 10, [4000, 4001, 9999]),
    (8192, 0, 32, 11, list(range(8192, 8200))), (8100, 8194, 16, 11, [8100, 8101]), (20000, 0, 8, 12, [20000, 20001]),
]
# (K, Kp, T, payload seed, loss probability, loss seed, overhead, arrival order of the received symbols)
DECODE = [
    (10, 0, 16, 21, 0.3, 5, 0, "sorted"), (10, 18, 16, 21, 0.3, 5, 1, "sorted"), (100, 0, 1024, 22, 0.06, 6, 0, "sorted"),
    (100, 0, 64, 22, 0.06, 6, 3, "shuffled"), (100, 0, 8, 23, 0.5, 7, 40, "sorted"), (95, 101, 32, 23, 0.1, 7, 0, "repair_first"),
    (1024, 0, 1280, 24, 0.05, 8, 0, "sorted"), (1024, 0, 64, 24, 0.06, 8, 52, "shuffled"), (1000, 1032, 24, 25, 0.3, 9, 1, "sorted"),
    (4000, 0, 16, 26, 0.2, 10, 0, "sorted"), (8192, 0, 32, 27, 0.1, 11, 0, "sorted"), (8192, 0, 32, 27, 0.1, 11, 2, "shuffled"),
    (8192, 0, 32, 27, 0.1, 11, 11, "sorted"), (8100, 8194, 16, 28, 0.1, 12, 0, "sorted"), (20000, 0, 8, 29, 0.1, 13, 0, "sorted"),
]


def received(K, lost, overhead, order, seed):
    """ESIs in arrival order: surviving source symbols and len(lost)+overhead repair symbols from ESI K."""
    keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
    rep = np.arange(K, K + len(lost) + overhead, dtype=np.uint32)
    if order == "sorted":
        return np.concatenate([keep, rep])
    if order == "repair_first":
        return np.concatenate([rep[::-1], keep])
    allr = np.concatenate([keep, rep])
    return allr[np.random.default_rng(seed).permutation(len(allr))]


def encode_case(K, Kp, T, seed, esis):
    src = payload(K * T, seed=seed).reshape(K, T)
    rep, inter, st = oracle.encode_block(src, K, T, esis, want_inter=True, Kp=Kp)
    return {"K": K, "Kp": Kp, "T": T, "payload_seed": seed, "esis": [int(e) for e in esis], "sha256_repair": sha(rep),
            "sha256_intermediate": sha(inter), "first_repair_hex": rep[0][:16].tobytes().hex(), "i": st["i"], "u": st["u"]}


def decode_case(K, Kp, T, seed, p, lseed, oh, order):
    src = payload(K * T, seed=seed).reshape(K, T)
    lost = loss_pattern(K, p, seed=lseed)
    esis = received(K, lost, oh, order, lseed)
    nrep = len(lost) + oh
    rep, _, _ = oracle.encode_block(src, K, T, np.arange(K, K + nrep, dtype=np.uint32), Kp=Kp)
    syms = np.stack([src[e] if e < K else rep[e - K] for e in esis]) if len(esis) else np.zeros((0, T), np.uint8)
    ok, out, st = oracle.decode_block(esis, syms, K, T, Kp=Kp)
    return {"K": K, "Kp": Kp, "T": T, "payload_seed": seed, "loss": p, "loss_seed": lseed, "overhead": oh, "order": order,
            "n_lost": int(len(lost)), "decodable": bool(ok), "sha256_recovered": sha(out) if ok else None,
            "equals_source": bool(ok and np.array_equal(out, src)), "i": st["i"], "u": st["u"]}


def failure_sweep():
    """K=12: random small receptions, about one in six rank deficient -- the verdict must match the reference algorithm's."""
    K, T = 12, 8
    rng = np.random.default_rng(7)
    out = []
    for _ in range(120):
        nl = int(rng.integers(1, 7))
        lost = np.sort(rng.choice(K, nl, replace=False)).astype(np.uint32)
        resi = (K + rng.choice(60, nl, replace=False)).astype(np.uint32)
        p = oracle.params(K)
        isis = list(range(p["Kp"]))
        for g, e in enumerate(lost):
            isis[int(e)] = int(resi[g]) + p["Kp"] - K
        r, _ = oracle.plan_probe(K, np.array(isis, np.uint32))
        out.append({"lost": [int(x) for x in lost], "repair_esis": [int(x) for x in resi], "decodable": r == 1})
    return {"K": K, "T": T, "cases": out}


def check_survey():
    rep, _, _ = oracle.encode_block(kat_payload(80), 10, 8, [10, 11, 12])
    assert {str(10 + k): rep[k].tobytes().hex() for k in range(3)} == SURVEY["small"]["symbols"]
    for c in SURVEY["sha256_of_repair"]:
        rep, _, _ = oracle.encode_block(kat_payload(c["K"] * c["T"]), c["K"], c["T"], list(range(c["esi_lo"], c["esi_hi"])))
        assert sha(rep) == c["sha256"], c


def main():
    os.makedirs(GOLD, exist_ok=True)
    check_survey()
    with open(os.path.join(GOLD, "survey_kat.json"), "w") as f:
        json.dump(SURVEY, f, indent=1)
        f.write("\n")
    vec = {"provenance": "oracle/rq_oracle.c (C restatement of reference lib/precode.c + lib/nanorq.c), generated by "
                         "tools/gen_golden.py; payload(nbytes, seed) and loss_pattern(K, p, seed) are tests/util.py's; "
                         "repair symbols received are ESI K..K+n_lost+overhead-1; order: sorted = surviving source "
                         "symbols ascending then repair ascending, repair_first = repair descending then source, "
                         "shuffled = numpy default_rng(loss_seed).permutation of the sorted list",
           "encode": [encode_case(*c) for c in ENCODE], "decode": [decode_case(*c) for c in DECODE],
           "failure_sweep": failure_sweep()}
    assert any(not c["decodable"] for c in vec["failure_sweep"]["cases"])
    with open(os.path.join(GOLD, "oracle_vectors.json"), "w") as f:
        json.dump(vec, f, indent=1)
        f.write("\n")
    print("wrote", len(vec["encode"]), "encode,", len(vec["decode"]), "decode,", len(vec["failure_sweep"]["cases"]), "sweep cases")


if __name__ == "__main__":
    main()
