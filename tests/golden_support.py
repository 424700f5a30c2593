"""Readers of the committed fixtures under tests/golden/ (written by tools/gen_golden.py) and the input formulas
they refer to."""
import hashlib
import json
import os

import numpy as np

from util import loss_pattern, payload

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def arrival_order(K, lost, overhead, order, seed):
    keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
    rep = np.arange(K, K + len(lost) + overhead, dtype=np.uint32)
    if order == "sorted":
        return np.concatenate([keep, rep])
    if order == "repair_first":
        return np.concatenate([rep[::-1], keep])
    allr = np.concatenate([keep, rep])
    return allr[np.random.default_rng(seed).permutation(len(allr))]


def decode_inputs(c):
    """(src[K,T], lost ESIs ascending, received ESIs in arrival order) of a decode fixture."""
    K, T = c["K"], c["T"]
    src = payload(K * T, seed=c["payload_seed"]).reshape(K, T)
    lost = loss_pattern(K, c["loss"], seed=c["loss_seed"])
    assert len(lost) == c["n_lost"]
    return src, lost, arrival_order(K, lost, c["overhead"], c["order"], c["loss_seed"])
