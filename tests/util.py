"""Deterministic synthetic inputs shared by tests and bench."""
import numpy as np


def kat_payload(n):
    """byte i = ((uint32) i * 2654435761) >> 24  -- the payload of the SURVEY section 8(c) KATs."""
    i = np.arange(n, dtype=np.uint64)
    return (((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(24)).astype(np.uint8)


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def payload(nbytes, seed=1, block=0):
    """byte i of block b = byte (i%8) of splitmix64(seed ^ b<<40 ^ i/8)."""
    nw = (nbytes + 7) // 8
    with np.errstate(over="ignore"):
        w = splitmix64(np.uint64(seed) ^ (np.uint64(block) << np.uint64(40)) ^ np.arange(nw, dtype=np.uint64))
    return w.view(np.uint8)[:nbytes].copy()


def loss_pattern(K, p, seed, block=0):
    """Bernoulli(p) per source ESI; returns sorted array of LOST ESIs."""
    with np.errstate(over="ignore"):
        w = splitmix64(np.uint64(seed * 7919 + 13) ^ (np.uint64(block) << np.uint64(40)) ^ np.arange(K, dtype=np.uint64))
    u = (w >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return np.nonzero(u < p)[0].astype(np.uint32)


def received_set(K, lost, overhead):
    """ESIs a decoder sees: surviving source ESIs ascending, then len(lost)+overhead repair ESIs from K
    (the shape reference benchmark.c:52-72 produces)."""
    keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
    rep = np.arange(K, K + len(lost) + overhead, dtype=np.uint32)
    return np.concatenate([keep, rep])
