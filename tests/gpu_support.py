"""Helpers for the -m gpu tier: run the HIP path through the C ABI with numpy in/out."""
import numpy as np

import nanorq_amd

_CTX = None


def ctx():
    global _CTX
    if _CTX is None:
        # Tests that keep blocks in HBM use torch tensors.  torch brings its own copy of the HIP runtime, and it only
        # finds the GPU if it initialises BEFORE the runtime libnanorq_hip.so is linked against opens the device
        # (bench.py has the same order): bring torch up first.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
                torch.empty(1, device="cuda")
        except ImportError:
            pass
        _CTX = nanorq_amd.Context(0)
        # the parity tests are about the DEVICE planner at every size and block count; a product context gives calls of one or two small
        # blocks to the host planner (nrq_decode_blocks_lazy "host_small"), which test_small_calls_take_the_host_planner covers
        _CTX.set_option("host_plan_auto", 0)
        # ... and about every FORM of it: a product context gives a batch of at most one block per compute unit the 1024-thread planner
        # workgroup; the 256- and 128-thread forms that batches of thousands of small blocks use are exercised here at a few blocks
        _CTX.set_option("plan_pack", 1)
        # ... and the single-wave solve workgroups, which a product context keeps for launches of more than ~500 strips
        _CTX.set_option("tiny_any", 1)
    return _CTX


def gpu_encode(src_blocks, K, T, esis, want_inter=False, Kp=0):
    """src_blocks: [nblk, K, T] uint8 -> (repair [nblk, nrep, T], inter [nblk, L, T] or None)"""
    c = ctx()
    src_blocks = np.ascontiguousarray(src_blocks, np.uint8)
    nblk = src_blocks.shape[0]
    esis = np.ascontiguousarray(esis, np.uint32)
    nrep = len(esis)
    L = nanorq_amd.params(Kp or K)["L"]
    d_src = c.alloc(nblk * K * T)
    d_rep = c.alloc(max(1, nblk * nrep * T))
    d_int = c.alloc(nblk * L * T) if want_inter else 0
    try:
        c.upload(d_src, src_blocks)
        c.memset(d_rep, 0xCD, max(1, nblk * nrep * T))
        c.encode_blocks(K, T, nblk, d_src, K * T, d_rep, nrep * T, esis, d_int, L * T, Kp=Kp)
        c.sync()
        rep = c.download(d_rep, nblk * nrep * T).reshape(nblk, nrep, T) if nrep else np.zeros((nblk, 0, T), np.uint8)
        inter = c.download(d_int, nblk * L * T).reshape(nblk, L, T) if want_inter else None
    finally:
        c.free(d_src); c.free(d_rep)
        if d_int:
            c.free(d_int)
    return rep, inter


def gpu_decode(work_blocks, K, T, lost_lists, rep_esi_lists, rep_syms_lists, want_inter=False, Kp=0):
    """work_blocks: [nblk, K, T] with received source symbols in place (missing rows arbitrary).
    Returns (status[nblk], recovered blocks [nblk,K,T], inter or None)."""
    c = ctx()
    work_blocks = np.ascontiguousarray(work_blocks, np.uint8)
    nblk = work_blocks.shape[0]
    lost_cap = max(1, max(len(x) for x in lost_lists))
    rep_cap = max(1, max(len(x) for x in rep_esi_lists))
    lost = np.zeros((nblk, lost_cap), np.uint32)
    resi = np.zeros((nblk, rep_cap), np.uint32)
    reps = np.zeros((nblk, rep_cap, T), np.uint8)
    for b in range(nblk):
        lost[b, :len(lost_lists[b])] = lost_lists[b]
        resi[b, :len(rep_esi_lists[b])] = rep_esi_lists[b]
        if len(rep_esi_lists[b]):
            reps[b, :len(rep_esi_lists[b])] = rep_syms_lists[b]
    nlost = np.array([len(x) for x in lost_lists], np.uint32)
    nrep = np.array([len(x) for x in rep_esi_lists], np.uint32)
    L = nanorq_amd.params(Kp or K)["L"]
    d_src = c.alloc(nblk * K * T)
    d_rep = c.alloc(nblk * rep_cap * T)
    d_int = c.alloc(nblk * L * T) if want_inter else 0
    try:
        c.upload(d_src, work_blocks)
        c.upload(d_rep, reps)
        st = c.decode_blocks(K, T, nblk, d_src, K * T, lost, nlost, resi, nrep, d_rep, rep_cap * T, d_int, L * T, Kp=Kp)
        c.sync()
        out = c.download(d_src, nblk * K * T).reshape(nblk, K, T)
        inter = c.download(d_int, nblk * L * T).reshape(nblk, L, T) if want_inter else None
    finally:
        c.free(d_src); c.free(d_rep)
        if d_int:
            c.free(d_int)
    return st, out, inter
