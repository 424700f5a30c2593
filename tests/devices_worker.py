"""Runs in a process of its own (the device list of the object layer is read once, from NANORQ_HIP_DEVICES): one object
through the batched calls on page-locked memory; prints a JSON line with the number of contexts and the SHA-256 of the
packets produced and of the recovered object."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.dirname(os.path.abspath(__file__))]
from capi import api, pinned_array, pinned_io  # noqa: E402
from util import loss_pattern, payload  # noqa: E402

K, T, Z = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
loss = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
L = api()
F = K * T * Z - 7          # (a short tail: the last symbol is cut at the transfer length)
data = payload(F, seed=5, block=0)
rq = L.nanorq_encoder_new_ex(F, T, K, 0, 8)
assert rq and L.nanorq_blocks(rq) == Z
io, mem = pinned_io(F)
mem[:] = data
assert L.nanorq_generate_symbols_all(rq, io) == Z
lost = [loss_pattern(K, loss, seed=9, block=b) for b in range(Z)]
nrep = max(len(x) for x in lost) + 2
rep = np.zeros((Z, nrep, T), np.uint8)
assert L.nanorq_encode_range_all(rq, rep.ctypes.data_as(C.c_void_p), K, nrep, io) == Z * nrep * T
one = np.zeros((nrep, T), np.uint8)                      # the per-block call gives the same symbols
for b in (0, Z - 1):
    assert L.nanorq_encode_range(rq, one.ctypes.data_as(C.c_void_p), K, nrep, b, io) == nrep * T
    assert np.array_equal(one, rep[b])
oti = (L.nanorq_oti_common(rq), L.nanorq_oti_scheme_specific(rq))
L.nanorq_free(rq)
io.contents.destroy(io)

dq = L.nanorq_decoder_new(*oti)
oio, out = pinned_io(F)
out[:] = 0
src = np.zeros(K * T * Z, np.uint8)
src[:F] = data
src = src.reshape(Z, K, T)
rows, tags = [], []
for b in range(Z):
    keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost[b])
    rows += [src[b][keep], rep[b][:len(lost[b]) + 2]]
    tags += [(b << 24) | keep, (b << 24) | (K + np.arange(len(lost[b]) + 2, dtype=np.uint32))]
rows, tags = np.concatenate(rows), np.concatenate(tags).astype(np.uint32)
addr, blob = pinned_array(rows.size)
blob[:] = rows.reshape(-1)
res = np.zeros(len(tags), np.int32)
added = L.nanorq_decoder_add_symbols(dq, C.c_void_p(addr), tags.ctypes.data_as(C.POINTER(C.c_uint32)), len(tags), res.ctypes.data_as(C.POINTER(C.c_int)), oio)
done = L.nanorq_repair_all(dq, oio)
ok = added == len(tags) and done == Z and np.array_equal(out, data) and not res.any()
obj_sha = hashlib.sha256(out.tobytes()).hexdigest()   # (before the context that owns the memory goes)
L.nanorq_free(dq)
oio.contents.destroy(oio)
L.nanorq_pinned_free(addr)
L.nanorq_trim()
print(json.dumps({"devices": int(L.nanorq_devices()), "ok": bool(ok), "added": int(added), "done": int(done),
                  "packets": hashlib.sha256(rep.tobytes()).hexdigest(), "object": obj_sha,
                  "max_repair_per_block": int(nrep)}))
