"""Worker of tests/test_sharding_gloo.py: one rank of a world_size-N gloo job.  Each rank 'processes' the
source blocks it owns (with the CPU oracle standing in for the GPU on this GPU-less tier), then the
timing/throughput reduction of bench.py is exercised."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

from nanorq_amd import shard  # noqa: E402
from util import loss_pattern, payload  # noqa: E402


def main():
    total_blocks, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out_dir = sys.argv[4]
    rank, world, local = shard.init(backend="gloo")
    import oracle
    mine = shard.blocks_of(rank, world, total_blocks)
    t0 = time.perf_counter()
    digests = {}
    for b in mine:
        # payload and loss are functions of the GLOBAL block id: a block's content does not depend on how many ranks share the job
        src = payload(K * T, seed=1, block=b).reshape(K, T)
        lost = loss_pattern(K, 0.1, seed=7, block=b)
        esis = np.arange(K, K + len(lost) + 3, dtype=np.uint32)
        rep, _, _ = oracle.encode_block(src, K, T, esis)
        keep = np.setdiff1d(np.arange(K, dtype=np.uint32), lost)
        ok, out, _ = oracle.decode_block(np.concatenate([keep, esis]), np.concatenate([src[keep], rep]), K, T)
        assert ok and np.array_equal(out, src)
        digests[b] = hashlib.sha256(rep.tobytes()).hexdigest()
    shard.barrier(world)
    elapsed = time.perf_counter() - t0 + 0.01 * rank  # rank-dependent so that MAX is distinguishable
    slowest = shard.reduce_max(elapsed, world)
    blocks_done = shard.reduce_sum(len(mine), world)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "world": world, "blocks": mine, "digests": digests, "elapsed": elapsed,
                   "slowest": slowest, "blocks_done": blocks_done}, f)
    shard.finalize(world)


if __name__ == "__main__":
    main()
