"""N>1 path on CPU: two gloo ranks shard the source blocks exactly as bench.py --gpus N does on GPUs
(block b -> rank b mod N, no data-path collective, MAX-reduced timing)."""
import json
import os
import subprocess
import sys

import pytest

from nanorq_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_partition_is_disjoint_and_complete():
    for world in (1, 2, 4, 8):
        for total in (1, 8, 64, 67):
            owned = [shard.blocks_of(r, world, total) for r in range(world)]
            flat = sorted(b for o in owned for b in o)
            assert flat == list(range(total))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
    assert shard.env_rank()[1] >= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_job(tmp_path):
    total, K, T = 6, 64, 32
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", OMP_NUM_THREADS="1")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(total), str(K),
                                       str(T), str(tmp_path)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert sorted(recs[0]["blocks"] + recs[1]["blocks"]) == list(range(total))
    assert not set(recs[0]["blocks"]) & set(recs[1]["blocks"])
    # the reduced step time is the slowest rank's, identical on both ranks; the block count is the job's
    assert recs[0]["slowest"] == recs[1]["slowest"] == max(recs[0]["elapsed"], recs[1]["elapsed"])
    assert recs[0]["blocks_done"] == recs[1]["blocks_done"] == total
    # a block's content does not depend on the number of ranks: same digests from a 1-rank run
    e = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    solo_dir = tmp_path / "solo"
    solo_dir.mkdir()
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(total), str(K), str(T),
                    str(solo_dir)], env=e, check=True, timeout=280)
    solo = json.load(open(solo_dir / "rank0.json"))
    merged = dict(recs[0]["digests"], **recs[1]["digests"])
    assert merged == solo["digests"]
