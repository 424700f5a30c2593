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


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "LOCAL_WORLD_SIZE")}


@pytest.mark.timeout(300)
def test_spawn_ranks_starts_the_job_by_itself(tmp_path):
    """bench.py --gpus N run on its own goes through shard.spawn_ranks: N processes with the launcher's environment, a
    rendezvous on 127.0.0.1, the largest exit code back."""
    total, K, T = 5, 48, 16
    rc = shard.spawn_ranks(2, [sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(total), str(K), str(T), str(tmp_path)],
                           env=dict(_clean_env(), OMP_NUM_THREADS="1"), timeout=280)
    assert rc == 0
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [r["world"] for r in recs] == [2, 2]
    assert sorted(recs[0]["blocks"] + recs[1]["blocks"]) == list(range(total))
    # a failing rank ends the job with its code
    rc = shard.spawn_ranks(2, [sys.executable, "-c", "import os, sys, time; time.sleep(0 if os.environ['RANK'] == '1' else 30); sys.exit(3)"],
                           env=_clean_env(), timeout=60)
    assert rc == 3


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` must not print an n_gpus=1 line: without two devices it exits non-zero (no GPU here)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 2 and b"refusing" in r.stderr and not r.stdout.strip()
    # and a launcher that started another number of ranks than --gpus says is refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(_clean_env(), RANK="0", LOCAL_RANK="0",
                       WORLD_SIZE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 2 and b"launcher started 1" in r.stderr
