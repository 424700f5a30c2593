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


def _fake_sysfs(root, gpus):
    """gpus: list of (numa node, cpulist text) in HIP order; node 0 of the topology is a CPU node (no SIMDs)."""
    nodes = root / "class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (node, cpulist) in enumerate(gpus):
        d = nodes / str(i + 1)
        d.mkdir()
        bus = 0x10 + i
        d.joinpath("properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        p = root / "bus/pci/devices" / ("0000:%02x:00.0" % bus)
        p.mkdir(parents=True)
        p.joinpath("numa_node").write_text("%d\n" % node)
        p.joinpath("local_cpulist").write_text(cpulist + "\n")


def test_ranks_are_bound_to_their_gpus_numa_node(tmp_path):
    """An 8-GPU MI355X board: four GPUs per socket.  Every rank gets cores of ITS GPU's NUMA node, ranks that share a node
    get disjoint slices, nothing outside the allowed set is handed out; without a topology the allowed cores are split evenly."""
    _fake_sysfs(tmp_path, [(0, "0-63,128-191")] * 4 + [(1, "64-127,192-255")] * 4)
    gpus = shard.gpu_cpu_lists(str(tmp_path))
    assert [g[0] for g in gpus] == [0, 0, 0, 0, 1, 1, 1, 1] and len(gpus[0][1]) == 128
    sets = shard.rank_cpus(8, allowed=range(256), gpus=gpus)
    assert all(len(s) == 32 for s in sets)
    assert all(set(sets[r]) <= set(gpus[r][1]) for r in range(8))
    assert len(set(c for s in sets for c in s)) == 256                       # disjoint
    # a cgroup that allows only part of a node; a device list that maps ranks to other GPUs; a single-GPU plumbing run
    sets = shard.rank_cpus(2, allowed=range(0, 16), gpus=gpus, device_of_rank=[0, 1])
    assert sorted(sets[0] + sets[1]) == list(range(16)) and not set(sets[0]) & set(sets[1])
    sets = shard.rank_cpus(2, allowed=range(256), gpus=gpus, device_of_rank=[5, 1])
    assert set(sets[0]) <= set(gpus[5][1]) and set(sets[1]) <= set(gpus[1][1])
    sets = shard.rank_cpus(4, allowed=range(256), gpus=gpus, device_of_rank=[0, 0, 0, 0])
    assert all(set(s) <= set(gpus[0][1]) and len(s) == 32 for s in sets) and len(set(c for s in sets for c in s)) == 128
    sets = shard.rank_cpus(3, allowed=[2, 3, 4, 5, 6, 7], gpus=[])
    assert sets == [[2, 3], [4, 5], [6, 7]]
    assert shard.rank_cpus(4, allowed=[0, 1], gpus=[]) and all(shard.rank_cpus(4, allowed=[0, 1], gpus=[]))  # more ranks than cores: everybody gets one
    assert shard.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert shard.visible_devices({"HIP_VISIBLE_DEVICES": "4,5"}) == [4, 5] and shard.visible_devices({}) is None


@pytest.mark.timeout(120)
def test_spawned_ranks_see_their_binding(tmp_path):
    """What spawn_ranks hands down: a disjoint CPU set per rank (applied before the child starts), NANORQ_RANK_CPUS naming it, and
    an OMP_NUM_THREADS that fits it."""
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("one core")
    code = ("import os, json; json.dump({'aff': sorted(os.sched_getaffinity(0)), 'env': os.environ['NANORQ_RANK_CPUS'], "
            "'omp': os.environ['OMP_NUM_THREADS']}, open(os.path.join(%r, 'r' + os.environ['RANK']), 'w'))" % str(tmp_path))
    env = {k: v for k, v in _clean_env().items() if k != "OMP_NUM_THREADS"}
    assert shard.spawn_ranks(2, [sys.executable, "-c", code], env=env, timeout=60, sysfs=str(tmp_path / "nosys")) == 0
    recs = [json.load(open(tmp_path / ("r%d" % r))) for r in range(2)]
    assert not set(recs[0]["aff"]) & set(recs[1]["aff"])
    for r in recs:
        assert r["aff"] == sorted(int(x) for x in r["env"].split(",")) and 1 <= int(r["omp"]) <= len(r["aff"])
    assert set(recs[0]["aff"]) | set(recs[1]["aff"]) <= set(os.sched_getaffinity(0))


@pytest.mark.timeout(600)
def test_eight_rank_job_gives_the_one_rank_digests(tmp_path):
    """The split the driver's 8-GPU run uses (block b -> rank b mod 8, eight processes started by spawn_ranks, gloo here): every
    block's repair symbols are the ones a one-rank run produces -- a block's content is a function of its GLOBAL id."""
    total, K, T = 19, 40, 16
    d8, d1 = tmp_path / "w8", tmp_path / "w1"
    d8.mkdir(); d1.mkdir()
    worker = [sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(total), str(K), str(T)]
    assert shard.spawn_ranks(8, worker + [str(d8)], env=dict(_clean_env(), OMP_NUM_THREADS="1"), timeout=500) == 0
    assert shard.spawn_ranks(1, worker + [str(d1)], env=dict(_clean_env(), OMP_NUM_THREADS="1"), timeout=200) == 0
    recs = [json.load(open(d8 / ("rank%d.json" % r))) for r in range(8)]
    assert [r["world"] for r in recs] == [8] * 8 and all(r["blocks_done"] == total for r in recs)
    assert [r["blocks"] for r in recs] == [list(range(r, total, 8)) for r in range(8)]
    merged = {}
    for r in recs:
        merged.update(r["digests"])
    assert merged == json.load(open(d1 / "rank0.json"))["digests"] and len(merged) == total


def test_a_container_with_some_of_the_nodes_gpus(tmp_path):
    """Eight KFD nodes, the PCI directory of one (what a 1-GPU slice of an 8-GPU node shows): that one is HIP device 0."""
    _fake_sysfs(tmp_path, [(0, "0-63")] * 3 + [(1, "64-127")] + [(0, "0-63")] * 4)
    import shutil
    for i in (0, 1, 2, 4, 5, 6, 7):
        shutil.rmtree(tmp_path / "bus/pci/devices" / ("0000:%02x:00.0" % (0x10 + i)))
    gpus = shard.gpu_cpu_lists(str(tmp_path))
    assert gpus == [(1, list(range(64, 128)))]
    assert shard.rank_cpus(2, allowed=range(128), gpus=gpus, device_of_rank=[0, 0]) == [list(range(64, 96)), list(range(96, 128))]


@pytest.mark.timeout(120)
def test_a_rank_of_another_launcher_binds_itself(tmp_path):
    """Under torch.distributed.run (the driver's N-GPU runs) nobody binds the ranks: bench.py calls shard.bind_self() before it
    imports torch -- same slices as spawn_ranks hands out, disjoint over the ranks, NANORQ_RANK_CPUS set; a rank that
    spawn_ranks has already bound, or NANORQ_NO_BIND=1, is left alone."""
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("one core")
    code = ("import os, sys, json; sys.path.insert(0, %r)\n"
            "from nanorq_amd import shard\n"
            "got = shard.bind_self(sysfs=%r)\n"
            "json.dump({'got': got, 'aff': sorted(os.sched_getaffinity(0)), 'env': os.environ.get('NANORQ_RANK_CPUS')}, "
            "open(os.path.join(%r, 'b' + os.environ['LOCAL_RANK'] + os.environ.get('TAG', '')), 'w'))\n" % (ROOT, str(tmp_path / "nosys"), str(tmp_path)))
    base = dict(_clean_env(), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2")
    for r in range(2):
        subprocess.run([sys.executable, "-c", code], env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), check=True, timeout=60)
    recs = [json.load(open(tmp_path / ("b%d" % r))) for r in range(2)]
    assert all(x["got"] == x["aff"] and x["env"] for x in recs) and not set(recs[0]["aff"]) & set(recs[1]["aff"])
    subprocess.run([sys.executable, "-c", code], env=dict(base, RANK="0", LOCAL_RANK="0", NANORQ_NO_BIND="1", TAG="n"), check=True, timeout=60)
    assert json.load(open(tmp_path / "b0n"))["got"] is None
