"""Multi-GPU plumbing: source blocks are independent, so N GPUs = N processes that each
own a disjoint set of blocks.  No collective touches symbol data; torch.distributed (RCCL on GPUs, gloo on
CPU test runs) is used only for the start/stop barrier and for reducing the timing to rank 0."""
import os
import socket
import subprocess
import sys


def env_rank():
    """(rank, world_size, local_rank) as torch.distributed.run exports them; (0, 1, 0) when run alone."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device_id=None):
    """Join the process group when WORLD_SIZE > 1. Returns (rank, world, local_rank)."""
    rank, world, local = env_rank()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            kw = {}
            if device_id is not None:
                kw["device_id"] = device_id
            dist.init_process_group(backend or "nccl", **kw)
    return rank, world, local


def blocks_of(rank, world, total_blocks):
    """Global ids of the source blocks rank `rank` owns: block b lives on GPU b mod world."""
    return list(range(rank, total_blocks, world))


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def reduce_max(value, world, device=None):
    """max over ranks of a python float (the slowest rank defines the step time)."""
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, world, device=None):
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def finalize(world):
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def launched():
    """True when a launcher (torch.distributed.run, or spawn_ranks below) has set this process's rank."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, env=None, timeout=None):
    """Start `argv` n times on this node, one process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run exports them, rendezvous on 127.0.0.1), and wait for all of them.  Rank 0 inherits stdout (it prints
    the one result line); the other ranks' stdout goes to stderr.  Returns the largest exit code; when a rank fails the others
    are ended (each child is the leader of its own process group: nothing is killed by pattern)."""
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else sys.stderr, start_new_session=True))
    import signal
    import time
    rc, t0 = 0, time.time()
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            rc = max(rc, abs(c))
        if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
            for p in live:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, signal.SIGKILL)
                    p.wait()
            return rc or 124
        if live:
            time.sleep(0.05)
    return rc
