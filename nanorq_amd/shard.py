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


# ---- CPU placement of the ranks -----------------------------------------------------------------------------------------
# One process per GPU means N host processes on one node; left alone they all float over every core, torch / OpenMP start a
# thread per hardware thread in each of them, and a rank's page-locked staging memory may land on the other socket from its
# GPU.  spawn_ranks therefore gives every rank the cores of its GPU's NUMA node (a disjoint slice of them when several GPUs
# share a node: 4 per socket on an 8-GPU MI355X board) and an OMP_NUM_THREADS that fits the slice.
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_cpu_lists(sysfs="/sys"):
    """Per GPU, in HIP device order: (numa node, cpus local to it), from the KFD topology (the nodes with SIMDs are the
    GPUs, in the order the runtime numbers them) and the PCI device's local_cpulist.  [] when the tree is not there."""
    import glob
    out = []
    nodes = sorted(glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*")), key=lambda d: int(os.path.basename(d)))
    for d in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(d, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue   # a CPU node
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
            pdir = os.path.join(sysfs, "bus/pci/devices", bdf)
            node = int(open(os.path.join(pdir, "numa_node")).read().strip())
            cpus = parse_cpulist(open(os.path.join(pdir, "local_cpulist")).read())
            out.append((node, cpus))
        except (OSError, ValueError):
            out.append((-1, []))
    # a container that is given some of a node's GPUs still sees every KFD node, but only its own GPUs' PCI directories: the
    # devices the runtime numbers 0, 1, ... are then the nodes whose directories exist, in order (seen on the 1-GPU test boxes:
    # eight nodes, the fourth one valid, and that one is HIP device 0)
    if any(c for _, c in out) and not all(c for _, c in out):
        out = [g for g in out if g[1]]
    return out


def visible_devices(env):
    """HIP device index -> physical GPU index under ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES (None: identity)."""
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(k)
        if v:
            try:
                return [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return None
    return None


def rank_cpus(n, allowed=None, gpus=None, device_of_rank=None):
    """CPU set of each of n ranks: the allowed cpus of the rank's GPU's NUMA node, dealt out disjointly among the ranks that
    share the node; without topology (or when a node has no allowed cpu) an even split of the allowed cpus.  Every rank gets
    at least one cpu."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    device_of_rank = device_of_rank or list(range(n))
    sets = [None] * n
    if gpus:
        by_node = {}
        for r in range(n):
            d = device_of_rank[r]
            if 0 <= d < len(gpus) and gpus[d][1]:
                by_node.setdefault((gpus[d][0], tuple(gpus[d][1])), []).append(r)
        for (_, cpus), ranks in by_node.items():
            mine = [c for c in cpus if c in set(allowed)]
            if len(mine) < len(ranks):
                continue
            per = len(mine) // len(ranks)
            for i, r in enumerate(ranks):
                sets[r] = mine[i * per:(i + 1) * per]
    rest = [r for r in range(n) if not sets[r]]
    if rest:
        taken = set(c for s_ in sets if s_ for c in s_)
        pool = [c for c in allowed if c not in taken] or allowed
        per = max(1, len(pool) // len(rest))
        for i, r in enumerate(rest):
            sets[r] = pool[(i * per) % len(pool):(i * per) % len(pool) + per] or [pool[i % len(pool)]]
    return sets


ORIG_AFFINITY = None   # what this process was allowed before bind_self narrowed it (a child that works for ALL devices gets it back)


def bind_self(sysfs="/sys"):
    """A rank started by another launcher (torch.distributed.run: the driver's N-GPU runs) binds ITSELF the way spawn_ranks binds
    its children: to its slice of the cores of its GPU's NUMA node, before torch (and its thread pools) come up.  Returns the
    cpu list, or None when nothing was done (already bound by spawn_ranks, NANORQ_NO_BIND=1, no affinity call, one rank)."""
    if os.environ.get("NANORQ_NO_BIND") == "1" or os.environ.get("NANORQ_RANK_CPUS") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        n = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or "1")
        r = int(os.environ.get("LOCAL_RANK", "0"))
    except ValueError:
        return None
    if n <= 1 or r >= n:
        return None
    vis = visible_devices(os.environ)
    dev_of = [(vis[k] if vis and k < len(vis) else k) for k in range(n)]
    if os.environ.get("NANORQ_FORCE_DEVICE", "") != "":
        dev_of = [int(os.environ["NANORQ_FORCE_DEVICE"])] * n
    cpus = rank_cpus(n, gpus=gpu_cpu_lists(sysfs), device_of_rank=dev_of)[r]
    if not cpus:
        return None
    global ORIG_AFFINITY
    try:
        ORIG_AFFINITY = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(cpus))
    except OSError:
        return None
    os.environ["NANORQ_RANK_CPUS"] = ",".join(str(c) for c in cpus)
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(len(cpus), 16))))
    return cpus


def spawn_ranks(n, argv, env=None, timeout=None, bind=True, sysfs="/sys"):
    """Start `argv` n times on this node, one process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run exports them, rendezvous on 127.0.0.1), and wait for all of them.  Rank 0 inherits stdout (it prints
    the one result line); the other ranks' stdout goes to stderr.  Returns the largest exit code; when a rank fails the others
    are ended (each child is the leader of its own process group: nothing is killed by pattern)."""
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(n)
    cpus = [None] * n
    if bind and base.get("NANORQ_NO_BIND") != "1" and hasattr(os, "sched_setaffinity"):
        vis = visible_devices(base)
        dev_of = [(vis[r] if vis and r < len(vis) else r) for r in range(n)]
        if base.get("NANORQ_FORCE_DEVICE", "") != "":   # (plumbing runs: every rank on one GPU -- its node for all of them)
            dev_of = [int(base["NANORQ_FORCE_DEVICE"])] * n
        cpus = rank_cpus(n, gpus=gpu_cpu_lists(sysfs), device_of_rank=dev_of)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        pre = None
        if cpus[r]:
            e["NANORQ_RANK_CPUS"] = ",".join(str(c) for c in cpus[r])
            e.setdefault("OMP_NUM_THREADS", str(max(1, min(len(cpus[r]), 16))))   # (torch / numpy / OpenMP pools sized for the slice)
            pre = (lambda cs: (lambda: os.sched_setaffinity(0, cs)))(set(cpus[r]))
        procs.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else sys.stderr, start_new_session=True, preexec_fn=pre))
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
