"""Multi-GPU plumbing: source blocks are independent, so N GPUs = N processes that each
own a disjoint set of blocks.  No collective touches symbol data; torch.distributed (RCCL on GPUs, gloo on
CPU test runs) is used only for the start/stop barrier and for reducing the timing to rank 0."""
import os


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


def block_seed(base_seed, global_block):
    """Payload / loss seeds are functions of the GLOBAL block id, so a block's content does not depend on how
    many GPUs share the job."""
    return base_seed, global_block


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
