"""Multi-GPU sharding of the hot path: independent chunks, no data-path collective.

A chunk's DP depends only on its <= part+overlap bases and the (replicated) template set
(reference main.cpp:88-96), so reads are dealt to ranks in contiguous blocks and every rank runs
the same single-GPU engine on its block.  torch.distributed (RCCL on GPUs, gloo in the CPU tests)
is used only for the barrier / max-over-ranks timing and for gathering small row counts.
"""
import os


def world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def block_range(n_items, rank, world_size):
    """Contiguous block [lo, hi) of n_items for `rank`: sizes differ by at most one, order kept."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def weak_range(items_per_rank, rank):
    """Weak scaling: every rank owns items_per_rank consecutive items of an unbounded stream."""
    return rank * items_per_rank, (rank + 1) * items_per_rank


def init_process_group(backend=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1; returns the module or None."""
    rank, local_rank, ws = world()
    if ws <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return dist


def barrier(dist, device=None):
    if dist is not None:
        if device is not None:
            dist.barrier(device_ids=[device])
        else:
            dist.barrier()


def max_over_ranks(dist, value, device="cpu"):
    """MAX over ranks of a python float (the job is as slow as its slowest rank)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
