"""One-process-per-GPU plumbing for the data-parallel forward.

HRNet.forward has no cross-sample operation, so the batch shards over ranks with NO data-path collective:
every rank runs the same kernels on its own B samples ("replicas"; weak scaling).  torch.distributed (backend
"nccl" == RCCL on ROCm, "gloo" in the CPU tests) is used only for the rendezvous, the barriers around the timed
region and the max-over-ranks reduction of the elapsed time.  Gradient all-reduce belongs to the training
path (row f3) and is not built yet.
"""
import os

import torch
import torch.distributed as dist


def world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched directly."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Join the process group when WORLD_SIZE > 1.  Returns (rank, local_rank, world_size)."""
    rank, local_rank, ws = world()
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, local_rank, ws


def barrier(device=None):
    if dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (identity without a process group)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shard(global_batch, rank, world_size):
    """Contiguous sample range [lo, hi) of rank `rank` (rank r takes samples [r*B, (r+1)*B))."""
    if global_batch % world_size != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world_size}")
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
