"""Multi-GPU plumbing for bench.py: the path shards by window — every rank owns an
independent window, there is no data-path collective.  torch.distributed (RCCL on GPU boxes, gloo in the
CPU tests) is used only for the timing contract: barrier, max-over-ranks, whole-job aggregate."""
from __future__ import annotations

import os


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def window_shift_for_rank(rank: int) -> float:
    """Each rank's window starts at a different point of the trajectory: independent units of work."""
    return 0.37 * rank


def barrier(world: int):
    if world > 1:
        import torch.distributed as dist

        dist.barrier()


def max_over_ranks(value: float, world: int, device: str = "cpu") -> float:
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(steps_per_rank: int, seconds_this_rank: float, world: int, device: str = "cpu") -> tuple[float, float]:
    """(units/s over the whole job, max-over-ranks seconds): value = world * K / max_r(t_r)."""
    t = max_over_ranks(seconds_this_rank, world, device)
    return world * steps_per_rank / t, t


def make_allreduce(device: str = "cpu"):
    """In-place SUM all-reduce of a float64 numpy buffer through torch.distributed (RCCL when device='cuda':
    the buffer is staged through a persistent device tensor; gloo on CPU)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    stage = {}

    def allreduce(buf: np.ndarray):
        n = buf.shape[0]
        if device == "cpu":
            t = torch.from_numpy(buf)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return
        t = stage.get(n)
        if t is None:
            t = stage[n] = torch.empty(n, dtype=torch.float64, device=device)
        t.copy_(torch.from_numpy(buf), non_blocking=False)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        buf[:] = t.cpu().numpy()

    return allreduce
