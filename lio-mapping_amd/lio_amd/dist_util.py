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


def make_rccl(lib, rank: int, world: int):
    """In-library RCCL communicator for this rank: rank 0 draws the unique id, torch.distributed (whatever backend the process
    group has) carries the 128 bytes to the others, every rank joins with ncclCommInitRank on its current device."""
    from . import capi

    if world <= 1:
        return capi.Rccl(lib, capi.Rccl.unique_id(lib), 0, 1)
    import torch.distributed as dist

    box = [capi.Rccl.unique_id(lib) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return capi.Rccl(lib, box[0], rank, world)


def refine_keyframes_sharded(lib, maps, keyframes, world: int, rank: int, device: str = "cpu", rccl=None, **cfg):
    """BASELINE.json configs[4] over N ranks: keyframes are independent, so rank r refines keyframes r, r+N, r+2N, ...
    (with the local maps they reference) on its own GPU, and the one exchange of the path is an all-gather of the
    refined poses (7 floats + 2 ints per keyframe).  maps = [(corner_map, surf_map)], keyframes = [(map_index,
    corner_stack, surf_stack, (q_xyzw, p))].  Returns dict(q, p, iterations, rows) for ALL keyframes on every rank."""
    import numpy as np

    from . import capi

    mine = list(range(rank, len(keyframes), world))
    batch = capi.KeyframeBatch(lib, **cfg)
    local_map = {}
    for k in mine:
        mi = keyframes[k][0]
        if mi not in local_map:
            local_map[mi] = batch.add_map(*maps[mi])
        batch.add_keyframe(local_map[mi], *keyframes[k][1:4])
    if rccl is not None and world > 1:   # the exchange inside the library: all-gather from the device pose buffer
        per_rank = -(-len(keyframes) // world)
        g = batch.refine_gather(rccl, per_rank)
        out = np.zeros((len(keyframes), 9), np.float32)
        for rr in range(world):
            idx = list(range(rr, len(keyframes), world))
            out[idx] = g[rr, : len(idx)]
        return dict(q=out[:, 0:4], p=out[:, 4:7], iterations=out[:, 7].astype(np.int32), rows=out[:, 8].astype(np.int32))
    r = batch.refine()
    packed = np.zeros((len(mine), 9), np.float32)
    if mine:
        packed[:, 0:4], packed[:, 4:7] = r["q"], r["p"]
        packed[:, 7], packed[:, 8] = r["iterations"], r["rows"]
    if world > 1:
        import torch
        import torch.distributed as dist

        per_rank = -(-len(keyframes) // world)                      # equal-size slots for all_gather
        buf = torch.zeros((per_rank, 9), dtype=torch.float32, device=device)
        buf[: len(mine)] = torch.from_numpy(packed).to(device)
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        out = np.zeros((len(keyframes), 9), np.float32)
        for rr in range(world):
            idx = list(range(rr, len(keyframes), world))
            out[idx] = gathered[rr].cpu().numpy()[: len(idx)]
    else:
        out = packed
    return dict(q=out[:, 0:4], p=out[:, 4:7], iterations=out[:, 7].astype(np.int32), rows=out[:, 8].astype(np.int32))
