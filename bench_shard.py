"""Control plane shared by bench.py and its CPU test: batch partitioning and the cross-rank timing
reduce.  No collective touches frame data (frames are independent)."""
from __future__ import annotations


def shard_frames(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, disjoint, covering partition of ``total`` frames: ranks < total % world get one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def aggregate_fps(frames_per_rank: int, steps: int, elapsed: float, world: int) -> tuple[float, float]:
    """Whole-job frames/s = frames of all ranks / MAX over ranks of the timed region."""
    tmax = elapsed
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t[0])
    return frames_per_rank * world * steps / tmax, tmax


def gather_floats(value: float, world: int) -> list[float]:
    """One float per rank, in rank order, on every rank (gloo all_gather on the host; timing data only)."""
    if world == 1:
        return [float(value)]
    import torch
    import torch.distributed as dist
    out = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(out, torch.tensor([value], dtype=torch.float64))
    return [float(t[0]) for t in out]
