"""Multi-GPU harness helpers: one process per GPU, replicas over disjoint images (no data-path collective).
The only exchanges are the barrier and the max-over-ranks wall time (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import torch


def rank_seed(base: int, rank: int, i: int) -> int:
    """Disjoint, reproducible seed per (rank, image batch) -- the reference seeds 4396*world+rank (base_evaluator.py)."""
    return base + 977 * rank + i


def max_over_ranks(seconds: float, dist=None, device="cpu") -> float:
    """Wall time of the slowest rank (the job is done when the last replica is)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(images_per_rank: int, steps: int, seconds_max: float, world: int) -> float:
    """Whole-job images/s: every rank produced images_per_rank*steps images in seconds_max."""
    return world * images_per_rank * steps / seconds_max
