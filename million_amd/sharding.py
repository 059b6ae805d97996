"""Request sharding and rank-level timing helpers for the multi-GPU bench.

The hot path shards by REQUEST: every (request, kv head) owns disjoint code pages, window rows and
outputs, and the codebook is replicated, so ranks never exchange data on the path.  torch.distributed
(backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests) is used only for the barrier around the
timed region and for the max-over-ranks of the elapsed time.
"""
from __future__ import annotations

import time
from typing import Callable, List, Tuple


def shard_requests(n_requests: int, world: int, rank: int) -> List[int]:
    """Contiguous, balanced request ids for `rank` (the first n % world ranks take one more)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    base, extra = divmod(n_requests, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def timed_steps_per_rank(step: Callable[[], None], steps: int, sync: Callable[[], None], dist=None) -> Tuple[float, List[float]]:
    """Time exactly `steps` calls of `step` between barrier + device-sync on both sides.  Returns (the MAX over all ranks of
    the elapsed seconds across both barriers - the job's time, the same number on every rank -, every rank's OWN elapsed
    seconds from the common start to its own device-sync, before it enters the closing barrier: list indexed by rank).  The
    list is what makes a slow rank visible: the job time alone is the slowest rank's, whoever that is."""
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1

    def barrier():
        if multi:
            dist.barrier()

    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    own = time.perf_counter() - t0
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    per_rank = [own]
    if multi:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        per_rank = gather_floats(own, dist)
    return elapsed, per_rank


def gather_floats(x: float, dist=None) -> List[float]:
    """One float of every rank, indexed by rank (all-gather over the job's backend: RCCL on the GPU box, gloo in the CPU tests)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(x)]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, float(x))
    return [float(v) for v in got]


def timed_steps(step: Callable[[], None], steps: int, sync: Callable[[], None], dist=None) -> float:
    """The job's time only (MAX over ranks; every rank gets the same number): timed_steps_per_rank()[0]."""
    return timed_steps_per_rank(step, steps, sync, dist)[0]


def aggregate_throughput(units_per_rank_step: int, steps: int, elapsed_max: float, world: int) -> Tuple[float, float]:
    """Whole-job units/s and ms per step from the max-over-ranks time (weak scaling: every rank does the
    same per-step work)."""
    total = units_per_rank_step * steps * world
    return total / elapsed_max, 1e3 * elapsed_max / steps
