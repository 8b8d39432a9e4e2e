"""Timing and multi-rank aggregation for bench.py, separated so that the N>1 path can be tested on CPU (gloo).

The hot path shards by game -> device with no exchange step (rust/kz-selfplay/src/server/server.rs:325-331), so ranks
never exchange tensors: torch.distributed only carries the barrier and the max of the elapsed time.
"""
import os
import time
from typing import Callable, Optional, Tuple


def rank_info() -> Tuple[int, int, int, bool]:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    return rank, local_rank, world, distributed


def init_control_plane():
    """Returns the torch.distributed module with an initialised gloo group, or None when not launched by torchrun."""
    rank, _, world, distributed = rank_info()
    if not distributed:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        # gloo's C++ side announces its connections on stdout; stdout carries exactly one JSON line, so the
        # announcement goes to stderr
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return dist


def gather_strings(dist, s: str):
    """Every rank's string, in rank order, on every rank (e.g. the PCI bus id of the GPU a rank drives)."""
    if dist is None:
        return [s]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, s)
    return out


def board_seed(rank: int) -> int:
    """Every rank evaluates its own boards (independent units: weak scaling)."""
    return 1000 + rank


def run_timed(step: Callable[[int], None], sync: Callable[[], None], steps: int, warmup: int, dist=None,
              on_timed_start: Optional[Callable[[], None]] = None) -> float:
    """W untimed warm-up steps, then exactly `steps` steps bracketed by barrier + device sync on both sides.
    Returns the MAX over ranks of the elapsed seconds."""
    for i in range(warmup):
        step(i)
    sync()
    if on_timed_start is not None:
        on_timed_start()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def whole_job_value(steps: int, batch: int, world: int, elapsed_max: float) -> float:
    """evals/s of the whole job: the units all ranks processed / the slowest rank's time."""
    return steps * batch * world / elapsed_max
