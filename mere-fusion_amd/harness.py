"""Timing / multi-GPU harness shared by bench.py and the multi-process CPU tests.

Sessions are independent units (one avatar, one audio stream: lipreal.py:161-172), so the path
shards by session with NO data-path collective ("replicas only": every GPU holds the full weights).
torch.distributed is used for exactly two things: the barrier around the timed region and the
MAX over ranks of the elapsed time.  With backend "nccl" (= RCCL) on the GPU box, "gloo" in tests.
"""
import os
import time

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) standalone."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_dist(backend):
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_sessions(n_sessions, rank, world):
    """Session s runs on GPU s mod world.  Returns the session ids of this rank."""
    return [s for s in range(n_sessions) if s % world == rank]


def timed_steps(step_fn, steps, warmup, sync_fn=None, device=None):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both
    sides.  Returns the max-over-ranks elapsed seconds (identical on every rank)."""
    sync = sync_fn or (lambda: None)
    world = dist.get_world_size() if dist.is_initialized() else 1
    for _ in range(warmup):
        step_fn()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_value(units_per_step_per_rank, steps, elapsed, world):
    """Whole-job throughput: units all ranks processed / max-over-ranks time."""
    return units_per_step_per_rank * steps * world / elapsed
