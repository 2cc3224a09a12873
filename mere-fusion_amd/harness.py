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


def timed_steps(step_fn, steps, warmup, sync_fn=None, device=None, collective=True):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both
    sides.  Returns the max-over-ranks elapsed seconds (identical on every rank).
    collective=False: a measurement only THIS rank makes (rank 0's extra legs at --gpus N > 1) -- no barrier, no reduction."""
    sync = sync_fn or (lambda: None)
    world = dist.get_world_size() if (collective and dist.is_initialized()) else 1
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


class SessionPlacer:
    """Node-level admission and placement of talking-head sessions (the path shards by session, no collective).

    The reference admits sessions against one global cap and runs them all on `cuda:0` (app.py:42,79-80 `if current_sessions >= MAX_SESSIONS:
    ... 429 'Maximum number of sessions reached'`, app.py:705 `--max_session`, lipreal.py:29 / musereal.py `device = 'cuda'`).  On an 8-GPU node
    the cap is the SUM of what each GPU sustains -- its measured capacity (bench.py `paced_sessions`: sessions at >= 25 fps with the p99 latency
    bound held), not a guess -- and a new session goes to the GPU with the lowest load fraction that still has room (ties: lowest index), so a
    half-empty node stays evenly loaded and no GPU is ever asked for more than it was measured to hold.

    Pure host logic, deterministic: every rank that builds it from the same capacities (all_gather of each rank's measured number, see
    `from_measured`) and sees the same start / stop sequence computes the same placement -- no data-path exchange is needed."""

    def __init__(self, capacities):
        self.capacity = [int(c) for c in capacities]
        if not self.capacity or min(self.capacity) < 0:
            raise ValueError("one non-negative capacity per GPU is required")
        self.active = {}                                              # session id -> gpu
        self.load = [0] * len(self.capacity)

    @classmethod
    def from_measured(cls, my_capacity):
        """Every rank contributes the capacity it measured on its own GPU (one all_gather of an int; gloo in tests, RCCL on the node)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            caps = [None] * dist.get_world_size()
            dist.all_gather_object(caps, int(my_capacity))
        else:
            caps = [int(my_capacity)]
        return cls(caps)

    @property
    def max_sessions(self):
        return sum(self.capacity)

    def start_session(self, session_id):
        """-> (code, gpu): (0, g) placed on GPU g; (1, None) 'Maximum number of sessions reached' (app.py:79-80)."""
        if session_id in self.active:
            return 0, self.active[session_id]
        if len(self.active) >= self.max_sessions:
            return 1, None
        g = min((g for g in range(len(self.capacity)) if self.load[g] < self.capacity[g]),
                key=lambda g: (self.load[g] / self.capacity[g], g))
        self.active[session_id] = g
        self.load[g] += 1
        return 0, g

    def stop_session(self, session_id):
        """app.py:97-120: frees the session's place; unknown ids are reported, not raised (the endpoint answers 404)."""
        g = self.active.pop(session_id, None)
        if g is None:
            return 1
        self.load[g] -= 1
        return 0

    def sessions_of(self, gpu):
        return [s for s, g in self.active.items() if g == gpu]
