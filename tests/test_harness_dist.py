"""CPU, world_size 2 over gloo: the bench harness' N>1 path -- session sharding, barrier-bracketed
timing, MAX over ranks -- without any GPU."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from mere_fusion_amd import harness as H
    r, lr, w = H.init_dist("gloo")
    assert (r, w) == (rank, world)
    mine = H.shard_sessions(5, r, w)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 if rank == 0 else 0.05)   # rank 1 is the slow one

    elapsed = H.timed_steps(step, steps=4, warmup=2)
    # node-level placement: every rank contributes the capacity it "measured" (3 on rank 0, 5 on rank 1); both build the same placer and, fed the same
    # start / stop sequence, compute the same placement without exchanging anything else
    placer = H.SessionPlacer.from_measured(3 if rank == 0 else 5)
    log = [placer.start_session(f"s{i}") for i in range(10)]
    log.append(placer.stop_session("s0"))
    log.append(placer.start_session("late"))
    q.put((rank, mine, len(calls), elapsed, H.aggregate_value(16, 4, elapsed, w), placer.capacity, log, sorted(placer.sessions_of(rank))))
    torch.distributed.destroy_process_group()


def test_two_rank_harness():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, c0, e0, v0, cap0, log0, own0), (r1, s1, c1, e1, v1, cap1, log1, own1) = res
    assert cap0 == cap1 == [3, 5] and log0 == log1
    # least-loaded-fraction first, ties to the lower GPU; the 9th and 10th requests meet the node cap (app.py:79-80: 'Maximum number of sessions reached')
    assert log0[:10] == [(0, 0), (0, 1), (0, 1), (0, 0), (0, 1), (0, 1), (0, 0), (0, 1), (1, None), (1, None)]
    assert log0[10] == 0 and log0[11] == (0, 0)                 # a stopped session frees its GPU for the next one
    assert own0 == ["late", "s3", "s6"] and own1 == ["s1", "s2", "s4", "s5", "s7"]
    assert s0 == [0, 2, 4] and s1 == [1, 3]               # session s -> rank s mod world
    assert c0 == c1 == 6                                   # 2 warm-up + exactly 4 timed
    assert e0 == pytest.approx(e1)                         # MAX over ranks, same on both
    assert e0 >= 4 * 0.05 * 0.95                           # bounded below by the slow rank
    assert v0 == pytest.approx(16 * 4 * 2 / e0)            # whole-job units / max time


def test_single_process_harness():
    from mere_fusion_amd import harness as H
    assert H.dist_env()[2] == int(os.environ.get("WORLD_SIZE", 1))
    n = []
    e = H.timed_steps(lambda: n.append(1), steps=3, warmup=1)
    assert len(n) == 4 and e > 0
    assert H.shard_sessions(4, 0, 1) == [0, 1, 2, 3]


def test_session_placer_admission_and_balance():
    from mere_fusion_amd import harness as H
    p = H.SessionPlacer([22, 22, 0, 21])                       # a GPU that measured 0 never gets a session
    assert p.max_sessions == 65
    placed = [p.start_session(i) for i in range(66)]
    assert [c for c, _ in placed].count(0) == 65 and placed[-1] == (1, None)
    assert p.load == [22, 22, 0, 21] and p.start_session(5) == (0, placed[5][1])      # asking again for a live session returns its place
    assert p.stop_session("nobody") == 1
    for i in range(0, 65, 2):
        p.stop_session(i)
    assert max(abs(p.load[a] / p.capacity[a] - p.load[b] / p.capacity[b]) for a in (0, 1, 3) for b in (0, 1, 3)) < 0.15
    with pytest.raises(ValueError):
        H.SessionPlacer([])
