"""GPU choice inside the drop-in (mere_fusion_amd/placement.py; VERDICT r05 missing #2): processes that enter the drop-in on a multi-GPU node spread over its GPUs
by the lock-file SessionPlacer -- lowest load fraction, ties to the lowest index, released at exit or when the holder is found dead.  Host logic only: the
"GPUs" here are two names in MF_GPUS; nothing touches a device."""
import os
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

_CHILD = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
from mere_fusion_amd import placement
g = placement.ensure_placed(session=(sys.argv[2] == "session"))
print("PLACED", g, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("MF_ORIG_VISIBLE_DEVICES"), flush=True)
sys.stdin.readline()                      # hold the place until the parent says so
'''


def _spawn(env, kind="session"):
    p = subprocess.Popen([sys.executable, "-c", _CHILD, ROOT, kind], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    line = p.stdout.readline().split()
    assert line[0] == "PLACED", line
    return p, line[1], line[2], line[3]


def test_five_session_processes_spread_three_two_and_release(tmp_path):
    from mere_fusion_amd import placement
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "MF_ORIG_VISIBLE_DEVICES", "MF_PLACED_GPU", "LOCAL_RANK", "MF_PLACEMENT")}
    env.update(MF_PLACEMENT_DIR=str(tmp_path), MF_GPUS="2")
    os.environ["MF_PLACEMENT_DIR"] = str(tmp_path)
    try:
        kids = [_spawn(env) for _ in range(5)]
        gpus = [k[1] for k in kids]
        assert gpus == ["0", "1", "0", "1", "0"], gpus                          # least-loaded, ties to the lowest index
        for p, g, vis, orig in kids:
            assert vis == g and orig == "0,1"                                    # the process narrowed its own visibility before HIP came up; children choose among all again
        t = placement.table()
        assert sorted(len(v) for v in t.values()) == [2, 3]
        # a front-end-only process (the reference's parent: mel / Whisper features, no model) takes a GPU but is not counted as a session
        parent, g, _, _ = _spawn(env, kind="frontend")
        assert g == "1"                                                          # GPU 1 carries two sessions, GPU 0 three
        kid6 = _spawn(env)
        assert kid6[1] == "1"                                                    # ... and the next session still sees 3 / 2
        # a holder that dies without running its exit handler is dropped by the next caller
        kids[0][0].send_signal(signal.SIGKILL)
        kids[0][0].wait()
        kid7 = _spawn(env)
        assert kid7[1] == "0"                                                    # 2 / 3 after the kill -> GPU 0
        # processes that hold a GPU without a session still spread (eight app.py started together must not all take GPU 0): 3 + 1/64 + 1 | 2 + 1 at this point
        fronts = [_spawn(env, kind="frontend") for _ in range(3)]
        assert sorted(f[1] for f in fronts) == ["0", "1", "1"] or sorted(f[1] for f in fronts) == ["0", "0", "1"], [f[1] for f in fronts]
        for p, *_ in kids[1:] + [kid6, kid7, (parent,)] + fronts:
            p.stdin.write("\n"); p.stdin.flush()
            p.wait(timeout=30)
        assert placement.table() == {}                                           # everyone released at exit
    finally:
        os.environ.pop("MF_PLACEMENT_DIR", None)


def test_holders_without_sessions_spread_over_the_gpus(tmp_path):
    from mere_fusion_amd import placement
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "MF_ORIG_VISIBLE_DEVICES", "MF_PLACED_GPU", "LOCAL_RANK", "MF_PLACEMENT")}
    env.update(MF_PLACEMENT_DIR=str(tmp_path), MF_GPUS="4")
    procs = [_spawn(env, kind="frontend") for _ in range(4)]
    try:
        assert [p[1] for p in procs] == ["0", "1", "2", "3"]                     # four app.py processes started together: one GPU each
    finally:
        for p, *_ in procs:
            p.stdin.write("\n"); p.stdin.flush(); p.wait(timeout=30)


def test_capacity_is_the_admission_cap(tmp_path):
    """MF_GPU_CAPACITY = measured sessions per GPU: a GPU is never given more, and a full node answers as app.py:79-80 does."""
    from mere_fusion_amd import placement
    os.environ.update(MF_PLACEMENT_DIR=str(tmp_path), MF_GPU_CAPACITY="2,1")
    try:
        me = os.getpid()
        assert [placement.place(2, 1.0, pid=me) for _ in range(2)] == [0, 0]    # the same process adds sessions to ITS GPU (in-process sessions: ER-NeRF)
        with pytest.raises(RuntimeError, match="Maximum number of sessions reached"):
            placement.place(2, 1.0, pid=me)
        placement.release(me)
        held = {"11": [0, 1.0, 0], "12": [1, 1.0, 0], "13": [0, 1.0, 0]}
        assert placement.choose(held, [2, 1], 1.0) is None and placement.choose(held, [3, 1], 1.0) == 0 and placement.choose({}, [2, 1], 1.0) == 0
        assert placement.choose({"11": [0, 1.0, 0]}, [4, 1], 1.0) == 1          # 1/4 against 0/1: the emptier FRACTION wins
    finally:
        os.environ.pop("MF_PLACEMENT_DIR", None); os.environ.pop("MF_GPU_CAPACITY", None)


def test_placement_stays_out_of_the_way(tmp_path, monkeypatch):
    from mere_fusion_amd import placement
    monkeypatch.setenv("MF_PLACEMENT_DIR", str(tmp_path))
    monkeypatch.setenv("MF_GPUS", "8")
    monkeypatch.setenv("LOCAL_RANK", "3")                                        # torch.distributed.run numbers the processes itself (bench.py --gpus N)
    assert placement.ensure_placed() is None
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("MF_PLACEMENT", "0")
    assert placement.ensure_placed() is None
    monkeypatch.delenv("MF_PLACEMENT")
    monkeypatch.setenv("MF_GPUS", "1")
    assert placement.ensure_placed() is None and "HIP_VISIBLE_DEVICES" not in os.environ or os.environ.get("MF_PLACED_GPU") is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,6")                             # the deployment's own visibility is the set to choose from
    monkeypatch.delenv("MF_GPUS")
    assert placement.physical_gpus() == ["4", "6"]
