#!/usr/bin/env python3
"""bench.py -- lip-sync frames/s of the MI355X frame generator (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: the Wav2Lip generator
(wav2lip/models/wav2lip.py:87-125 as called at lipreal.py:124-125) on a batch of 16 mel chunks +
16 face crops already resident in HBM -- BASELINE.json configs[1].  One process per GPU; sessions
are independent so ranks share nothing but the barrier and the MAX of the elapsed time ("weak").
Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline     : the dominant HIP kernel's algorithmic TFLOP/s (HIP events around every launch,
                 mf_wav2lip_profile) against the dense MFMA peak of the arithmetic mode
  cpu_baseline : the oracle (torch fp32 restatement of the reference) timed on this box's host cores
  alt          : the same step in the other arithmetic mode, with its measured parity error
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mere_fusion_amd import _lib, harness, weights as W  # noqa: E402
from mere_fusion_amd.wav2lip.models import Wav2Lip  # noqa: E402

GFLOP_PER_FRAME = 7.934          # 2 x 3.967 GMAC, the 51 conv layers
BF16_DENSE_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
MFMA_PASSES = {"bf16": 1, "bf16x3": 3}


def build_model(precision, device):
    m = Wav2Lip(precision=precision)
    m.load_state_dict(W.make_wav2lip_state_dict(0))
    return m.to(device).eval()


class Runner:
    """Calls mf_wav2lip_forward (the C ABI the custom op wraps) on resident device buffers."""

    def __init__(self, precision, batch, device, seed=0):
        self.model = build_model(precision, device)
        mel, face, _ = W.make_lip_inputs(batch, seed)
        self.mel, self.face = mel.to(device), face.to(device)
        self.out = torch.empty((batch, 3, 96, 96), dtype=torch.float32, device=device)
        self.batch, self.device = batch, device
        self.h = self.model._ensure_handle(torch.device(device))
        self.lib = _lib.lib()
        self.stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def step(self):
        rc = self.lib.mf_wav2lip_forward(self.h, self.mel.data_ptr(), self.face.data_ptr(), self.out.data_ptr(),
                                         self.batch, self.stream)
        if rc:
            _lib.check(rc, "wav2lip_forward")

    def profile(self, iters):
        n = self.lib.mf_wav2lip_num_launches(self.h, self.batch)
        ms = (C.c_float * n)()
        _lib.check(self.lib.mf_wav2lip_profile(self.h, self.mel.data_ptr(), self.face.data_ptr(), self.out.data_ptr(),
                                               self.batch, iters, ms, self.stream), "wav2lip_profile")
        rows = []
        for i in range(n):
            name, kern, fl = C.create_string_buffer(96), C.create_string_buffer(96), C.c_double()
            _lib.check(self.lib.mf_wav2lip_launch_info(self.h, i, self.batch, name, 96, kern, 96, C.byref(fl)))
            rows.append(dict(layer=name.value.decode(), kernel=kern.value.decode(), flops=fl.value, ms=float(ms[i])))
        return rows


class MultiSession:
    """S independent talking-head sessions on one GPU, one hipStream + one generator handle each
    (BASELINE.json configs[3] per-GPU shape: lipreal.py runs one inference loop per session)."""

    def __init__(self, precision, batch, device, sessions):
        self.runners = []
        self.streams = [torch.cuda.Stream(device=device) for _ in range(sessions)]
        for i, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                self.runners.append(Runner(precision, batch, device, seed=100 + i))
        self.batch = batch

    def step(self):
        for r in self.runners:
            r.step()


def roofline(rows, precision):
    by = {}
    for r in rows:
        k = by.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, launches=0))
        k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["launches"] += 1
    dom = max(by, key=lambda k: by[k]["ms"])
    d = by[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    peak = BF16_DENSE_PEAK_TF / MFMA_PASSES[precision]
    total_ms = sum(r["ms"] for r in rows)
    return {
        "bound": "mfma", "kernel": dom, "launches_per_step": d["launches"],
        "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
        "alg_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 4),
        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None,
        "mfma_passes_per_product": MFMA_PASSES[precision],
        "kernel_share_of_step": round(d["ms"] / total_ms, 3),
        "sum_of_launches_ms": round(total_ms, 4),
    }, by


def parity_error(runner_model, batch=2):
    from oracle import wav2lip_ref
    mel, face, _ = W.make_lip_inputs(batch, 0)
    want = wav2lip_ref.wav2lip_forward(W.make_wav2lip_state_dict(0), mel, face)
    with torch.no_grad():
        got = runner_model(mel.cuda(), face.cuda()).cpu()
    return float((got - want).abs().max())


def host_threads(requested):
    """Threads for the CPU leg: the cores this process may actually run on (cgroup quota / affinity),
    not os.cpu_count() -- oversubscribing a quota-limited container makes torch crawl."""
    if requested > 0:
        return requested
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(batch, seconds, threads):
    """The oracle on the host cores: a reported baseline, not a target."""
    from oracle import wav2lip_ref
    sd = W.make_wav2lip_state_dict(0)
    mel, face, _ = W.make_lip_inputs(batch, 0)
    # pick the faster of {all available cores, half of them}: torch's conv oversubscribes quota-limited boxes
    cand = [host_threads(threads)] if threads > 0 else sorted({host_threads(0), max(1, host_threads(0) // 2)})
    best = None
    for nt in cand:
        torch.set_num_threads(nt)
        wav2lip_ref.wav2lip_forward(sd, mel, face)   # warm-up
        t0 = time.perf_counter()
        wav2lip_ref.wav2lip_forward(sd, mel, face)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    n, t0 = 0, time.perf_counter()
    while True:
        wav2lip_ref.wav2lip_forward(sd, mel, face)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds and n >= 2:
            break
    return {"value": round(n * batch / el, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} forwards of the fp32 oracle (oracle/wav2lip_ref.py) at batch {batch}, {el:.1f} s",
            "gflops": round(n * batch * GFLOP_PER_FRAME / el, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--precision", default=os.environ.get("MF_PRECISION", "bf16x3"), choices=sorted(MFMA_PASSES))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 skips the CPU baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = cores available to this process (capped at 64)")
    ap.add_argument("--profile-iters", type=int, default=10)
    ap.add_argument("--dump-layers", default=None, help="write the per-launch table (JSON) to this path")
    ap.add_argument("--sessions", type=int, default=8, help="concurrent sessions/GPU for the extra multi_session leg (0 = skip)")
    args = ap.parse_args()

    rank, local_rank, world = harness.init_dist("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to measure")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    run = Runner(args.precision, args.batch, device, seed=rank)
    elapsed = harness.timed_steps(run.step, args.steps, args.warmup, sync_fn=torch.cuda.synchronize, device=device)
    value = harness.aggregate_value(args.batch, args.steps, elapsed, world)
    ms_per_step = elapsed / args.steps * 1e3

    if rank == 0:
        line = {
            "metric": "lip-sync frames/sec", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "Wav2Lip generator 96x96, batch=16 mel-chunks per GPU, inputs resident in HBM "
                                   "(BASELINE.json configs[1]); seeded random-init weights",
                       "batch_per_gpu": args.batch, "sessions_at_25fps": round(value / 25.0, 1),
                       "parallelism": f"{world} independent replicas, sessions sharded by GPU, no collective"},
            "net_tflops": round(value * GFLOP_PER_FRAME / 1e3, 2),
        }
        rows = run.profile(args.profile_iters)
        rf, by = roofline(rows, args.precision)
        line["roofline"] = rf
        line["parity"] = {"linf_vs_oracle": parity_error(run.model), "tolerance": 1e-3 if args.precision == "bf16x3" else 8e-2}
        if args.dump_layers:
            with open(args.dump_layers, "w") as f:
                json.dump({"rows": rows, "by_kernel": by}, f, indent=1)
        if world == 1:
            other = "bf16" if args.precision == "bf16x3" else "bf16x3"
            alt = Runner(other, args.batch, device)
            el2 = harness.timed_steps(alt.step, max(args.steps // 2, 1), args.warmup, sync_fn=torch.cuda.synchronize)
            v2 = args.batch * max(args.steps // 2, 1) / el2
            rf2, _ = roofline(alt.profile(args.profile_iters), other)
            line["alt"] = {"dtype": other, "value": round(v2, 1), "unit": "frames/s",
                           "net_tflops": round(v2 * GFLOP_PER_FRAME / 1e3, 2),
                           "linf_vs_oracle": parity_error(alt.model),
                           "roofline": {k: rf2[k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_us")}}
            if args.sessions > 0:
                ms_ = MultiSession(args.precision, args.batch, device, args.sessions)
                el3 = harness.timed_steps(ms_.step, max(args.steps // 4, 1), 5, sync_fn=torch.cuda.synchronize)
                v3 = args.sessions * args.batch * max(args.steps // 4, 1) / el3
                big = Runner(args.precision, args.batch * args.sessions, device)
                el4 = harness.timed_steps(big.step, max(args.steps // 4, 1), 5, sync_fn=torch.cuda.synchronize)
                v4 = args.sessions * args.batch * max(args.steps // 4, 1) / el4
                line["multi_session"] = {
                    "sessions_per_gpu": args.sessions, "batch_per_session": args.batch,
                    "streams": {"value": round(v3, 1), "unit": "frames/s", "net_tflops": round(v3 * GFLOP_PER_FRAME / 1e3, 1),
                                "note": "one hipStream + handle per session, B=16 each (configs[3] per-GPU shape)"},
                    "cross_session_batch": {"value": round(v4, 1), "unit": "frames/s", "net_tflops": round(v4 * GFLOP_PER_FRAME / 1e3, 1),
                                            "note": f"one launch chain over {args.batch * args.sessions} frames"}}
                del ms_, big
            if args.cpu_seconds > 0:
                line["cpu_baseline"] = cpu_baseline(args.batch, args.cpu_seconds, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
