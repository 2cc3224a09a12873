#!/usr/bin/env python3
"""bench.py -- lip-sync frames/s of the MI355X frame generator (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

BASELINE.json's metric is "lip-sync frames/sec @256x256": the headline workload is configs[2], the MuseTalk
step of musereal.py:100-108 -- pe(audio) -> UNet(t=0) -> VAE.decode_latents -> uint8 256x256 BGR frames --
on a batch of 8 latents + 8 Whisper chunks already resident in HBM (the reference's own
"actual avg infer fps" brackets exactly this, musereal.py:99-115).  `--workload wav2lip` selects
configs[1] instead (Wav2Lip generator 96x96, batch 16), which is also reported in the default line
under "wav2lip".  One process per GPU; sessions are independent so ranks share nothing but the barrier and the
MAX of the elapsed time ("weak").
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

# Launch configurations come from the tuning table shipped beside the library (mere-fusion_amd/tune/gfx950.txt): the bench, its PMC child processes and a
# deployment all launch the same kernels.  Nothing in this script measures configurations (MF_AUTOTUNE is left unset).

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mere_fusion_amd import _lib, harness, weights as W  # noqa: E402
from mere_fusion_amd.wav2lip.models import Wav2Lip  # noqa: E402

GFLOP_PER_FRAME = 7.934          # 2 x 3.967 GMAC, the 51 conv layers
BF16_DENSE_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
MFMA_PASSES = {"bf16": 1, "bf16x3": 3}


def cpu_model():
    """`Model name` of the host CPU (north star: "core count stated" -- and which cores)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class PowerSampler:
    """Socket power and shader clock of one GPU while something runs: the amdgpu hwmon files of the device's PCI function (power1_input in uW,
    freq1_input = sclk in Hz, power1_cap), read by a thread every `period` seconds.  The f16 + FP6 conv kernel runs AT the board's power cap
    (profiles/r04_halo_sp_study.md): its rate is set by joules per output, and a TFLOP/s figure without the clock and the watts beside it misleads."""

    def __init__(self, device_index=0, period=0.02):
        import glob
        self.dir, self.period, self.rows = None, period, []
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                self.dir = hw[0]
        except Exception:
            self.dir = None

    def _read(self, name):
        try:
            return float(open(os.path.join(self.dir, name)).read())
        except (OSError, ValueError):
            return None

    def __enter__(self):
        import threading
        self.rows, self._stop = [], threading.Event()
        if self.dir is None:
            return self

        def loop():
            while not self._stop.is_set():
                w, f = self._read("power1_input"), self._read("freq1_input")
                if w is not None and f is not None:
                    self.rows.append((w * 1e-6, f * 1e-6))
                time.sleep(self.period)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.dir is not None:
            self._th.join(1.0)

    def report(self, skip_frac=0.25):
        """mean over the samples after the first `skip_frac` of the interval (the firmware needs a moment to settle)"""
        if self.dir is None or len(self.rows) < 4:
            return None
        r = self.rows[int(len(self.rows) * skip_frac):]
        cap = self._read("power1_cap")
        return {"socket_w": round(sum(x[0] for x in r) / len(r), 0), "socket_w_max": round(max(x[0] for x in r), 0), "cap_w": round(cap * 1e-6, 0) if cap else None,
                "sclk_mhz": round(sum(x[1] for x in r) / len(r), 0), "sclk_mhz_min": round(min(x[1] for x in r), 0), "samples": len(r)}


def build_model(precision, device):
    m = Wav2Lip(precision=precision)
    m.load_state_dict(W.make_wav2lip_state_dict(0))
    return m.to(device).eval()


class Runner:
    """lipreal.py:120-121 `pred = model(mel_batch, img_batch)` on resident device tensors, through the drop-in module (`wav2lip.models.Wav2Lip.forward` ->
    the torch.library op -> mf_wav2lip_forward), as the MuseTalk leg goes through its drop-in objects."""

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
        with torch.no_grad():
            self.out = self.model(self.mel, self.face)
        return self.out

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


class MuseTalkRunner:
    """musereal.py:100-108 on resident device tensors through the drop-in objects (C ABI underneath)."""

    def __init__(self, precision, batch, device, seed=0):
        from mere_fusion_amd.musetalk.models.unet import UNet
        from mere_fusion_amd.musetalk.models.vae import VAE
        from mere_fusion_amd.musetalk.config import MUSETALK_V1, unet_config_json, vae_config_json
        self.cfg = MUSETALK_V1
        self.usd = W.make_musetalk_unet_state_dict(self.cfg, 0)
        self.vsd = W.make_musetalk_vae_state_dict(self.cfg, 0)
        with torch.cuda.device(device):
            self.unet = UNet(unet_config_json(self.cfg["unet"]), self.usd, precision=precision, max_batch=batch)
            self.vae = VAE(config=vae_config_json(self.cfg["vae"]), state_dict=self.vsd, precision=precision, max_batch=batch)
        lat, aud = W.make_musetalk_inputs(batch, seed)
        self.lat_cpu, self.aud_cpu = lat, aud
        self.lat, self.aud = lat.to(device), aud.to(device)
        self.t0 = torch.tensor([0], device=device)
        self.batch = batch

    def step(self):
        pred = self.unet.model(self.lat, self.t0, encoder_hidden_states=self.unet.pe(self.aud)).sample
        return self.vae.decode_latents_device(pred)

    def step_d2h(self):
        """The reference's own "actual avg infer fps" bracket (musereal.py:99-115) ends after `vae.decode_latents`, whose last line copies the
        uint8 frames to the host (vae.py:105): the same step with that D2H + sync inside."""
        pred = self.unet.model(self.lat, self.t0, encoder_hidden_states=self.unet.pe(self.aud)).sample
        return self.vae.decode_latents(pred)

    def gflop_per_frame(self):
        from mere_fusion_amd.musetalk.config import algorithmic_flops_per_frame
        fu, fv = algorithmic_flops_per_frame(self.unet, self.vae)
        return (fu + fv) / 1e9

    def profile(self, iters):
        l = _lib.lib()
        rows = []
        for tag, h, nops, info, prof in (("unet", self.unet.model._h, l.mf_unet_num_ops, l.mf_unet_op_info, l.mf_unet_profile),
                                         ("vae", self.vae._h, l.mf_vae_num_ops, l.mf_vae_op_info, l.mf_vae_profile)):
            n = nops(h)
            ms = (C.c_float * n)()
            _lib.check(prof(h, self.batch, iters, ms, None), tag + "_profile")
            for i in range(n):
                nm, kn, fl = C.create_string_buffer(160), C.create_string_buffer(96), C.c_double()
                _lib.check(info(h, i, nm, 160, kn, 96, C.byref(fl)))
                rows.append(dict(layer=f"{tag}:{nm.value.decode()}", kernel=kn.value.decode(), flops=fl.value * self.batch, ms=float(ms[i])))
        return rows

    def parity(self):
        from oracle import musetalk_ref as R
        torch.set_num_threads(host_threads(0))
        want_u8, want_pred = R.musetalk_step(self.usd, self.vsd, self.cfg, self.lat_cpu[:1], self.aud_cpu[:1])
        pred = self.unet.model(self.lat[:1], self.t0, encoder_hidden_states=self.unet.pe(self.aud[:1])).sample
        got = self.vae.decode_latents(pred)
        d = np.abs(got.astype(int) - want_u8.astype(int))
        return {"latent_linf_vs_oracle": float((pred.cpu() - want_pred).abs().max()), "u8_max_diff": int(d.max()),
                "u8_differing_pixels": float((d > 0).mean()), "oracle": "parity unpinned (diffusers absent): oracle/musetalk_ref.py"}

    def cpu_baseline(self, seconds, threads):
        from oracle import musetalk_ref as R
        torch.set_num_threads(host_threads(threads))
        R.musetalk_step(self.usd, self.vsd, self.cfg, self.lat_cpu[:1], self.aud_cpu[:1])
        n, t0 = 0, time.perf_counter()
        while True:
            R.musetalk_step(self.usd, self.vsd, self.cfg, self.lat_cpu[:1], self.aud_cpu[:1])
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds and n >= 2:
                break
        return {"value": round(n / el, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
                "sample": f"{n} single-frame steps of the fp32 oracle (oracle/musetalk_ref.py: UNet + VAE decode), {el:.1f} s",
                "gflops": round(n * self.gflop_per_frame() / el, 1)}


def lean_runner(precision, batch, device):
    """The handles of a serving rank in its host-lean mode: created under MF_NO_GRAPH=2 every forward is a plain launch chain on the caller's stream -- no
    hipGraphLaunch and no cross-stream event wait, either of which keeps a ROCm 7.2 runtime thread spinning for as long as the GPU is busy (0.8 - 0.9 host core
    per process: tools/host_wait_probe2.py).  At 56 - 64 frames per step the ~450 eager launches are 2 % of the step; the graphs stay where a step is 8 frames."""
    old = os.environ.get("MF_NO_GRAPH")
    os.environ["MF_NO_GRAPH"] = "2"
    try:
        return MuseTalkRunner(precision, batch, device)
    finally:
        if old is None:
            os.environ.pop("MF_NO_GRAPH", None)
        else:
            os.environ["MF_NO_GRAPH"] = old


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


def lip_paced_streams(ms_, periods=5):
    """BASELINE.json configs[3] per-GPU shape as the reference runs it: every session on its own clock -- one batch of B mel chunks per B x 40 ms
    (lipreal.py:75-141 paced by the 25 fps audio, basereal) at a seeded random phase -- launched on ITS stream the moment it arrives.
    A batch's latency runs from its arrival to its frames being complete in HBM (event on the session's stream)."""
    S, B = len(ms_.runners), ms_.batch
    P = B * 0.040
    phase = np.random.default_rng(S).uniform(0.0, P, S)
    torch.cuda.synchronize()
    t_start = time.perf_counter() + 0.01
    nxt, issued, inflight, lats = [t_start + float(x) for x in phase], [0] * S, [], []
    while len(lats) < S * periods:
        now = time.perf_counter()
        for k in range(S):
            if issued[k] < periods and nxt[k] <= now:
                ms_.runners[k].step()
                ev = torch.cuda.Event()
                ev.record(ms_.streams[k])
                inflight.append((nxt[k], ev))
                nxt[k] += P
                issued[k] += 1
        still = []
        for t_arr, ev in inflight:
            if ev.query():
                lats.append(time.perf_counter() - t_arr)
            else:
                still.append((t_arr, ev))
        inflight = still
        if not inflight:
            due = [nxt[k] for k in range(S) if issued[k] < periods]
            dt = (min(due) if due else now) - time.perf_counter()
            if dt > 1e-3:
                time.sleep(dt - 5e-4)
    l = np.sort(np.asarray(lats)) * 1e3
    return {"sessions": S, "batch_per_session": B, "period_ms": round(P * 1e3), "batches": int(len(l)), "p50_ms": round(float(l[len(l) // 2]), 2),
            "p99_ms": round(float(l[min(len(l) - 1, int(np.ceil(0.99 * len(l))) - 1)]), 2), "max_ms": round(float(l[-1]), 2),
            "sustained": bool(l[-1] <= P * 1e3), "note": "arrival -> frames complete in HBM, one hipStream per session, launched on arrival (host polling included)"}


def roofline(rows, precision, only_mfma=False):
    by = {}
    for r in rows:
        if only_mfma and not r["kernel"].startswith("k_conv"):
            continue
        name, _, grid = r["kernel"].partition(" grid ")              # (" grid N": the launch's thread count, mf_conv_kernel_name)
        grid = grid.split()[0] if grid else grid                     # (" +gn" behind it: the GroupNorm-apply runs inside this conv)
        k = by.setdefault(name, dict(ms=0.0, flops=0.0, launches=0, grids=[]))
        k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["launches"] += 1
        if grid:
            k["grids"].append(int(grid))
    dom = max(by, key=lambda k: by[k]["ms"])
    d = by[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    # MFMA pass-equivalents per product of the DOMINANT kernel's operand format: 3 bf16 MFMAs (bf16x3), or one f16 MFMA + half a block-scaled FP6
    # MFMA at 4x the rate (the VAE decoder's resnet convs, "f16+fp6" in the kernel's name) = 1.5
    passes = 1.5 if "f16+fp6" in dom else MFMA_PASSES[precision]
    peak = BF16_DENSE_PEAK_TF / passes
    total_ms = sum(r["ms"] for r in rows)
    return {
        "bound": "mfma", "kernel": dom, "launches_per_step": d["launches"],
        "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
        "alg_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 4),
        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None,
        "frac_of_dense_f16_peak": round(achieved / BF16_DENSE_PEAK_TF, 4),      # against the chip's 2.5 PF, whatever the operand format (`peak` = 2.5 PF / passes)
        "mfma_passes_per_product": passes,
        "operand_format": "f16 + FP6 (e2m3, MX block scales) correction terms" if "f16+fp6" in dom else precision,
        "kernel_share_of_step": round(d["ms"] / total_ms, 3),
        "sum_of_launches_ms": round(total_ms, 4),
        "launch_grids": sorted(set(d["grids"])),
    }, by


def mfma_only_ceiling(precision):
    """What a kernel issuing nothing but matrix instructions sustains on this device (random operand bits), in TFLOP/s of the convolution's own
    arithmetic, for the shipped operand format and for the two fewer-pass formats DESIGN.md proposes (mf_probe_mfma_ceiling): the nominal peak of
    `roofline.peak` assumes 2.4 GHz, which an MFMA-dense kernel never holds."""
    l = _lib.lib()
    out = {}
    for mix, key in ((0, "bf16x3_12_bf16_mfma_per_128k"), (1, "f16_plus_2_mx_fp8"), (2, "f16_plus_2_mx_fp6"), (3, "f16_plus_2_mx_fp6_layer_data"),
                     (4, "bf16x3_12_bf16_mfma_per_128k_layer_data")):
        v = C.c_float()
        _lib.check(l.mf_probe_mfma_ceiling(mix, C.byref(v)), "probe_mfma_ceiling")
        out[key] = round(float(v.value), 1)
    if precision == "bf16":
        out["bf16_single_pass"] = round(3 * out["bf16x3_12_bf16_mfma_per_128k"], 1)
    return out


def roofline_kernel_power(batch, device, seconds=1.5):
    """The roofline kernel (f16 + FP6 halo conv) under SUSTAINED load: each of its three VAE grids launched back to back for ~`seconds` through the
    conv2d C ABI (mf_conv2d_time: hipEvents around N launches of the one kernel) while the package power and shader clock are sampled.  Inside a step
    the kernel alternates with memory-bound ones and sees a warmer power budget; this is the steady state the firmware converges to."""
    l = _lib.lib()
    rows = []
    rng = np.random.default_rng(0)
    for cin, hw in ((128, 256), (256, 128), (512, 64)):
        w = torch.from_numpy((rng.standard_normal((cin, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
        b = torch.zeros(cin)
        d = _lib.MfConv2dDesc(cin=cin, cout=cin, kh=3, kw=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1, transposed=0, output_padding=0, residual=0, act=0,
                              in_h=hw, in_w=hw, upsample=0)
        h = C.c_void_p()
        _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS["f16q"], C.byref(h)))
        x = torch.randn(batch, cin, hw, hw, device=device)
        y = torch.empty(batch, cin, hw, hw, device=device)
        for _ in range(2):
            _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), batch, None))
        t = C.c_float()
        _lib.check(l.mf_conv2d_time(h, batch, 20, C.byref(t), None))
        iters = max(int(seconds / max(t.value * 1e-3, 1e-5)), 50)
        with PowerSampler(torch.device(device).index or 0) as ps:
            _lib.check(l.mf_conv2d_time(h, batch, iters, C.byref(t), None))
        gf = 2.0 * batch * hw * hw * cin * cin * 9 / 1e9
        row = {"layer": f"{cin} -> {cin} @{hw}^2, batch {batch}", "launch_us": round(t.value * 1e3, 1), "algorithmic_tflops": round(gf / t.value, 1), "launches": iters}
        pw = ps.report()
        if pw:
            row.update(pw)
        rows.append(row)
        l.mf_conv2d_destroy(h)
        del x, y
    torch.cuda.empty_cache()
    return {"rows": rows, "note": "the kernel alone, launched back to back (steady state); socket power and shader clock from the amdgpu hwmon files, mean over the last 75 % of "
                                   "the interval.  At the cap the firmware lowers the clock until the package fits: the kernel's rate is joules per output, not issue slots "
                                   "(profiles/r04_halo_sp_study.md)"}


def pmc_traffic(kernel, workload, precision, batch_args, want_clock=True, grids=None):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC counters, collected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE
    and WRITE_SIZE in SEPARATE passes (they do not fit one), --kernel-trace only beside --pmc, values in KiB; on gfx950 FETCH_SIZE
    tallies 128-byte requests of wide coalesced reads at 64 bytes, so the read side is doubled.  WRITE_SIZE is uncalibrated (guide).
    A third pass reads GRBM_GUI_ACTIVE: effective clock = busy cycles / dispatch wall time (the guide's DVFS note) -- MFMA-dense kernels run
    power-capped well under the 2.4 GHz the nominal peak assumes.  Each pass re-runs this script for 2 steps in a child process."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", None
    want = "void" + kernel.split(" f16+fp6")[0].replace(" ", "")[:-1]                # "voidk_conv_igemm<256,256,2,4,true,32" (+ ",<ring depth>>" or ">")
    if "f16+fp6" in kernel:
        want += ",-1"                                                                # the plain 3x3 instantiation (PHASE = -1), not the four upsample-phase ones
    vals, clock = {}, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE") + (("GRBM_GUI_ACTIVE",) if want_clock else ()):
        d = tempfile.mkdtemp(prefix="mf_pmc_")
        cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--workload", workload, "--precision", precision, "--steps", "2", "--warmup", "1", "--extras", "0", "--cpu-seconds", "0",
               "--profile-iters", "0", "--pmc-traffic", "0"] + batch_args
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
            tot, n, ns = 0.0, 0, 0.0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    kn = r.get("Kernel_Name", "").replace(" ", "")
                    if grids and int(float(r.get("Grid_Size", 0))) not in grids:
                        continue                                     # (the same kernel symbol also runs the channel-slice-split launches: other grids)
                    if r.get("Counter_Name") == ctr and kn.startswith(want) and kn[len(want):len(want) + 1] in (",", ">"):
                        tot += float(r["Counter_Value"]); n += 1
                        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                            ns += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if n == 0:
                if ctr == "GRBM_GUI_ACTIVE":
                    continue
                return None, f"no {ctr} rows for {kernel}", None
            if ctr == "GRBM_GUI_ACTIVE":
                if ns > 0:
                    ghz = tot / ns                                   # cycles per ns
                    if ghz > 4.0:                                    # (summed over the 8 XCDs)
                        ghz /= 8.0
                    clock = round(ghz, 3)
            else:
                vals[ctr] = tot / n * 1024.0
        except Exception as e:   # measurement leg only: the bench line is still valid without it
            if ctr == "GRBM_GUI_ACTIVE":
                continue
            return None, f"{ctr} pass failed: {type(e).__name__}", None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    note = "2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate rocprofv3 --pmc passes, per launch"
    if grids:
        note += f"; averaged over the launches `avg_launch_us` averages (grids {sorted(grids)})"
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], note,
            clock)


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
    return {"value": round(n * batch / el, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{n} forwards of the fp32 oracle (oracle/wav2lip_ref.py) at batch {batch}, {el:.1f} s",
            "gflops": round(n * batch * GFLOP_PER_FRAME / el, 1)}


def wav2lip_config0(args, device, threads):
    """BASELINE.json configs[0] / SURVEY 8d cfg 1: ONE session streaming at batch 1 -- one face crop, a 3 s wav, fps 50, l = r = 10 (lipasr.py:14-37 ->
    lipreal.py:75-141): the context primed with 20 silent chunks, then per video frame two new 20 ms chunks -> the (2 + 20) x 320 = 7040-sample window ->
    mel (80, 36) -> the one 16-column chunk at column 16 -> generator at B = 1 -> frame x 255 -> uint8.  75 frames.  CPU leg: the oracle (mel_ref, glue_ref,
    wav2lip_ref).  GPU leg: the same loop through the drop-in objects (mf_melspec -> chunking -> k_faces_u8 -> generator -> k_head), one device sync per frame as a
    session's loop has (the frame goes to the consumer before the next chunk pair arrives): per-frame latency, not throughput."""
    from oracle import wav2lip_ref, mel_ref, glue_ref
    from mere_fusion_amd.lip_driver import LipASRFrontend, LipSession
    rng = np.random.default_rng(0)
    face = rng.integers(0, 256, (1, 96, 96, 3), dtype=np.uint8)
    wav = np.clip(0.1 * rng.standard_normal(48000), -1.0, 1.0).astype(np.float32)
    chunks = [wav[i * 320:(i + 1) * 320] for i in range(150)]
    sd = W.make_wav2lip_state_dict(0)
    n_frames = 75
    # ---- CPU: the restatement of the reference loop --------------------------------------------------------------------------
    torch.set_num_threads(threads)
    ctx = [np.zeros(320, np.float32) for _ in range(20)]
    cpu_u8, lat_c = [], []
    for it in range(n_frames + 1):                                    # (first iteration = warm-up, not timed)
        k = max(it - 1, 0)
        t0 = time.perf_counter()
        fr = ctx + chunks[2 * k:2 * k + 2]
        mel = mel_ref.melspectrogram(np.concatenate(fr))
        ml, _ = glue_ref.mel_chunks(mel, len(fr), 10, 10, 50)
        img, melb = glue_ref.face_batch(face, ml)
        pred = wav2lip_ref.wav2lip_forward(sd, torch.from_numpy(melb), torch.from_numpy(img))
        u8 = glue_ref.to_uint8(glue_ref.frames_from_pred(pred.numpy()))
        dt = time.perf_counter() - t0
        if it > 0:
            lat_c.append(dt)
            cpu_u8.append(u8[0])
            ctx = fr[-20:]
    # ---- GPU: the same session through the drop-in path -----------------------------------------------------------------------
    model = build_model(args.precision, device)
    fe = LipASRFrontend(1, fps=50, stride_left=10, stride_right=10, device=device)
    sess = LipSession(model, face)
    lat_g, gpu_u8 = [], []
    for rep_ in range(2):                                             # (first pass = warm-up: graph capture at B = 1, first-touch)
        fe.warm_up()
        sess.index = 0
        lat_g, gpu_u8 = [], []
        for k in range(n_frames):
            t0 = time.perf_counter()
            mb = fe.run_step(chunks[2 * k:2 * k + 2])
            frames, _ = sess.step(mb)
            u8 = frames.to(torch.uint8).cpu().numpy()                 # process_frames' astype(uint8) (truncation, lipreal.py:211) + the copy to the host
            lat_g.append(time.perf_counter() - t0)
            gpu_u8.append(u8[0])
    d = np.abs(np.stack(gpu_u8).astype(int) - np.stack(cpu_u8).astype(int))
    lc, lg = np.sort(np.asarray(lat_c)) * 1e3, np.sort(np.asarray(lat_g)) * 1e3
    del model, sess, fe
    return {"workload": "Wav2Lip 96x96, 1 face crop + 3 s wav, batch=1 streaming, fps 50, l = r = 10, 75 frames (BASELINE.json configs[0]; lipasr.py:14-37 -> lipreal.py:75-141)",
            "cpu_baseline": {"value": round(n_frames / float(np.sum(lat_c)), 2), "unit": "frames/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
                             "ms_per_frame_p50": round(float(lc[len(lc) // 2]), 2), "sample": f"{n_frames} frames: mel of the 7040-sample window + chunk + B = 1 forward of the fp32 oracle + uint8"},
            "gpu": {"value": round(n_frames / float(np.sum(lat_g)), 1), "unit": "frames/s", "ms_per_frame_p50": round(float(lg[len(lg) // 2]), 3),
                    "ms_per_frame_p99": round(float(lg[min(len(lg) - 1, int(np.ceil(0.99 * len(lg))) - 1)]), 3), "dtype": args.precision,
                    "note": "host PCM chunks in -> uint8 frame on the host out, one sync per frame (latency of ONE session; a real-time session needs 25 frames/s)"},
            "parity": {"u8_max_diff_vs_cpu_oracle": int(d.max()), "u8_differing_fraction": float((d > 0).mean()),
                       "note": "truncating uint8 conversion: a value within 1e-5 of an integer may land on either side"}}


def wav2lip_report(args, device, world, rank, value=None, ms_per_step=None, run=None):
    """The configs[1] leg (Wav2Lip generator, B=16).  Headline when --workload wav2lip, else an extra object."""
    run = run or Runner(args.precision, args.w2l_batch, device, seed=rank)
    if value is None:
        steps = args.steps if args.workload == "wav2lip" else 200
        el = harness.timed_steps(run.step, steps, 20, sync_fn=torch.cuda.synchronize)
        value, ms_per_step = args.w2l_batch * steps / el, el / steps * 1e3
    out = {"value": round(value, 1), "unit": "frames/s", "ms_per_step": round(ms_per_step, 4), "dtype": args.precision,
           "workload": "Wav2Lip generator 96x96, batch=16 mel-chunks, inputs resident in HBM (BASELINE.json configs[1])",
           "net_tflops": round(value / world * GFLOP_PER_FRAME / 1e3, 2)}
    if args.profile_iters <= 0:            # child of a PMC pass: the timed steps are all it needs to run
        return out
    rows = run.profile(args.profile_iters)
    rf, by = roofline(rows, args.precision)
    if world == 1 and args.pmc_traffic and bool(args.extras):
        rf["traffic"], rf["traffic_note"], _ = pmc_traffic(rf["kernel"], "wav2lip", args.precision, ["--w2l-batch", str(args.w2l_batch)], want_clock=False)
        if rf["traffic"]:
            rf["traffic"] = round(rf["traffic"])
            rf["traffic_gbytes_per_s"] = round(rf["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9, 1)
    out["roofline"] = rf
    out["parity"] = {"linf_vs_oracle": parity_error(run.model), "tolerance": 1e-3 if args.precision == "bf16x3" else 8e-2,
                     "oracle": "pinned to the reference (tests/golden/wav2lip_golden.npz)"}
    if args.dump_layers:
        with open(args.dump_layers, "w") as f:
            json.dump({"rows": rows, "by_kernel": by}, f, indent=1)
    if world == 1:
        other = "bf16" if args.precision == "bf16x3" else "bf16x3"
        alt = Runner(other, args.w2l_batch, device)
        el2 = harness.timed_steps(alt.step, 100, 20, sync_fn=torch.cuda.synchronize)
        v2 = args.w2l_batch * 100 / el2
        rf2, _ = roofline(alt.profile(args.profile_iters), other)
        out["alt"] = {"dtype": other, "value": round(v2, 1), "unit": "frames/s", "net_tflops": round(v2 * GFLOP_PER_FRAME / 1e3, 2),
                      "linf_vs_oracle": parity_error(alt.model),
                      "roofline": {k: rf2[k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_us")}}
        del alt
        if args.sessions > 0:
            ms_ = MultiSession(args.precision, args.w2l_batch, device, args.sessions)
            el3 = harness.timed_steps(ms_.step, 50, 5, sync_fn=torch.cuda.synchronize)
            v3 = args.sessions * args.w2l_batch * 50 / el3
            paced = lip_paced_streams(ms_) if getattr(args, "paced", 1) else None
            big = Runner(args.precision, args.w2l_batch * args.sessions, device)
            el4 = harness.timed_steps(big.step, 50, 5, sync_fn=torch.cuda.synchronize)
            v4 = args.sessions * args.w2l_batch * 50 / el4
            out["multi_session"] = {
                "sessions_per_gpu": args.sessions, "batch_per_session": args.w2l_batch,
                "streams": {"value": round(v3, 1), "unit": "frames/s", "note": "one hipStream + handle per session (configs[3] per-GPU shape)"},
                "cross_session_batch": {"value": round(v4, 1), "unit": "frames/s", "note": f"one launch chain over {args.w2l_batch * args.sessions} frames"}}
            if paced:
                out["multi_session"]["paced_25fps"] = paced
            del ms_, big
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.w2l_batch, min(args.cpu_seconds, 8.0), args.cpu_threads)
            out["config0"] = wav2lip_config0(args, device, out["cpu_baseline"]["cores"])
    return out


def muse_multi_session(args, device):
    """The north star's 8 sessions per GPU through the session driver (mere_fusion_amd/muse_driver.py): one UNet / VAE pair, every step
    gathers each session's mirror-indexed cached latents + its Whisper chunks into ONE sessions x batch frame step (musereal.py:91-108 for all
    sessions at once).  Second number: the same step with the paste-back of musereal.py:238-247 done on the device into 720p frames."""
    from mere_fusion_amd import muse_driver as D
    from mere_fusion_amd.paste import AvatarFrames
    S, B = args.sessions, args.batch
    big = MuseTalkRunner(args.precision, S * B, device)
    rng = np.random.default_rng(0)
    H_, W_, n = 720, 1280, 25
    sessions, chunks = [], []
    for s in range(S):
        lats = W.make_musetalk_inputs(n, 500 + s)[0]
        boxes = [(500 + i, 200 + i, 500 + i + 260, 200 + i + 270) for i in range(n)]
        crops = [(b[0] - 26, b[1] - 27, b[2] + 26, b[3] + 27) for b in boxes]
        masks = [np.repeat(rng.integers(0, 256, (c[3] - c[1], c[2] - c[0], 1), dtype=np.uint8), 3, axis=2) for c in crops]
        av = AvatarFrames(torch.randint(0, 256, (n, H_, W_, 3), dtype=torch.uint8, device=device), boxes, masks, crops, device=device)
        sessions.append(D.MuseSession(lats, avatar_frames=av))
        chunks.append(W.make_musetalk_inputs(B, 700 + s)[1].to(device))
    rep = {"sessions_per_step": S, "batch_per_session": B, "driver": "mere_fusion_amd.muse_driver.MuseBatcher", "unit": "frames/s"}
    for key, paste in (("value", False), ("with_gpu_paste_back_720p", True)):
        bat = D.MuseBatcher(big.unet, big.vae, sessions, batch_size=B, paste=paste, device=device)
        el = harness.timed_steps(lambda: bat.step(chunks), 5, 2, sync_fn=torch.cuda.synchronize)
        rep[key] = round(S * B * 5 / el, 1)
        if not paste:
            rep["ms_per_step"] = round(el / 5 * 1e3, 3)
            rep["sessions_at_25fps"] = round(S * B * 5 / el / 25.0, 1)
            rep["fps_per_session"] = round(B * 5 / el, 1)
    rows_b = big.profile(2)
    cb = [r for r in rows_b if r["layer"].startswith("unet:") and r["flops"] > 0 and "attention" not in r["layer"]]
    tb, fb = sum(r["ms"] for r in cb), sum(r["flops"] for r in cb)
    rep["unet_conv_blocks"] = {"achieved_tflops": round(fb / (tb * 1e-3) / 1e12, 1),
                               "mfma_issue_frac_of_bf16_peak": round(MFMA_PASSES[args.precision] * fb / (tb * 1e-3) / 1e12 / BF16_DENSE_PEAK_TF, 3)}
    del big, bat, sessions
    torch.cuda.empty_cache()
    if getattr(args, "paced", 1):
        _stage("paced sessions")
        lean = lean_runner(args.precision, S * B, device)            # the serving rank's handles: eager, single stream (host-lean)
        rep["paced_sessions"] = muse_paced_sessions(lean, args, device, rep["value"], full=bool(getattr(args, "full", 0)), lean=True)
        rep["paced_sessions"]["rank_mode"] = "host-lean: handles under MF_NO_GRAPH=2 (eager launch chain, no side branches), single-stream scheduler"
        if getattr(args, "full", 0):
            # the same search with graph-replayed handles and the copy / Whisper side streams (round 4's rank): what the host-lean mode costs in sessions
            _stage("paced sessions, graph-mode handles")
            big2 = MuseTalkRunner(args.precision, S * B, device)
            g = muse_paced_sessions(big2, args, device, rep["value"], full=False, lean=False)
            rep["paced_sessions"]["graph_mode_rank"] = {"max_sessions_sustained": g.get("max_sessions_sustained"), "at_max": g["end_to_end"]["at_max"]}
            del big2
        del lean
        torch.cuda.empty_cache()
    return rep


def thread_cpu_times():
    """{tid: (comm, utime + stime seconds)} of this process's threads (/proc/self/task): who spends the host CPU of a serving rank"""
    out = {}
    try:
        tck = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                st = open(f"/proc/self/task/{tid}/stat").read()
                comm = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[int(tid)] = (comm, (int(f[11]) + int(f[12])) / tck)
            except (OSError, ValueError, IndexError):
                pass
    except (OSError, ValueError):
        pass
    return out


class PacedRig:
    """What the real-time legs share: the warmed-up UNet / VAE pair, one Whisper encoder, and per session an avatar (25 cached latents + 25 cached
    720p frames with paste geometry on the device), an endless 16 kHz PCM stream cut into 20 ms chunks, a MuseASRFrontend and a FrameRing of 2B slots
    (musereal.py:153) -- built once for the largest N a search can reach and reused by every trial."""

    def __init__(self, big, args, device, n_max, lean=False):
        from mere_fusion_amd import muse_driver as D
        self.lean = bool(lean)          # handles created under MF_NO_GRAPH=2 + a single-stream scheduler: no hipGraphLaunch, no cross-stream wait anywhere
        from mere_fusion_amd.paste import AvatarFrames
        from mere_fusion_amd.transport import FrameRing
        from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature
        self.D, self.big, self.device, self.S, self.B, self.n_max = D, big, device, args.sessions, args.batch, n_max
        self.P = self.B * 0.040
        self.lat25 = W.make_musetalk_inputs(25, 4242)[0]
        self.chunk = W.make_musetalk_inputs(self.B, 4243)[1].to(device)
        self.a2f = Audio2Feature(state_dict=W.make_whisper_encoder_state_dict(0), n_head=6, precision=args.precision, device=device)
        rng = np.random.default_rng(0)
        H_, W_, n = 720, 1280, 25
        boxes = [(500 + i, 200 + i, 500 + i + 260, 200 + i + 270) for i in range(n)]
        crops = [(b[0] - 26, b[1] - 27, b[2] + 26, b[3] + 27) for b in boxes]
        masks = [np.repeat(rng.integers(0, 256, (c[3] - c[1], c[2] - c[0], 1), dtype=np.uint8), 3, axis=2) for c in crops]
        self.avatars = [AvatarFrames(torch.randint(0, 256, (n, H_, W_, 3), dtype=torch.uint8, device=device), boxes, masks, crops, device=device)
                        for _ in range(n_max)]
        self.pcm = [W.make_speech_like_wav(2 * self.B * 320 * 32, 1000 + k) for k in range(min(n_max, 8))]       # 32 batches of audio per stream, looped
        self.rings_full = [FrameRing(2 * self.B, (H_, W_, 3)) for _ in range(n_max)]
        self.rings_face = None
        self.FrameRing = FrameRing
        # every step size the schedulers can issue: eager forward, graph capture (MuseBatcher.prewarm), and the Whisper encoder at every window count
        t_w = time.perf_counter()
        warm = D.MuseBatcher(big.unet, big.vae, [D.MuseSession(self.lat25) for _ in range(self.S)], batch_size=self.B, device=device)
        warm.prewarm()
        nwin = (2 * self.B + 20) * 320
        for k in range(self.S, 0, -1):
            self.a2f.audio2feat_windows_device(torch.zeros((k, nwin), device=device))
        self.step_ms = {}
        for k in range(1, self.S + 1):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(3):
                warm.step([self.chunk] * self.S, only=range(k))
            torch.cuda.synchronize(device)
            self.step_ms[str(k)] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
        self.warm_s = time.perf_counter() - t_w

    def chunks_of(self, k, j):
        """the 2B 20 ms chunks of session k's j-th batch"""
        wav = self.pcm[k % len(self.pcm)]
        n = 2 * self.B * 320
        o = (j * n) % (len(wav) - n + 1)
        return [wav[o + i * 320:o + (i + 1) * 320] for i in range(2 * self.B)]

    def close(self):
        for r in self.rings_full + (self.rings_face or []):
            r.close()

    def trial(self, N, seconds, stages=("whisper", "paste", "ring")):
        """N sessions in real time for `seconds`: every session completes one batch of 2B PCM chunks per B x 40 ms at a seeded random phase.
        stages: () = the round-2 measurement (precomputed Whisper chunks in HBM -> UNet -> VAE, frames complete in HBM); "whisper": the chunks
        come from each session's PCM through its front-end and ONE encoder call per step; "paste": 720p paste-back on the device; "ring": the
        frames leave through the session's FrameRing and the latency ends when the consumer thread HOLDS the batch's last tuple."""
        import threading
        D, B, P, dev = self.D, self.B, self.P, self.device
        periods = max(int(round(seconds / P)), 4)
        use_w, use_p, use_r = "whisper" in stages, "paste" in stages, "ring" in stages
        sessions = [D.MuseSession(self.lat25, avatar_frames=self.avatars[k] if use_p else None) for k in range(N)]
        bat = D.MuseBatcher(self.big.unet, self.big.vae, sessions, batch_size=B, paste=use_p, device=dev, max_sessions_per_step=self.S)
        rings = None
        if use_r:
            if use_p:
                rings = self.rings_full[:N]
            else:
                if self.rings_face is None or len(self.rings_face) < N:
                    for r in self.rings_face or []:
                        r.close()
                    self.rings_face = [self.FrameRing(2 * B, (256, 256, 3)) for _ in range(self.n_max)]
                rings = self.rings_face[:N]
        if use_w or use_r:
            fes = [D.MuseASRFrontend(self.a2f, B) for _ in range(N)]
            for fe in fes:
                fe.warm_up()
            sch = D.EndToEndScheduler(bat, fes, self.a2f, rings=rings, period_s=P, fixed_chunks=None if use_w else self.chunk,
                                      single_stream=self.lean)
        else:
            sch = D.SessionScheduler(bat, period_s=P)
        # the consumers (`process_frames`, lipreal.py:195 / musereal.py:226): one thread per session BLOCKED in its ring's get(timeout), as the reference's are
        # in `res_frame_queue.get(block=True, timeout=1)`; a batch is delivered when its last tuple is held.  (Round 4 polled all rings from one thread every
        # millisecond: 22 k get_nowait()s per second were a quarter of the rank's host CPU.)
        got_t = [[] for _ in range(N)]
        stop = threading.Event()
        cons_tids = set()

        def drain(k):
            import queue as _q
            cons_tids.add(threading.get_native_id())
            cnt = 0
            while not stop.is_set():
                try:
                    f, idx, au = rings[k].get(block=True, timeout=0.05, copy=False)
                except _q.Empty:
                    continue
                if f is not None:
                    rings[k].release()
                cnt += 1
                if cnt % B == 0:
                    got_t[k].append(time.perf_counter())
        ths = []
        if use_r:
            ths = [threading.Thread(target=drain, args=(k,), daemon=True) for k in range(N)]
            for th in ths:
                th.start()
        import resource
        phase = np.random.default_rng(N).uniform(0.0, P, N)
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        th0, main_tid = thread_cpu_times(), threading.get_native_id()
        t_start = time.perf_counter() + 0.01
        nxt = [t_start + float(ph) for ph in phase]
        issued, arrivals, lats, total = [0] * N, [[] for _ in range(N)], [], N * periods
        n_done = 0
        while n_done < total:
            now = time.perf_counter()
            for k in range(N):
                while issued[k] < periods and nxt[k] <= now:
                    if use_w or use_r:
                        sch.submit(k, self.chunks_of(k, issued[k]), nxt[k])
                    else:
                        sch.submit(k, self.chunk, nxt[k])
                    arrivals[k].append(nxt[k])
                    nxt[k] += P
                    issued[k] += 1
            done = sch.run_once(now)
            if done:
                n_done += len(done)
                if not use_r:
                    lats.extend(d[3] for d in done)
                continue
            due = [t for t in ([nxt[k] for k in range(N) if issued[k] < periods] + [sch.next_due()]) if t is not None]
            dt = (min(due) if due else now + 0.05) - time.perf_counter()
            if hasattr(sch, "idle_wait"):
                sch.idle_wait(dt - 2e-4 if dt > 1e-3 else dt)               # woken by the completion of a step in flight (blocking-sync event on a waiter thread) or at the next due time
            elif dt > 1e-3:
                time.sleep(dt - 5e-4)
            elif dt > 0:
                time.sleep(dt)
        wall = time.perf_counter() - t_start
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        host_cpu = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / wall     # scheduler loop + consumer threads + runtime threads of this process
        th1 = thread_cpu_times()
        by = {}
        for tid, (comm, sec) in th1.items():
            d = sec - th0.get(tid, (comm, 0.0))[1]
            waiter = getattr(getattr(sch, "_waiter", None), "native_id", None)
            key = ("scheduler loop (main thread)" if tid == main_tid else "consumer threads" if tid in cons_tids else "completion waiter thread" if tid == waiter
                   else f"{comm} (tid {tid}, runtime / other)")
            by[key] = by.get(key, 0.0) + d
        host_by_thread = {k: round(v / wall, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:6] if v / wall >= 0.005}
        if use_r:
            t_end = time.perf_counter() + 2.0
            while any(len(got_t[k]) < periods for k in range(N)) and time.perf_counter() < t_end:
                time.sleep(0.001)
            stop.set()
            for th in ths:
                th.join(2.0)
            # by arrival order over all sessions, so that first / last thirds mean what they mean in the other legs
            pairs = sorted((arrivals[k][j], got_t[k][j] - arrivals[k][j]) for k in range(N) for j in range(min(len(got_t[k]), periods)))
            lats = [l for _, l in pairs]
        if hasattr(sch, "close"):
            sch.close()
        l = np.sort(np.asarray(lats)) * 1e3
        p99 = float(l[min(len(l) - 1, int(np.ceil(0.99 * len(l))) - 1)])
        n3 = max(len(lats) // 3, 1)                                          # served first / last: a queue that grows shows up as a drift between them
        first, third = np.asarray(lats[:n3]) * 1e3, np.asarray(lats[-n3:]) * 1e3
        drift = float(third.mean() - first.mean())
        return {"sessions": N, "stages": list(stages), "seconds": round(wall, 1), "batches": int(len(l)), "first_third_mean_ms": round(float(first.mean()), 1),
                "last_third_mean_ms": round(float(third.mean()), 1), "all_mean_ms": round(float(l.mean()), 1), "p50_ms": round(float(l[len(l) // 2]), 1),
                "p99_ms": round(p99, 1), "max_ms": round(float(l[-1]), 1),
                "sustained": bool(len(l) == total and p99 <= P * 1e3 and drift <= 0.1 * P * 1e3),
                "sessions_per_step_mean": round(sch.sessions_served / max(sch.steps, 1), 2),
                "gpu_busy_frac": round(sch.busy_s / wall, 3), "frames_per_s": round(N * periods * B / wall, 1),
                "host_cpu_s_per_wall_s": round(host_cpu, 3), "host_cpu_by_thread": host_by_thread}

    def search(self, cap, stages, screen_s, confirm_s, fail_s):
        """Largest N whose trial holds the bound: short screening trials walk from cap + 1 (which should fail) to the first N that holds, then ONE long
        trial confirms it (and steps down if the long run disagrees) and a medium one shows N + 1 failing."""
        trials, best = [], None
        N = cap + 1
        r = self.trial(N, screen_s, stages); trials.append(r)
        if r["sustained"]:
            while r["sustained"] and N < self.n_max:
                N += 1
                r = self.trial(N, screen_s, stages); trials.append(r)
            N = N - 1 if not r["sustained"] else N
        else:
            while not r["sustained"] and N > 1:
                N -= 1
                r = self.trial(N, screen_s, stages); trials.append(r)
        while N >= 1:
            r = self.trial(N, confirm_s, stages); trials.append(r)
            if r["sustained"]:
                best = r
                break
            N -= 1
        if best is not None and N + 1 <= self.n_max and fail_s > 0:
            trials.append(self.trial(N + 1, fail_s, stages))
        return best, trials


def muse_paced_sessions(big, args, device, free_fps, full=True, lean=False):
    """BASELINE.json's second metric -- "max concurrent >= 25 fps sessions" -- measured instead of extrapolated (a session is
    sustained when the p99 latency of its B-frame batches is <= B x 40 ms).  N sessions run on their own clocks in real time; the per-GPU
    scheduler (mere_fusion_amd.muse_driver) packs whoever waits into steps of up to `--sessions` sessions on the one UNet / VAE pair.
    `end_to_end`: the whole session loop inside the measurement -- each session's 20 ms PCM chunks -> its ASR front-end, the
    Whisper encoder once per step for all picked sessions, UNet + VAE, 720p paste-back on the device, frames out through the session's FrameRing;
    latency = arrival of the batch's last audio chunk -> the consumer thread holds the batch's last (res_frame, idx, audio_frames) tuple.
    `unet_vae_only`: the round-2 measurement (precomputed Whisper chunks resident in HBM, frames complete in HBM).  `stages`: what each stage costs,
    at the N the end-to-end loop sustains and one above."""
    S, B = args.sessions, args.batch
    P = B * 0.040
    cap = max(int(free_fps / 25.0), 1)
    rig = PacedRig(big, args, device, n_max=cap + 4, lean=lean)
    try:
        # default run: 3 s screening trials + ONE 15 s confirmation (the whole bench stays under ~4 minutes); --full 1: the 40 s confirmation, N + 1 shown failing
        # over 20 s, the UNet + VAE only search and the per-stage trials (round 4's measurement, ~150 s)
        long_ = bool(getattr(args, "full", 0))
        e2e, e2e_trials = rig.search(cap, ("whisper", "paste", "ring"), screen_s=4.0 if long_ else 3.0, confirm_s=40.0 if long_ else 15.0, fail_s=20.0 if long_ else 0.0)
        rep = {"criterion": f"p99 latency of a session's {B}-frame batch <= {B} x 40 ms = {P * 1e3:.0f} ms, every batch delivered, and no queue growth (mean latency of the "
                            f"last third of the batches - of the first third <= {P * 100:.0f} ms); real time, seeded random phases, one batch per session per {P * 1e3:.0f} ms",
               "scheduler": f"mere_fusion_amd.muse_driver.EndToEndScheduler / SessionScheduler: oldest first, <= {S} sessions per step, a partly filled step waits <= {P * 250:.0f} ms; "
                            "frames copied out on a second stream behind the step",
               "end_to_end": {"latency": "arrival of the batch's last 20 ms PCM chunk -> the consumer thread holds the batch's last (res_frame, idx, audio_frames) tuple",
                              "stages": "museasr.py:15-29 front-end + one Whisper encoder call per step, musereal.py:91-108 UNet + VAE, :238-247 paste-back into 720p frames on the device, "
                                        ":116,153 FrameRing hand-off (D2H of the 720p frames)",
                              "max_sessions_sustained": e2e["sessions"] if e2e else None, "at_max": e2e, "trials": e2e_trials},
               "max_sessions_sustained": e2e["sessions"] if e2e else None,
               "step_ms_by_sessions_in_step": rig.step_ms, "warmup_s": round(rig.warm_s, 1)}
        if full:
            uv, uv_trials = rig.search(cap, (), screen_s=4.0, confirm_s=20.0, fail_s=0.0)
            rep["unet_vae_only"] = {"latency": "arrival of the batch's Whisper chunks (already in HBM) -> uint8 256 x 256 frames complete in HBM (the round-2 measurement)",
                                    "max_sessions_sustained": uv["sessions"] if uv else None, "at_max": uv, "trials": uv_trials}
            if e2e:
                n0 = e2e["sessions"]
                rep["stages"] = {"note": f"5 s trials at N = {n0} (what the end-to-end loop sustains) and N = {n0 + 1}: which stage costs the next session",
                                 "rows": [rig.trial(n, 5.0, st) for st in ((), ("whisper",), ("whisper", "paste"), ("whisper", "ring")) for n in (n0, n0 + 1) if n <= rig.n_max]}
    finally:
        rig.close()
    return rep


def muse_node_rank_leg(args, device):
    """One rank's share of the node metric at --gpus N > 1: the 8-sessions-per-step handles, the free-running 8 x 8 rate (where the search starts), then the
    end-to-end paced search -- every rank runs this at the same time on its own GPU (host cores and PCIe are shared, as on a serving node)."""
    from mere_fusion_amd import muse_driver as D
    S, B = args.sessions, args.batch
    big = lean_runner(args.precision, S * B, device)
    bat = D.MuseBatcher(big.unet, big.vae, [D.MuseSession(W.make_musetalk_inputs(25, 500 + s)[0]) for s in range(S)], batch_size=B, device=device)
    chunks = [W.make_musetalk_inputs(B, 700 + s)[1].to(device) for s in range(S)]
    bat.prewarm()
    el = harness.timed_steps(lambda: bat.step(chunks), 5, 2, sync_fn=lambda: torch.cuda.synchronize(device), device=device)
    rep = muse_paced_sessions(big, args, device, S * B * 5 / el, full=False, lean=True)
    rep["free_running_8x8_frames_per_s"] = round(S * B * 5 / el, 1)
    del big
    torch.cuda.empty_cache()
    return rep


def ranks_on_one_gpu_worker(args, device):
    """One of R processes sharing GPU 0 (bench.py --ranks-on-one-gpu R spawns them): its own UNet / VAE / Whisper handles, `--sessions` end-to-end sessions in real
    time for --worker-seconds, started together with the other workers (ready files in --worker-dir).  Prints one JSON line."""
    import resource
    S, B = args.sessions, args.batch
    big = lean_runner(args.precision, S * B, device)
    rig = PacedRig(big, args, device, n_max=S, lean=True)
    k, R, d = args.worker_rank, args.ranks_on_one_gpu, args.worker_dir
    open(os.path.join(d, f"ready_{k}"), "w").close()
    t_wait = time.perf_counter()
    while len([f for f in os.listdir(d) if f.startswith("ready_")]) < R and time.perf_counter() - t_wait < 600:
        time.sleep(0.05)
    try:
        r = rig.trial(S, args.worker_seconds)
    finally:
        rig.close()
    ru = resource.getrusage(resource.RUSAGE_SELF)
    r.update({"rank": k, "process_cpu_s_total": round(ru.ru_utime + ru.ru_stime, 1), "warmup_s": round(rig.warm_s, 1)})
    print("WORKER " + json.dumps(r), flush=True)


def ranks_on_one_gpu(args, R, sessions_per_rank, seconds):
    """is the HOST side of an 8-rank node viable?  R processes share THIS GPU (each with its own handles, scheduler, consumer thread and rings,
    exactly what one rank of `--gpus 8` runs), each driving `sessions_per_rank` end-to-end sessions in real time at the same time; the GPU is time-shared, so the
    sessions are chosen to fit ONE GPU in total.  Reported: every rank's p99 and its host CPU seconds per wall second (process + threads)."""
    import subprocess, tempfile, shutil
    d = tempfile.mkdtemp(prefix="mf_ranks_")
    cmd = [sys.executable, os.path.abspath(__file__), "--ranks-on-one-gpu", str(R), "--worker-dir", d, "--worker-seconds", str(seconds), "--sessions", str(sessions_per_rank),
           "--batch", str(args.batch), "--precision", args.precision]
    procs = [subprocess.Popen(cmd + ["--worker-rank", str(k)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(R)]
    rows, t0 = [], time.perf_counter()
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            out = ""
        for ln in out.splitlines():
            if ln.startswith("WORKER "):
                rows.append(json.loads(ln[7:]))
    shutil.rmtree(d, ignore_errors=True)
    rows.sort(key=lambda r: r["rank"])
    ok = len(rows) == R
    rep = {"ranks": R, "sessions_per_rank": sessions_per_rank, "seconds": seconds, "wall_s": round(time.perf_counter() - t0, 1), "host_cores_available": host_threads(0),
           "cpu_model": cpu_model(), "latency_bound_ms": args.batch * 40,
           "what": "R processes sharing ONE GPU, each the full per-rank stack of --gpus R (own UNet / VAE / Whisper handles, EndToEndScheduler, consumer thread, FrameRings), "
                   "all trials started together; end-to-end latency as in multi_session.paced_sessions.end_to_end"}
    if ok:
        rep.update({"worst_rank_p99_ms": max(r["p99_ms"] for r in rows), "all_sustained": all(r["sustained"] for r in rows),
                    "host_cpu_s_per_wall_s_per_rank": [r["host_cpu_s_per_wall_s"] for r in rows],
                    "host_cpu_s_per_wall_s_total": round(sum(r["host_cpu_s_per_wall_s"] for r in rows), 2),
                    "frames_per_s_total": round(sum(r["frames_per_s"] for r in rows), 1), "per_rank": rows})
    else:
        rep["error"] = f"{len(rows)} of {R} workers reported"
        rep["per_rank"] = rows
    return rep


def _transport_producer(q, n, shape, batched):
    frames = np.random.default_rng(0).integers(0, 256, (8,) + shape, dtype=np.uint8)
    audio = [(np.zeros(320, np.float32), 0)] * 16
    if batched:
        for i in range(0, n, 8):
            q.put_batch(frames, list(range(i, i + 8)), audio)
    else:
        for i in range(n):
            q.put((frames[i % 8], i, audio[:2]))
    q.put((None, -1, []))


def transport_report(args, device):
    """the reference's `res_frame_queue` (pickled (frame, idx, audio_frames) tuples through mp.Queue, musereal.py:116,153)
    against the shared-memory FrameRing behind the same tuple contract, one producer process -> this process, 256 x 256 x 3 uint8 frames;
    and the device -> ring leg of one batch (page-locked slots, one pitched DMA)."""
    import multiprocessing as mp
    from mere_fusion_amd.transport import FrameRing
    ctx = mp.get_context("spawn")
    shape, n = (256, 256, 3), 2000
    rep = {"frame_bytes": int(np.prod(shape)), "frames": n, "unit": "frames/s (one producer process -> one consumer)"}
    for name, make, batched in (("mp_queue_pickled", lambda: ctx.Queue(16), False), ("frame_ring", lambda: FrameRing(16, shape, ctx=ctx), False),
                                ("frame_ring_put_batch", lambda: FrameRing(16, shape, ctx=ctx), True)):
        q = make()
        p = ctx.Process(target=_transport_producer, args=(q, n, shape, batched))
        p.start()
        q.get(timeout=120)                                            # first item: the producer is up
        t0, got = time.perf_counter(), 1
        while True:
            f, idx, _ = q.get(timeout=60)
            if idx == -1:
                break
            got += 1
        rep[name] = round((got - 1) / (time.perf_counter() - t0), 0)
        p.join(30)
        if hasattr(q, "close") and isinstance(q, FrameRing):
            q.close()
    ring = FrameRing(16, shape, ctx=ctx)
    frames = torch.randint(0, 256, (args.batch,) + shape, dtype=torch.uint8, device=device)
    audio = [(np.zeros(320, np.float32), 0)] * (2 * args.batch)

    def one():
        ring.put_batch(frames, list(range(args.batch)), audio)
        for _ in range(args.batch):
            ring.get(timeout=5, copy=False)
            ring.release()
    for _ in range(3):
        one()
    # per-batch times, median reported: round 5's first runs showed 1.7 - 2.2 ms as the MEAN of 50 where every probe of the same path gave 0.11 ms -- one stall of
    # ~80 ms somewhere in the 50 (the producer children of the legs above being reaped, a collection) is a property of this process, not of the hand-off
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    rep["device_batch_to_ring_and_out_ms"] = round(ts[len(ts) // 2] * 1e3, 3)
    rep["device_batch_to_ring_and_out_ms_mean_max"] = [round(sum(ts) / len(ts) * 1e3, 3), round(ts[-1] * 1e3, 3)]
    t0 = time.perf_counter()
    for _ in range(50):
        frames.cpu().numpy()                                          # vae.py:105
    rep["device_batch_cpu_numpy_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 3)
    # the reference's hand-off of the same batch, like for like: vae.py:105 `.cpu().numpy()`, then one pickled tuple per frame through mp.Queue
    # (musereal.py:116-119) and the consumer's get() (musereal.py:226)
    q = ctx.Queue(2 * args.batch)

    def ref_one():
        host = frames.cpu().numpy()
        for i in range(args.batch):
            q.put((host[i], i, audio[2 * i:2 * i + 2]))
        for _ in range(args.batch):
            q.get(timeout=5)
    for _ in range(3):
        ref_one()
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        ref_one()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    rep["device_batch_reference_handoff_ms"] = round(ts[len(ts) // 2] * 1e3, 3)
    rep["note"] = ("device_batch_*: one batch of B uint8 256 x 256 frames from HBM to the consumer's hands in this process: FrameRing.put_batch (one DMA into page-locked "
                   "slots, one descriptor message) + B get()s, against the reference's `.cpu().numpy()` + B pickled mp.Queue puts + B gets; "
                   "device_batch_cpu_numpy_ms is the bare copy with no hand-off at all")
    ring.close()
    return rep


def whisper_report(args, device):
    """H3 for the same batch: MuseASR.run_step's audio2feat on the B = 8 window ((2B + l + r) * 320 = 11520 samples -> feat (36, 5, 384),
    museasr.py:22-27) -- the reference pads every window to 30 s and runs the whole 1500-token encoder (transcribe.py:108).
    Reported per window: the literal evaluation, the exact one without the unconsumed work (one window, and 8 sessions' windows in one
    call), and a shortened context with its measured error (never the default: the encoder's attention is global)."""
    from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature
    a2f = Audio2Feature(state_dict=W.make_whisper_encoder_state_dict(0), n_head=6, precision=args.precision, device=device)
    n = (2 * args.batch + 20) * 320
    S = max(args.sessions, 1)
    wavs = torch.stack([torch.from_numpy(W.make_speech_like_wav(n, i)) for i in range(S)]).to(device)
    rep = {"samples_per_window": n, "feature_rows_per_window": n // 320, "unit": "ms per window"}

    def timed(f, windows):
        el = harness.timed_steps(f, 20, 3, sync_fn=torch.cuda.synchronize)
        return round(el / 20 * 1e3 / windows, 3)
    a2f.set_mode("exact_full")
    ref = a2f.audio2feat_windows_device(wavs[:1]).clone()
    rep["exact_full_1500_tokens"] = timed(lambda: a2f.audio2feat_windows_device(wavs[:1]), 1)
    a2f.set_mode("exact")
    rep["exact_pruned"] = timed(lambda: a2f.audio2feat_windows_device(wavs[:1]), 1)
    rep["exact_pruned_linf_vs_full"] = float((a2f.audio2feat_windows_device(wavs[:1]) - ref).abs().max())
    if S > 1:
        rep[f"exact_pruned_{S}_windows_per_call"] = timed(lambda: a2f.audio2feat_windows_device(wavs), S)
    for ctx in (128, 512):
        a2f.set_mode("windowed", ctx)
        rep[f"windowed_{ctx}_tokens"] = {"ms": timed(lambda: a2f.audio2feat_windows_device(wavs[:1]), 1),
                                         "linf_vs_exact": float((a2f.audio2feat_windows_device(wavs[:1]) - ref).abs().max()),
                                         "feature_abs_max": float(ref.abs().max())}
    return rep


class ErNeRFRunner:
    """configs[4]: one 512 x 512 head frame per step -- near/far, march -> field -> composite until no ray is alive
    (renderer.py:231-291), synthetic ball-shaped occupancy grid, seeded field weights, rays resident in HBM."""

    def __init__(self, precision, width, device, seed=0):
        from mere_fusion_amd.ernerf.field import HipNeRFField, grid_geometry
        from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
        self.width = width
        offsets, pls = grid_geometry()
        sd = W.make_ernerf_field_state_dict(int(offsets[-1]), seed)
        self.sd = {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}
        self.offsets, self.S = offsets, float(np.log2(pls))
        self.bitfield = W.make_ernerf_sphere_bitfield()
        ro, rd = W.make_ernerf_camera_rays(width)
        self.ro_h, self.rd_h = ro, rd
        self.ro, self.rd = torch.from_numpy(ro).to(device), torch.from_numpy(rd).to(device)
        g = torch.Generator().manual_seed(seed)
        self.enc_a, self.ind, self.eye = torch.randn(1, 32, generator=g), torch.randn(1, 4, generator=g) * 0.1, torch.tensor([[0.4]])
        self.d_enc_a, self.d_ind, self.d_eye = self.enc_a.to(device), self.ind.to(device), self.eye.to(device)
        self.precision = precision
        self.field = HipNeRFField(self.sd, precision=precision, max_samples=width * width, device=device)
        # the full nerfreal.py frame: audio window -> enc_a, torso over the background, head loop, uint8 frame
        from mere_fusion_amd.ernerf.audio import HipAudioEncoder
        from mere_fusion_amd.ernerf.torso import HipTorso
        t_offs, _ = grid_geometry(num_levels=16, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048)
        self.torso_sd = W.make_ernerf_torso_state_dict(int(t_offs[-1]), seed)
        self.torso = HipTorso(self.torso_sd, precision=precision, max_pixels=width * width, device=device)
        tmpl = {"audio_net.encoder_conv.0.weight": torch.empty(32, 44, 3), "audio_net.encoder_conv.0.bias": torch.empty(32),
                "audio_net.encoder_conv.2.weight": torch.empty(32, 32, 3), "audio_net.encoder_conv.2.bias": torch.empty(32),
                "audio_net.encoder_conv.4.weight": torch.empty(64, 32, 3), "audio_net.encoder_conv.4.bias": torch.empty(64),
                "audio_net.encoder_conv.6.weight": torch.empty(64, 64, 3), "audio_net.encoder_conv.6.bias": torch.empty(64),
                "audio_net.encoder_fc1.0.weight": torch.empty(64, 64), "audio_net.encoder_fc1.0.bias": torch.empty(64),
                "audio_net.encoder_fc1.2.weight": torch.empty(32, 64), "audio_net.encoder_fc1.2.bias": torch.empty(32),
                "audio_att_net.attentionNet.0.weight": torch.empty(8, 8), "audio_att_net.attentionNet.0.bias": torch.empty(8)}
        for i, (ci, co) in enumerate(((32, 16), (16, 8), (8, 4), (4, 2), (2, 1))):
            tmpl[f"audio_att_net.attentionConvNet.{2 * i}.weight"] = torch.empty(co, ci, 3)
            tmpl[f"audio_att_net.attentionConvNet.{2 * i}.bias"] = torch.empty(co)
        self.audio_sd = W.make_ernerf_audio_state_dict(tmpl, seed)
        self.audio = HipAudioEncoder(self.audio_sd, att=2, device=device)
        self.auds = torch.randn(8, 44, 16, generator=g).to(device)
        u = (torch.arange(width, dtype=torch.float32) + 0.5) / width * 2 - 1
        yy, xx = torch.meshgrid(u, u, indexing="ij")
        self.bg_coords = torch.stack([xx, yy], -1).reshape(-1, 2).to(device)
        self.pose = torch.eye(4)[None]
        self.r = HipHeadRenderer(self.field, torch.from_numpy(self.bitfield).to(device), density_scale=40.0, torso=self.torso, audio=self.audio,
                                 ind_code=self.d_ind, smooth_lips=True)
        self.last, self.trace, self.loop = None, None, os.environ.get("MF_NERF_LOOP", "device")

    def step(self):
        # MF_NERF_LOOP=host keeps the reference's host-synced compaction (renderer.py:266); the default runs the round control on the device
        if self.loop == "host":
            self.last = self.r.render(self.ro, self.rd, self.auds, self.bg_coords, self.pose, self.d_eye, bg_color=1.0, want_u8=True)
            self.trace = self.last["trace"]
        else:
            self.last = self.r.render(self.ro, self.rd, self.auds, self.bg_coords, self.pose, self.eye, bg_color=1.0, want_u8=True, loop="device")

    def launched_rounds(self):
        """rounds of the next frame that go out as (march, field, composite) launches; one tail launch stands for the other max_steps - that many
        (mf_nerf_head_render follows the round counts the frames before posted)"""
        if getattr(self.r, "_head", None) is None:
            return None
        k = C.c_int(0)
        _lib.check(_lib.lib().mf_nerf_head_plan_rounds(self.r._head, 16, C.byref(k)), "mf_nerf_head_plan_rounds")
        return {"as_launches": k.value, "tail_launch_covers": 16 - k.value}

    def roofline(self, iters=5):
        """Per-kernel rates of one frame's head loop, measured live: the reference-shaped host loop (renderer.run_cuda) enqueues every kernel
        from Python on torch's current stream, so torch.cuda.Event pairs on that stream bracket each one.  MFMA kernel: the fused field,
        46,368 algorithmic FLOP per sample against the bf16 dense peak / MFMA passes.  HBM kernels: the algorithmic bytes
        per unit of DESIGN.md section 4 against 8 TB/s."""
        from mere_fusion_amd.ernerf import _raymarching_face as rm
        r, N = self.r, self.width * self.width
        acc = {}

        def timed(name, units, fn, *a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = fn(*a); e1.record()
            acc.setdefault(name, []).append((e0, e1, units))
            return out
        orig = (rm.near_far_from_aabb, rm.march_rays, rm.composite_rays_triplane, r.field.forward)
        try:
            rm.near_far_from_aabb = lambda *a: timed("k_near_far_from_aabb", 32.0 * a[3], orig[0], *a)
            rm.march_rays = lambda *a: timed("k_march_rays", a[0] * (28.0 + 32.0 * a[1]), orig[1], *a)
            rm.composite_rays_triplane = lambda *a: timed("k_composite_rays_triplane", a[0] * (56.0 + 36.0 * a[1]), orig[2], *a)
            r.field.forward = lambda x, *a: timed("k_nerf_field_fused", 46368.0 * x.shape[0], orig[3], x, *a)
            for _ in range(iters + 1):
                r.run_cuda(self.ro, self.rd, r.enc_a if r.enc_a is not None else self.d_enc_a, self.d_ind, self.d_eye, bg_color=1.0)
        finally:
            rm.near_far_from_aabb, rm.march_rays, rm.composite_rays_triplane = orig[:3]
            r.field.forward = orig[3]
        torch.cuda.synchronize()
        rows = {}
        for name, evs in acc.items():
            per = len(evs) // (iters + 1)
            evs = evs[per:]                                    # drop the warm-up frame
            ms = sum(a.elapsed_time(b) for a, b, _ in evs) / iters
            rows[name] = {"launches_per_frame": per, "ms_per_frame": round(ms, 4), "units": sum(u for _, _, u in evs) / iters}
        peak_tf = BF16_DENSE_PEAK_TF / MFMA_PASSES[self.precision]
        f = rows["k_nerf_field_fused"]
        tf = f["units"] / (f["ms_per_frame"] * 1e-3) / 1e12
        out = {"bound": "mfma", "kernel": "k_nerf_field_fused", "launches_per_frame": f["launches_per_frame"],
               "avg_launch_us": round(f["ms_per_frame"] / f["launches_per_frame"] * 1e3, 2), "alg_flop_per_sample": 46368,
               "achieved": round(tf, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4), "traffic": None,
               "note": "host-synced loop of renderer.run_cuda (one launch per kernel per round); the bench's device loop runs the same field kernel",
               "hbm_kernels": {}}
        for name in ("k_near_far_from_aabb", "k_march_rays", "k_composite_rays_triplane"):
            k = rows[name]
            gbs = k["units"] / (k["ms_per_frame"] * 1e-3) / 1e9
            out["hbm_kernels"][name] = {"launches_per_frame": k["launches_per_frame"], "ms_per_frame": k["ms_per_frame"],
                                        "alg_bytes_per_frame": int(k["units"]), "achieved_gbytes_per_s": round(gbs, 1), "frac_of_8tbs": round(gbs / 8000.0, 4)}
        return out

    def through_dropin(self, frames=30):
        """The same frame through the DROP-IN seam: a module with the reference NeRFNetwork's attributes and state-dict names and
        `HipRenderMixin` in front -- what `from ernerf.nerf_triplane.network import NeRFNetwork` resolves to -- rendered by `model.render(...)` with the arguments
        `Trainer.test_step` passes (utils.py:949-950): device tensors for the eye feature, `index`, `**vars(opt)`.  One host sync per frame, as the reference has it
        (`test_gui_with_data` copies the frame to the host right away).  (The reference's own class needs its checkout, which the GPU box lacks;
        tests/test_dropin_ernerf.py checks the import chain against it in the build container.)"""
        import argparse
        from mere_fusion_amd.ernerf.network import HipRenderMixin
        dev = self.ro.device
        full = {**self.sd, **self.torso_sd, **self.audio_sd}
        runner = self

        class _RefShaped(torch.nn.Module):                    # renderer.py:62-133 / network.py:93-165: the attributes the render path reads
            def __init__(s, opt):
                super().__init__()
                s.opt, s.bound, s.grid_size, s.density_scale, s.min_near = opt, 1, 128, 40.0, 0.05
                s.exp_eye, s.test_train, s.smooth_lips, s.torso, s.train_camera, s.emb, s.att = True, False, True, True, False, False, 2
                s.individual_dim, s.individual_dim_torso, s.density_thresh_torso, s.mean_density_torso = 4, 8, 0.01, 0.0
                s.enc_a, s._n = None, {}
                s.individual_codes = torch.nn.Parameter(runner.ind.expand(16, 4).contiguous().clone(), requires_grad=False)
                s.register_buffer("density_bitfield", torch.from_numpy(runner.bitfield).clone())
                for k, v in full.items():
                    n = "p_" + k.replace(".", "__")
                    s.register_parameter(n, torch.nn.Parameter(v.clone().float(), requires_grad=False))
                    s._n[n] = k

            def state_dict(s, *a, **k):
                return {s._n.get(key, key): v for key, v in super().state_dict(*a, **k).items()}

            def run_cuda(s, *a, **k):
                raise RuntimeError("the reference's run_cuda was reached")

            def render(s, rays_o, rays_d, auds, bg_coords, poses, staged=False, max_ray_batch=4096, **kw):      # renderer.py:657-677
                return s.run_cuda(rays_o, rays_d, auds, bg_coords, poses, **kw)

        class _Net(HipRenderMixin, _RefShaped):
            pass
        m = _Net(argparse.Namespace(torso_shrink=0.8)).to(dev).eval()
        kw = dict(eye=self.d_eye, index=[0], staged=True, bg_color=None, perturb=False, dt_gamma=1 / 256, max_steps=16, T_thresh=1e-4, torso_shrink=0.8)
        args = (self.ro[None], self.rd[None], self.auds, self.bg_coords[None], self.pose.to(dev))
        import gc
        for _ in range(3):
            m.render(*args, **kw)
        torch.cuda.synchronize()
        gc.collect()                                          # (handles of earlier legs die here, not inside the timed frames: their hipFree synchronises the device)
        per = []
        for _ in range(frames):
            t0 = time.perf_counter()
            out = m.render(*args, **kw)
            torch.cuda.synchronize()                          # the reference syncs here: outputs['image'] goes to the host (utils.py:1211)
            per.append(time.perf_counter() - t0)
        dt = float(np.median(per))
        rep = {"frames_per_s": round(1.0 / dt, 1), "ms_per_frame": round(dt * 1e3, 3), "ms_per_frame_mean": round(float(np.mean(per)) * 1e3, 3),
               "device_loop_frames": int(m.mf_frames), "image_shape": list(out["image"].shape),
               "note": "model.render(...) through HipRenderMixin with the arguments Trainer.test_step passes; one host sync per frame (as Trainer.test_gui_with_data has)"}
        # The WHOLE per-frame sequence of nerfreal.py:70-111 around that render: the loader's get_rays (provider.py:302 -> utils.py:255-341), model.render,
        # test_gui_with_data's resize + device -> host copies (utils.py:1208-1212), the uint8 conversion of nerfreal.py:111 -- once with the reference's operations
        # (restated: the GPU box has no reference checkout) and once with the drop-in's `utils` module (mere_fusion_amd/ernerf/frontend.py: same results)
        import torch.nn.functional as F
        from mere_fusion_amd.ernerf import frontend as fe
        Wd = self.width
        intr = np.array([Wd / 0.7, Wd / 0.7, Wd / 2, Wd / 2])          # the camera of weights.make_ernerf_camera_rays as (fx, fy, cx, cy) + the pose below

        def ref_get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, rect=None):
            device, B = poses.device, poses.shape[0]
            fx, fy, cx, cy = intrinsics
            i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device), indexing="ij")
            i = i.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
            j = j.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
            inds = torch.arange(H * W, device=device).expand([B, H * W])
            zs = torch.ones_like(i)
            directions = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
            directions = directions / torch.norm(directions, dim=-1, keepdim=True)
            rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
            return {"i": i, "j": j, "inds": inds, "rays_o": poses[..., :3, 3][..., None, :].expand_as(rays_d), "rays_d": rays_d}

        class _Model:
            def eval(s):
                pass

        class _Tr(fe.TrainerMixin):
            def __init__(s):
                s.model, s.ema, s.fp16, s.opt = _Model(), None, True, argparse.Namespace(color_space="srgb")

            def test_step(s, data, perturb=False):
                o = m.render(data["rays_o"], data["rays_d"], args[2], args[3], data["poses"], **kw)
                return o["image"].reshape(-1, Wd, Wd, 3), o["depth"].reshape(-1, Wd, Wd)
        tr = _Tr()
        pose_h = torch.eye(4)[None].clone()
        pose_h[0, :3, 3] = torch.tensor([0.02, -0.01, -2.2])
        pose_d = pose_h.to(dev)                                         # the loader keeps its poses on the device (provider.py:253) and indexes them per frame

        def frame_reference_style():
            poses = pose_d[[0]]
            r = ref_get_rays(poses, intr, Wd, Wd)
            with torch.no_grad():
                with torch.cuda.amp.autocast(enabled=True):
                    preds, depth = tr.test_step({"rays_o": r["rays_o"], "rays_d": r["rays_d"], "poses": poses})
            preds = F.interpolate(preds.permute(0, 3, 1, 2), size=(Wd, Wd), mode="bilinear").permute(0, 2, 3, 1).contiguous()
            depth = F.interpolate(depth.unsqueeze(1), size=(Wd, Wd), mode="nearest").squeeze(1)
            img, _ = preds[0].detach().cpu().numpy(), depth[0].detach().cpu().numpy()
            return (img * 255).astype(np.uint8)

        def frame_dropin():
            poses = pose_d[[0]]
            r = fe.get_rays(ref_get_rays, poses, intr, Wd, Wd)
            o = tr.test_gui_with_data({"rays_o": r["rays_o"], "rays_d": r["rays_d"], "poses": poses}, Wd, Wd)
            return (o["image"] * 255).astype(np.uint8)
        loop = {}
        for name, fn in (("reference_operations", frame_reference_style), ("dropin_utils", frame_dropin)):
            for _ in range(3):
                last = fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(frames):
                t0 = time.perf_counter()
                last = fn()
                ts.append(time.perf_counter() - t0)
            loop[name] = {"ms_per_frame": round(float(np.median(ts)) * 1e3, 3), "frames_per_s": round(1.0 / float(np.median(ts)), 1)}
            loop[name + "_u8_mean"] = round(float(last.mean()), 3)
        loop["same_frame"] = bool(loop.pop("reference_operations_u8_mean") == loop.pop("dropin_utils_u8_mean"))
        loop["note"] = ("pose -> get_rays -> model.render -> resize + device -> host -> uint8, as nerfreal.py:70-111 runs it per frame: with the reference's operations "
                        "around the render, and with the drop-in's ernerf.nerf_triplane.utils (rays cached per (H, W, intrinsics), pinned copies, one sync)")
        rep["whole_frame_loop"] = loop
        return rep

    def samples_per_frame(self):
        if self.trace is None:        # the device loop keeps no host-side trace: count the same frame once through the host loop
            self.trace = self.r.run_cuda(self.ro, self.rd, self.r.enc_a, self.d_ind, self.d_eye, bg_color=1.0)["trace"]
        return sum(a * s for a, s in self.trace)

    def parity(self, width=48):
        from mere_fusion_amd.ernerf.field import HipNeRFField
        from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
        from oracle import ernerf_render_ref as RR
        ro, rd = W.make_ernerf_camera_rays(width)
        dev = self.ro.device
        r = HipHeadRenderer(self.field, torch.from_numpy(self.bitfield).to(dev), density_scale=40.0)
        got = r.run_cuda(torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev), self.d_enc_a, self.d_ind, self.d_eye, bg_color=1.0)
        want = RR.run_cuda(self.sd, self.offsets, self.S, ro, rd, self.enc_a, self.ind, self.eye, self.bitfield, bg_color=1.0, density_scale=40.0)
        err = np.abs(got["image"].cpu().numpy() - want["image"]).max(1)
        return {"image_linf_max_vs_oracle": float(err.max()), "rays": width * width,
                "oracle": "oracle/ernerf_render_ref.py, pinned to the reference Python above the extension boundary (tests/golden/make_ernerf_golden.py); CUDA kernels unpinned"}

    def cpu_baseline(self, seconds, threads, width=64):
        from oracle import ernerf_render_ref as RR
        torch.set_num_threads(threads)
        ro, rd = W.make_ernerf_camera_rays(width)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds or n == 0:
            RR.run_cuda(self.sd, self.offsets, self.S, ro, rd, self.enc_a, self.ind, self.eye, self.bitfield, bg_color=1.0, density_scale=40.0)
            n += 1
        dt = (time.perf_counter() - t0) / n
        scale = (self.width * self.width) / (width * width)
        return {"value": round(1.0 / (dt * scale), 3), "unit": "frames/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
                "sample": f"{n} frames of {width}x{width} rays through the C / torch restatement, scaled by ray count to {self.width}x{self.width}"}


def wav2vec2_report(args, device):
    """NerfASR's per-step network call (nerfasr.py:105-143): one (l + m + r) x 20 ms window = 8960 samples through the XLSR-53-large CTC network
    every m = 8 video-rate steps (160 ms of audio); 1 window and 8 sessions' windows per call; logits against transformers on the CPU."""
    from mere_fusion_amd.ernerf.asr import HipWav2Vec2ForCTC
    from oracle import wav2vec2_ref as R
    cfg = W.WAV2VEC2_XLSR_LARGE
    sd = W.make_wav2vec2_state_dict(cfg, 0)
    wav = np.stack([W.make_speech_like_wav(8960, s) for s in range(8)])
    m = HipWav2Vec2ForCTC(cfg, sd, max_windows=8, precision=args.precision, device=device)
    x1, x8 = torch.from_numpy(wav[:1]).to(device), torch.from_numpy(wav).to(device)
    rep = {"network": "Wav2Vec2ForCTC, XLSR-53 large (24 x 1024, 315 M parameters), 44 symbols; seeded random-init weights", "samples_per_window": 8960,
           "audio_ms_per_call": 160, "unit": "ms per call"}
    for tag, x in (("one_window", x1), ("eight_windows_per_call", x8)):
        el = harness.timed_steps(lambda: m(x), 20, 3, sync_fn=torch.cuda.synchronize)
        rep[tag] = round(el / 20 * 1e3, 3)
    t0 = time.perf_counter()
    torch.set_num_threads(host_threads(args.cpu_threads))
    ref_model = R.build(cfg, sd)
    want = R.frame_to_logits(ref_model, wav[:1])
    t1 = time.perf_counter()
    for _ in range(3):
        R.frame_to_logits(ref_model, wav[:1])
    rep["cpu_transformers_ms_per_window"] = round((time.perf_counter() - t1) / 3 * 1e3, 1)
    rep["logits_linf_vs_transformers"] = float(np.abs(m(x1).logits.cpu().numpy() - want).max())
    del m
    torch.cuda.empty_cache()
    return rep


def ernerf_report(args, device, world, rank, value=None, ms_per_step=None, run=None):
    """The configs[4] leg (ER-NeRF 512x512).  Headline when --workload ernerf, else an extra object."""
    if run is None:
        run = ErNeRFRunner(args.precision, 512, device, seed=rank)
        steps = 200                                           # (0.1 s: 20 frames left the leg with +- 2 % of run-to-run noise)
        el = harness.timed_steps(run.step, steps, 10, sync_fn=torch.cuda.synchronize)
        value, ms_per_step = steps / el, el / steps * 1e3
    smp = run.samples_per_frame()
    rep = {"workload": "ER-NeRF frame 512x512 (nerfreal.py path): audio nets -> enc_a, torso deform / colour nets over the background, near/far + "
                       "(march -> tri-plane field -> composite) x <= 16 steps, uint8 frame; synthetic occupancy, rays resident in HBM "
                       "(BASELINE.json configs[4])",
           "value": round(value, 1), "unit": "frames/s", "ms_per_step": round(ms_per_step, 3), "dtype": args.precision,
           "samples_per_frame": int(smp), "march_iterations": len(run.trace), "loop": run.loop, "launched_rounds": run.launched_rounds(),
           "field_tflops_algorithmic": round(smp * 46368 * value / max(world, 1) / 1e12, 2)}
    rep["parity"] = run.parity()
    if getattr(args, "extras", 1):
        rep["asr_frontend"] = wav2vec2_report(args, device)
        rep["through_dropin"] = run.through_dropin()
    if args.profile_iters > 0:
        rep["roofline"] = run.roofline(args.profile_iters)
    if args.cpu_seconds > 0:
        rep["cpu_baseline"] = run.cpu_baseline(min(args.cpu_seconds, 10.0), host_threads(args.cpu_threads))
    return rep


_T0 = time.perf_counter()

# ---- the ONE stdout line --------------------------------------------------------------------------------------------------------------
# The driver keeps only the tail of stdout and parses its last line: round 4's line had grown to 24 KB and did not parse (BENCH_r04.json "parsed": null).
# The stdout line is now a compact object (<= COMPACT_LIMIT bytes: contract keys, `roofline` and `cpu_baseline` as numbers, one summary object per
# secondary leg); everything else -- trial lists, per-rank rows, notes -- goes to bench_detail.json beside this script (and gpurun_out/ when it exists).
COMPACT_LIMIT = 6000
DETAIL_NAME = "bench_detail.json"


def _dig(o, *path, default=None):
    for k in path:
        if isinstance(o, dict) and k in o:
            o = o[k]
        elif isinstance(o, (list, tuple)) and isinstance(k, int) and -len(o) <= k < len(o):
            o = o[k]
        else:
            return default
    return o


def _pick(o, keys, rename=None):
    """the sub-dict of `o` with `keys` that are present and not None (floats rounded to 6 significant digits)"""
    out = {}
    for k in keys:
        v = _dig(o, k)
        if v is None:
            continue
        if isinstance(v, float):
            v = float(f"{v:.6g}")
        out[(rename or {}).get(k, k)] = v
    return out


def _short(sv, n):
    return sv if not isinstance(sv, str) or len(sv) <= n else sv[:n - 3].rstrip() + "..."


_ROOFLINE_KEYS = ("bound", "kernel", "launches_per_step", "launches_per_frame", "avg_launch_us", "alg_gflop_per_launch", "alg_flop_per_sample", "achieved", "peak", "unit", "frac",
                  "frac_of_dense_f16_peak", "traffic", "traffic_gbytes_per_s", "mfma_passes_per_product", "kernel_share_of_step", "sum_of_launches_ms")


def compact_roofline(rf):
    """`roofline` as numbers only: the contract's fields, the dense-peak fraction whatever the operand format, the package power, cap and
    shader clock the kernel runs at (back to back, worst = largest-share grid first) and the MFMA-only ceiling of its instruction mix."""
    if not isinstance(rf, dict):
        return rf
    out = _pick(rf, _ROOFLINE_KEYS)
    out.setdefault("traffic", None)
    rows = _dig(rf, "power", "rows", default=[]) or []
    if rows:
        ws = [r.get("socket_w") for r in rows if r.get("socket_w") is not None]
        fs = [r.get("sclk_mhz") for r in rows if r.get("sclk_mhz") is not None]
        if ws:
            out["socket_w"] = round(sum(ws) / len(ws))
            out["cap_w"] = next((r.get("cap_w") for r in rows if r.get("cap_w")), None)
        if fs:
            out["sclk_mhz"] = round(sum(fs) / len(fs))
        out["alone_tflops_by_grid"] = [r.get("algorithmic_tflops") for r in rows]
    ceil = rf.get("mfma_only_ceiling_tflops")
    if isinstance(ceil, dict):
        key = "f16_plus_2_mx_fp6" if "f16+fp6" in str(rf.get("kernel")) else ("bf16_single_pass" if "bf16_single_pass" in ceil else "bf16x3_12_bf16_mfma_per_128k")
        out["mfma_only_ceiling"] = _pick(ceil, (key, key + "_layer_data"), {key: "random_bits", key + "_layer_data": "layer_data"})
    if rf.get("frac_of_mfma_only_ceiling") is not None:
        out["frac_of_mfma_only_ceiling"] = rf["frac_of_mfma_only_ceiling"]
    return out


def compact_cpu(cb):
    if not isinstance(cb, dict):
        return cb
    out = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "gflops"))
    out["sample"] = _short(cb.get("sample", ""), 110)
    return out


def compact_line(full):
    """The line the driver parses: every contract key of the full line, `roofline` / `cpu_baseline` / `parity` / `repeats`, and one summary object per secondary leg."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full[k] for k in keep if k in full}
    cfg = dict(full.get("config", {}))
    cfg["workload"] = _short(cfg.get("workload", ""), 200)
    cfg["parallelism"] = _short(cfg.get("parallelism", ""), 70)
    out["config"] = cfg
    for k in ("net_tflops", "samples_per_frame", "march_iterations", "field_tflops_algorithmic"):
        if k in full:
            out[k] = full[k]
    if "roofline" in full:
        out["roofline"] = compact_roofline(full["roofline"])
        pw = _dig(full, "repeats", "power")
        if pw:                                                             # the timed region's own package power and shader clock
            out["roofline"].update(_pick(pw, ("socket_w", "cap_w", "sclk_mhz"), {"socket_w": "step_socket_w", "cap_w": "cap_w", "sclk_mhz": "step_sclk_mhz"}))
    if "cpu_baseline" in full:
        out["cpu_baseline"] = compact_cpu(full["cpu_baseline"])
    if "parity" in full:
        out["parity"] = {k: (_short(v, 60) if isinstance(v, str) else v) for k, v in _pick(full["parity"], tuple(full["parity"])).items() if k != "note"}
    if "repeats" in full:
        out["repeats"] = _pick(full["repeats"], ("n", "median", "min", "max", "spread_pct"))
    if "phases" in full:
        out["phases"] = full["phases"]
    if "unet_conv_blocks" in full:
        out["unet_conv_blocks"] = full["unet_conv_blocks"]
    if "with_d2h" in full:
        out["with_d2h"] = _pick(full["with_d2h"], ("value", "ms_per_step"))
    if "alt" in full:
        out["alt"] = _pick(full["alt"], ("dtype", "value", "latent_linf_vs_oracle", "linf_vs_oracle", "u8_max_diff"))
    ms_ = full.get("multi_session")
    if isinstance(ms_, dict):
        at = _dig(ms_, "paced_sessions", "end_to_end", "at_max", default={}) or {}
        s_ = _pick(ms_, ("sessions_per_step", "batch_per_session", "value", "ms_per_step", "with_gpu_paste_back_720p"), {"value": "frames_per_s"})
        s_["sessions_per_gpu_at_25fps_end_to_end"] = _dig(ms_, "paced_sessions", "max_sessions_sustained")
        if _dig(ms_, "paced_sessions", "rank_mode"):
            s_["rank_mode"] = "host-lean (eager single-stream handles)"
        gm = _dig(ms_, "paced_sessions", "graph_mode_rank", "max_sessions_sustained")
        if gm is not None:
            s_["graph_mode_rank_sessions"] = gm
        s_.update(_pick(at, ("p50_ms", "p99_ms", "seconds", "gpu_busy_frac", "host_cpu_s_per_wall_s")))
        uv = _dig(ms_, "paced_sessions", "unet_vae_only", "max_sessions_sustained")
        if uv is not None:
            s_["unet_vae_only_sessions"] = uv
        ucb = _dig(ms_, "unet_conv_blocks", "mfma_issue_frac_of_bf16_peak")
        if ucb is not None:
            s_["unet_conv_blocks_mfma_issue_frac"] = ucb
        for k in ("streams", "cross_session_batch"):                        # (the wav2lip headline's multi_session)
            if _dig(ms_, k, "value") is not None:
                s_[k + "_frames_per_s"] = ms_[k]["value"]
        out["sessions"] = s_
    nd = full.get("node")
    if isinstance(nd, dict):
        n_ = _pick(nd, ("gpus", "sessions_per_node", "worst_rank_p99_ms", "latency_bound_ms", "host_cpu_s_per_wall_s_per_rank_at_capacity", "host_cores_for_8_ranks_at_capacity"))
        r1 = nd.get("ranks_on_one_gpu_one_session_each") or {}
        if r1:
            n_["ranks_on_one_gpu_8x1"] = _pick(r1, ("all_sustained", "worst_rank_p99_ms", "host_cpu_s_per_wall_s_total", "frames_per_s_total", "error"))
        if world_rows := nd.get("per_rank"):
            if len(world_rows) > 1:
                n_["sessions_by_rank"] = [r.get("max_sessions_sustained") for r in world_rows]
        out["node"] = n_
    w = full.get("wav2lip")
    if isinstance(w, dict):
        w_ = _pick(w, ("value", "unit", "ms_per_step", "dtype", "net_tflops"))
        w_["roofline"] = _pick(w.get("roofline", {}), ("kernel", "launches_per_step", "avg_launch_us", "achieved", "peak", "frac", "frac_of_dense_f16_peak", "traffic"))
        w_["linf_vs_oracle"] = _dig(w, "parity", "linf_vs_oracle")
        w_["cpu_baseline"] = _pick(w.get("cpu_baseline", {}), ("value", "unit", "cores", "kind"))
        w_["cross_session_batch_frames_per_s"] = _dig(w, "multi_session", "cross_session_batch", "value")
        w_["config0_b1_streaming"] = {"gpu_ms_per_frame_p50": _dig(w, "config0", "gpu", "ms_per_frame_p50"), "cpu_frames_per_s": _dig(w, "config0", "cpu_baseline", "value"),
                                      "u8_max_diff": _dig(w, "config0", "parity", "u8_max_diff_vs_cpu_oracle")}
        out["wav2lip"] = w_
    e = full.get("ernerf")
    if isinstance(e, dict):
        e_ = _pick(e, ("value", "unit", "ms_per_step", "dtype", "samples_per_frame"))
        e_["roofline"] = _pick(e.get("roofline", {}), ("kernel", "avg_launch_us", "achieved", "peak", "frac"))
        e_["image_linf_vs_oracle"] = _dig(e, "parity", "image_linf_max_vs_oracle")
        e_["through_dropin_frames_per_s"] = _dig(e, "through_dropin", "frames_per_s")
        wl = _dig(e, "through_dropin", "whole_frame_loop")
        if wl:            # nerfreal.py's whole per-frame sequence (rays -> render -> resize -> host -> uint8): the reference's operations around the render / the drop-in's utils
            e_["whole_frame_loop_ms"] = {"reference_operations": _dig(wl, "reference_operations", "ms_per_frame"), "dropin_utils": _dig(wl, "dropin_utils", "ms_per_frame")}
        e_["cpu_baseline"] = _pick(e.get("cpu_baseline", {}), ("value", "unit", "cores", "kind"))
        out["ernerf"] = e_
    wh = full.get("whisper")
    if isinstance(wh, dict):
        out["whisper_ms_per_window"] = _pick(wh, tuple(k for k in wh if k.startswith("exact_") and not k.endswith("linf_vs_full")))
    ft = full.get("frame_transport")
    if isinstance(ft, dict):
        out["frame_transport"] = _pick(ft, ("device_batch_to_ring_and_out_ms", "device_batch_reference_handoff_ms", "frame_ring_put_batch", "mp_queue_pickled"))
    c0 = full.get("config0")
    if isinstance(c0, dict):                                                # (the wav2lip headline carries its configs[0] leg at top level)
        out["config0_b1_streaming"] = {"gpu_ms_per_frame_p50": _dig(c0, "gpu", "ms_per_frame_p50"), "cpu_frames_per_s": _dig(c0, "cpu_baseline", "value")}
    out["bench_wall_s"] = round(time.perf_counter() - _T0, 1)
    out["detail"] = DETAIL_NAME
    # the size guard: drop the optional summaries, last first, until the line fits
    for k in ("frame_transport", "whisper_ms_per_window", "alt", "with_d2h", "phases", "ernerf", "wav2lip", "node", "sessions", "repeats", "unet_conv_blocks"):
        if len(json.dumps(out)) <= COMPACT_LIMIT:
            break
        out.pop(k, None)
    return out


def emit(full):
    """bench_detail.json (the full object) beside the script and under gpurun_out/ when that exists; the compact object as the ONE stdout line."""
    line = compact_line(full)
    blob = json.dumps(full, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_NAME), "w") as f:
                    f.write(blob)
            except OSError as e_:
                print(f"[bench] could not write {d}/{DETAIL_NAME}: {e_}", file=sys.stderr)
    text = json.dumps(line)
    assert len(text) <= COMPACT_LIMIT and "\n" not in text
    sys.stdout.flush()
    print(text, flush=True)
    return line


def _stage(name):
    """Wall-clock of the bench's own stages on stderr (the JSON line on stdout stays alone)."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f} s] {name}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 40 for musetalk, 200 for wav2lip)")
    ap.add_argument("--warmup", type=int, default=0, help="untimed warm-up steps (default: 5 / 20)")
    ap.add_argument("--workload", default="musetalk", choices=["musetalk", "wav2lip", "ernerf"])
    ap.add_argument("--batch", type=int, default=8, help="MuseTalk frames per step (configs[2]: 8)")
    ap.add_argument("--w2l-batch", type=int, default=16, help="Wav2Lip frames per step (configs[1]: 16)")
    ap.add_argument("--precision", default=os.environ.get("MF_PRECISION", "bf16x3"), choices=sorted(MFMA_PASSES))
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="0 skips the CPU baseline legs")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = cores available to this process (capped at 64)")
    ap.add_argument("--profile-iters", type=int, default=5)
    ap.add_argument("--dump-layers", default=None, help="write the per-launch tables (JSON) to this path")
    ap.add_argument("--sessions", type=int, default=8, help="concurrent sessions per GPU for the multi_session legs (0 = skip)")
    ap.add_argument("--paced", type=int, default=1, help="0 skips the real-time paced-sessions measurement of the multi_session leg")
    ap.add_argument("--pmc-traffic", type=int, default=1, help="0 skips the two rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--extras", type=int, default=1, help="0: only the headline workload (no second workload, alt mode, CPU legs)")
    ap.add_argument("--full", type=int, default=0, help="1: the long variants of the secondary legs (40 s paced confirmation + failing N + 1, UNet + VAE only search, per-stage trials, "
                                                        "8 ranks x 2 sessions on one GPU, the GRBM clock pass): ~6.5 min instead of ~3.5")
    ap.add_argument("--ranks-on-one-gpu", type=int, default=0, help="R > 0: ONLY the host-readiness leg -- R processes sharing GPU 0, each driving --sessions end-to-end "
                                                                      "sessions (default 2) in real time; the default line runs it with R = 8")
    ap.add_argument("--worker-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--worker-dir", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--worker-seconds", type=float, default=20.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.ranks_on_one_gpu > 0:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X; there is no CPU path to measure")
        if args.worker_rank >= 0:
            torch.cuda.set_device(0)
            ranks_on_one_gpu_worker(args, "cuda:0")
        else:
            n = args.sessions if any(a.startswith("--sessions") for a in sys.argv) else 2
            print(json.dumps({"ranks_on_one_gpu": ranks_on_one_gpu(args, args.ranks_on_one_gpu, n, args.worker_seconds)}), flush=True)
        return
    if args.steps <= 0:
        args.steps = {"musetalk": 40, "ernerf": 30}.get(args.workload, 200)
    if args.warmup <= 0:
        args.warmup = {"musetalk": 5, "ernerf": 3}.get(args.workload, 20)

    rank, local_rank, world = harness.init_dist("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to measure")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    extras = bool(args.extras) and world == 1

    if args.workload == "ernerf":
        run = ErNeRFRunner(args.precision, 512, device, seed=rank)
        elapsed = harness.timed_steps(run.step, args.steps, args.warmup, sync_fn=torch.cuda.synchronize, device=device)
        value = harness.aggregate_value(1, args.steps, elapsed, world)
        if rank == 0:
            if not extras:
                args.cpu_seconds = 0
            rep = ernerf_report(args, device, world, rank, value, elapsed / args.steps * 1e3, run)
            line = {"metric": "lip-sync frames/sec", "value": rep["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": args.precision, "data": "synthetic",
                    "config": {"workload": rep["workload"] + "; seeded random-init field", "sessions_at_25fps": round(value / 25.0, 1),
                               "parallelism": f"{world} independent replicas, sessions sharded by GPU, no collective"}}
            for k in ("samples_per_frame", "march_iterations", "field_tflops_algorithmic", "roofline", "parity", "cpu_baseline", "through_dropin"):
                if k in rep:
                    line[k] = rep[k]
            emit(line)
    elif args.workload == "wav2lip":
        run = Runner(args.precision, args.w2l_batch, device, seed=rank)
        elapsed = harness.timed_steps(run.step, args.steps, args.warmup, sync_fn=torch.cuda.synchronize, device=device)
        value = harness.aggregate_value(args.w2l_batch, args.steps, elapsed, world)
        if rank == 0:
            if not extras:
                args.sessions, args.cpu_seconds = 0, 0
            rep = wav2lip_report(args, device, world, rank, value, elapsed / args.steps * 1e3, run)
            line = {"metric": "lip-sync frames/sec", "value": rep["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                    "config": {"workload": rep["workload"] + "; seeded random-init weights", "batch_per_gpu": args.w2l_batch,
                               "sessions_at_25fps": round(value / 25.0, 1),
                               "parallelism": f"{world} independent replicas, sessions sharded by GPU, no collective"}}
            for k in ("roofline", "parity", "alt", "multi_session", "cpu_baseline", "config0"):
                if k in rep:
                    line[k] = rep[k]
            emit(line)
    else:
        _stage("build MuseTalk handles")
        run = MuseTalkRunner(args.precision, args.batch, device, seed=rank)
        _stage("timed steps")
        elapsed = harness.timed_steps(run.step, args.steps, args.warmup, sync_fn=torch.cuda.synchronize, device=device)
        value = harness.aggregate_value(args.batch, args.steps, elapsed, world)
        repeats = None
        if world == 1 and bool(args.extras) and args.profile_iters > 0:
            # the headline's own run-to-run spread (the timed region is short: K steps of ~19 ms): five more repeats of the same K steps, with the package power
            # and shader clock sampled over them
            _stage("headline repeats")
            vals = []
            with PowerSampler(local_rank) as ps:
                for _ in range(5):
                    el_r = harness.timed_steps(run.step, args.steps, 1, sync_fn=torch.cuda.synchronize, device=device, collective=False)
                    vals.append(args.batch * args.steps / el_r)
            vs = sorted(vals)
            repeats = {"n": len(vs), "steps_each": args.steps, "median": round(vs[len(vs) // 2], 1), "min": round(vs[0], 1), "max": round(vs[-1], 1),
                       "spread_pct": round((vs[-1] - vs[0]) / vs[len(vs) // 2] * 100, 2), "unit": "frames/s",
                       "note": "`value` above is the contract's single timed region; these are five further regions of the same length in the same process"}
            pw = ps.report(0.1)
            if pw:
                repeats["power"] = pw
        _stage("per-op profile")
        solo = world == 1
        q_on = args.precision == "bf16x3" and os.environ.get("MF_CONV_Q", "1") != "0"
        line = None
        if rank == 0:
            gf_frame = run.gflop_per_frame()      # summed over the handles' own op lists (tests/test_musetalk_full.py holds it to SURVEY Appendix C)
            line = {"metric": "lip-sync frames/sec @256x256", "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3+f16q" if q_on else args.precision, "data": "synthetic",
                    "config": {"workload": "MuseTalk step 256x256: pe(audio) + UNet(t=0) + VAE decode to uint8 frames, batch=8 latents + "
                                           "Whisper chunks per GPU, inputs resident in HBM (BASELINE.json configs[2]); assumed MuseTalk-v1 / "
                                           "sd-vae-ft-mse architecture, seeded random-init weights",
                               "batch_per_gpu": args.batch, "sessions_at_25fps": round(value / 25.0, 1),
                               "algorithmic_gflop_per_frame": round(gf_frame, 1),
                               "parallelism": f"{world} independent replicas, sessions sharded by GPU, no collective"},
                    "net_tflops": round(value / world * gf_frame / 1e3, 1)}
            if repeats:
                line["repeats"] = repeats
            if q_on:
                line["dtype_note"] = ("bf16x3 (hi, lo bf16 pairs, three MFMAs per product) everywhere except the VAE decoder's 3x3 resnet convs and upsamplers on maps >= 32 x 32 "
                                      "(28 % of the step's time): f16 + FP6 (e2m3, MX block scales) correction terms, 1.5 pass-equivalents per product, outputs bf16 (hi, lo); "
                                      "parity bound unchanged (tests/test_musetalk_full.py)")
            if args.profile_iters <= 0:          # child of a PMC pass: the timed steps above are all it needs to run
                print(json.dumps(line), flush=True)
                return
            rows = run.profile(args.profile_iters)
            rf, by = roofline(rows, args.precision, only_mfma=True)
            if extras and args.pmc_traffic:
                _stage("musetalk PMC traffic passes")
                rf["traffic"], rf["traffic_note"], clk = pmc_traffic(rf["kernel"], "musetalk", args.precision, ["--batch", str(args.batch)], want_clock=bool(args.full), grids=set(rf["launch_grids"]))
                if rf["traffic"]:
                    rf["traffic"] = round(rf["traffic"])
                    rf["traffic_gbytes_per_s"] = round(rf["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9, 1)
                if clk:
                    # the nominal peak assumes 2.4 GHz; under this kernel's MFMA load the chip clocks to its power budget (DVFS)
                    rf["effective_clock_ghz"] = clk
                    rf["peak_at_effective_clock"] = round(rf["peak"] * clk / 2.4, 1)
                    rf["frac_of_peak_at_effective_clock"] = round(rf["achieved"] / (rf["peak"] * clk / 2.4), 4)
                    rf["clock_note"] = "GRBM_GUI_ACTIVE / dispatch wall time under rocprofv3 (profiled passes clock ~3 % lower than un-profiled ones)"
            if bool(args.extras):
                # the ceiling a kernel of pure matrix instructions reaches on this device, next to the nominal peak
                ceil = mfma_only_ceiling(args.precision)
                rf["mfma_only_ceiling_tflops"] = ceil
                ref = ceil["f16_plus_2_mx_fp6"] if "f16+fp6" in rf["kernel"] else ceil.get("bf16_single_pass", ceil["bf16x3_12_bf16_mfma_per_128k"])
                rf["frac_of_mfma_only_ceiling"] = round(rf["achieved"] / ref, 4)
                rf["ceiling_note"] = ("mf_probe_mfma_ceiling: TFLOP/s of the convolution's own arithmetic from a loop of nothing but MFMAs (8 waves per CU, random "
                                      "operand bits) for the instruction mix of one product: 3 bf16 MFMAs as shipped; f16 + two block-scaled FP8 / FP6 correction terms "
                                      "(the FP6 form is what the VAE decoder's resnet convs run; numerics in tools/numerics_split_study.py, layouts in tools/mx_cross_probe.hip); "
                                      "profiles/r03_halo_q_loop_study.md: this kernel's own matrix-instruction floor, LDS floor and DMA share by ablation")
            if bool(args.extras) and "f16+fp6" in rf["kernel"]:
                _stage("roofline kernel under sustained load (power / clock)")
                try:
                    rf["power"] = roofline_kernel_power(args.batch, device)
                except Exception as e:     # measurement leg only
                    rf["power"] = {"error": f"{type(e).__name__}: {e}"}
            line["roofline"] = rf
            conv_rows = [r for r in rows if r["layer"].startswith("unet:") and r["flops"] > 0 and "attention" not in r["layer"]]
            if conv_rows:
                t = sum(r["ms"] for r in conv_rows); f = sum(r["flops"] for r in conv_rows)
                line["unet_conv_blocks"] = {"achieved_tflops": round(f / (t * 1e-3) / 1e12, 1), "ms": round(t, 3),
                                            "mfma_issue_frac_of_bf16_peak": round(MFMA_PASSES[args.precision] * f / (t * 1e-3) / 1e12 / BF16_DENSE_PEAK_TF, 3)}
            # where the step's time goes, by network: the per-op hipEvent sums (graph off) of the UNet and of the VAE decoder
            ph = {}
            for tag in ("unet", "vae"):
                rr = [r for r in rows if r["layer"].startswith(tag + ":")]
                ph[tag + "_ms"] = round(sum(r["ms"] for r in rr), 3)
                ph[tag + "_ops"] = len(rr)
                fl = sum(r["flops"] for r in rr)
                ph[tag + "_tflops"] = round(fl / max(ph[tag + "_ms"], 1e-9) / 1e9, 1)
            line["phases"] = ph
            _stage("parity vs oracle")
            line["parity"] = run.parity()
            # the same step bracketed as the reference brackets it (musereal.py:99-115): uint8 frames copied to the host inside the timed region
            el_h = harness.timed_steps(run.step_d2h, max(args.steps // 2, 1), 2, sync_fn=lambda: torch.cuda.synchronize(device), collective=False)
            line["with_d2h"] = {"value": round(args.batch * max(args.steps // 2, 1) / el_h, 1), "unit": "frames/s",
                                "ms_per_step": round(el_h / max(args.steps // 2, 1) * 1e3, 3),
                                "note": "step + vae.py:105's `.cpu().numpy()` of the uint8 frames (pageable host memory, one sync per step)"}
            if extras:
                _stage("whisper")
                line["whisper"] = whisper_report(args, device)
                _stage("frame transport")
                line["frame_transport"] = transport_report(args, device)
            if args.dump_layers:
                with open(args.dump_layers, "w") as f_:
                    json.dump({"musetalk_rows": rows, "by_kernel": by}, f_, indent=1)
            if bool(args.extras) and args.cpu_seconds > 0:
                _stage("cpu baseline")
                line["cpu_baseline"] = run.cpu_baseline(args.cpu_seconds, args.cpu_threads)
            if extras:
                del run
                run = None
                torch.cuda.empty_cache()
                other = "bf16" if args.precision == "bf16x3" else "bf16x3"
                _stage("alt precision")
                alt = MuseTalkRunner(other, args.batch, device)
                el2 = harness.timed_steps(alt.step, max(args.steps // 2, 1), 3, sync_fn=torch.cuda.synchronize)
                par = alt.parity()
                line["alt"] = {"dtype": other, "value": round(args.batch * max(args.steps // 2, 1) / el2, 1), "unit": "frames/s",
                               "latent_linf_vs_oracle": par["latent_linf_vs_oracle"], "u8_max_diff": par["u8_max_diff"]}
                del alt
                torch.cuda.empty_cache()
        # ---- BASELINE.json's second metric, on EVERY rank at once: max concurrent >= 25 fps sessions per GPU -> per node --------------------------
        if bool(args.extras) and args.sessions > 0 and args.profile_iters > 0:
            del run
            torch.cuda.empty_cache()
            if solo:
                _stage("multi session")
                ms_rep = muse_multi_session(args, device)       # cross-session batching (8 sessions x 8 frames per step) + the paced legs
                mine = ms_rep.get("paced_sessions")
                line["multi_session"] = ms_rep
            else:
                _stage("paced sessions on every rank")
                mine = muse_node_rank_leg(args, device)
            # each rank's measured capacity -> the node's admission cap and placement (harness.SessionPlacer; no data-path exchange)
            caps = None
            if mine is not None:
                my_cap = int(mine.get("max_sessions_sustained") or 0)
                placer = harness.SessionPlacer.from_measured(my_cap)
                caps = placer.capacity
                rows_n = [None] * world
                row = {"rank": rank, "gpu": local_rank, "max_sessions_sustained": my_cap,
                       "p99_ms_at_max": (mine["end_to_end"]["at_max"] or {}).get("p99_ms"), "frames_per_s_at_max": (mine["end_to_end"]["at_max"] or {}).get("frames_per_s")}
                if world > 1:
                    torch.distributed.all_gather_object(rows_n, row)
                else:
                    rows_n = [row]
                if rank == 0:
                    p99s = [r["p99_ms_at_max"] for r in rows_n if r["p99_ms_at_max"] is not None]
                    line["node"] = {"gpus": world, "sessions_per_node": int(sum(caps)), "unit": "concurrent MuseTalk sessions at >= 25 fps, end to end "
                                    "(PCM chunks -> Whisper -> UNet -> VAE -> 720p paste-back -> FrameRing), every rank measured at the same time",
                                    "worst_rank_p99_ms": max(p99s) if p99s else None, "latency_bound_ms": args.batch * 40, "per_rank": rows_n,
                                    "admission": "harness.SessionPlacer: cap = sum of the measured per-GPU capacities (app.py:42,79-80,705), a new session goes to "
                                                 "the GPU with the lowest load fraction",
                                    "north_star_target": ">= 64 sessions on 8 GPUs = 8 per GPU"}
                    hc = (mine["end_to_end"]["at_max"] or {}).get("host_cpu_s_per_wall_s")
                    if hc is not None:
                        # what the HOST side of a full node needs: every rank carrying its sustained session count (scheduler loop, consumer thread, PCM hand-in, launches)
                        line["node"]["host_cpu_s_per_wall_s_per_rank_at_capacity"] = hc
                        line["node"]["host_cores_for_8_ranks_at_capacity"] = round(8 * hc, 1)
                    if not solo:
                        line["paced_sessions_rank0"] = mine
        if rank == 0 and extras and args.sessions > 0 and getattr(args, "paced", 1):
            _stage("8 ranks on one GPU (host readiness)")
            torch.cuda.empty_cache()
            # 8 x 2 sessions load ONE GPU to ~85 % from eight time-sharing processes (the ranks' p99 then mostly measures that sharing); 8 x 1 leaves it half idle and
            # shows the host side by itself
            if args.full:
                line.setdefault("node", {})["ranks_on_one_gpu"] = ranks_on_one_gpu(args, 8, 2, 20.0)
            light = ranks_on_one_gpu(args, 8, 1, 12.0 if args.full else 8.0)
            line.setdefault("node", {})
            line["node"]["ranks_on_one_gpu_one_session_each"] = {k: light.get(k) for k in ("ranks", "sessions_per_rank", "seconds", "wall_s", "worst_rank_p99_ms", "all_sustained",
                                                                                              "host_cpu_s_per_wall_s_per_rank", "host_cpu_s_per_wall_s_total", "frames_per_s_total", "error")}
        if rank == 0 and extras:
            dl = args.dump_layers
            args.dump_layers = dl + ".wav2lip.json" if dl else None
            _stage("wav2lip leg")
            line["wav2lip"] = wav2lip_report(args, device, world, rank)
            _stage("ernerf leg")
            line["ernerf"] = ernerf_report(args, device, world, rank)
        if rank == 0:
            _stage("done")
            emit(line)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
