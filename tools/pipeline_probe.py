"""Does overlapping the UNet of step n+1 (below the power cap) with the VAE decode of step n (at the cap) raise throughput?
One UNet handle + one VAE handle, two streams, the UNet's output cloned per step; compared with the plain sequential step on the same box.
usage (GPU box): python tools/pipeline_probe.py [batch] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
r = bench.MuseTalkRunner("bf16x3", batch, dev)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def sequential(k):
    for _ in range(k):
        r.step()


def pipelined(k, depth=2):
    evs, outs = [], []
    for n in range(k):
        with torch.cuda.stream(sA):
            if n >= depth:
                sA.wait_event(outs[n - depth])          # bound the run-ahead: UNet n waits for VAE n - depth
            pred = r.unet.model(r.lat, r.t0, encoder_hidden_states=r.unet.pe(r.aud)).sample
            p = pred.clone()
            p.record_stream(sB)
            ev = torch.cuda.Event(); ev.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev)
            r.vae.decode_latents_device(p)
            done = torch.cuda.Event(); done.record(sB)
        outs.append(done)
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)


def timed(fn, k):
    fn(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(k); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return k * batch / el, el / k * 1e3

for rep in range(3):
    for name, fn in (("sequential", sequential), ("pipelined", pipelined)):
        with bench.PowerSampler(0) as ps:
            fps, ms = timed(fn, steps)
        pw = ps.report() or {}
        print(f"rep {rep} batch {batch} {name:10s}: {fps:7.1f} frames/s  {ms:7.3f} ms per step  {pw}", flush=True)
