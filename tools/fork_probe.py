#!/usr/bin/env python3
"""Fork or spawn on ROCm?  lipreal.py / musereal.py import torch, ask `torch.cuda.is_available()` at import time (lipreal.py:29) and then
start one `mp.Process(target=inference)` per session with the DEFAULT start method (fork on Linux; lipreal.py:170, musereal.py:162); the child loads the model and runs
it on the GPU.  This probe does the same with the drop-in Wav2Lip, for fork and for spawn, with and without the parent having touched the device.
    python tools/fork_probe.py        (GPU box)"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mere-fusion_amd", "dropin")]


def inference(q, tag):
    try:
        import torch
        from wav2lip.models import Wav2Lip                       # lipreal.py:25 through the drop-in
        from mere_fusion_amd import weights as W
        m = Wav2Lip()
        m.load_state_dict(W.make_wav2lip_state_dict(0))
        m = m.to("cuda").eval()
        mel, face, _ = W.make_lip_inputs(2, 0)
        with torch.no_grad():
            out = m(mel.cuda(), face.cuda())
        q.put((tag, "ok", float(out.float().mean())))
    except BaseException as e:                                   # noqa: BLE001
        q.put((tag, "FAILED", repr(e)[:300]))


def trial(method, parent_touches):
    ctx = mp.get_context(method)
    q = ctx.Queue()
    tag = f"{method}, parent {parent_touches}"
    p = ctx.Process(target=inference, args=(q, tag))
    p.start()
    try:
        print(q.get(timeout=240), flush=True)
    except Exception as e:                                       # noqa: BLE001
        print((tag, "NO ANSWER", repr(e)), flush=True)
    p.join(timeout=30)
    if p.is_alive():
        p.kill()


if __name__ == "__main__":
    import torch
    trial("fork", "imported torch only")
    trial("spawn", "imported torch only")
    print("parent: torch.cuda.is_available() ->", torch.cuda.is_available(), "(lipreal.py:29); initialised:", torch.cuda.is_initialized(), flush=True)
    trial("fork", "asked is_available()")
    trial("spawn", "asked is_available()")
    torch.zeros(1, device="cuda")
    print("parent: allocated on the device; initialised:", torch.cuda.is_initialized(), flush=True)
    trial("fork", "has a HIP context")
    trial("spawn", "has a HIP context")
