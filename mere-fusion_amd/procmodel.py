"""The reference's process model on ROCm.

`lipreal.py:29` / `musereal.py:58` ask `torch.cuda.is_available()` when the module is imported and then start one `mp.Process(target=inference)` per session with the
DEFAULT start method (`lipreal.py:170`, `musereal.py:162`) -- fork on Linux.  On CUDA that works (`is_available()` goes through NVML and does not initialise the
runtime); on ROCm it brings up HIP in the parent, and PyTorch refuses every device call in a forked child: "Cannot re-initialize CUDA in forked subprocess. To use CUDA
with multiprocessing, you must use the 'spawn' start method" (`tools/fork_probe.py` on MI355X: fork fails as soon as the parent has asked, spawn always works).  The
reference's files stay untouched, so the drop-in packages -- which those files import ABOVE that line (`lipreal.py:25`, `musereal.py:21-24`) -- choose the start method
before the reference creates its first queue, event or process: `spawn`, unless the application has already chosen one.

MF_MP_START=keep: do nothing (a deployment that sets the start method itself); MF_MP_START=fork|spawn|forkserver: that one."""
import multiprocessing as mp
import os
import sys
import warnings


def ensure_start_method():
    want = os.environ.get("MF_MP_START", "spawn")
    if want == "keep":
        return mp.get_start_method(allow_none=True)
    cur = mp.get_start_method(allow_none=True)
    if cur is None:
        mp.set_start_method(want)
        return want
    if cur == "fork" and want != "fork" and sys.platform.startswith("linux"):
        warnings.warn("mere-fusion_amd: the multiprocessing start method is already 'fork'; on ROCm a forked session process cannot use the GPU once the parent has "
                      "called torch.cuda.is_available() (lipreal.py:29).  Set it to 'spawn' before importing the drop-in, or leave it unset.", RuntimeWarning, stacklevel=2)
    return cur
