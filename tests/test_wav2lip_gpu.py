"""GPU parity tests: the HIP generator (through the drop-in module -> custom op -> C ABI) against the
CPU oracle on the same seeded inputs, and against the golden vectors recorded from the reference.

Tolerances (written here as the contract):
  bf16x3 : L-inf <= 1e-3 on the [0,1] frames  -- the north-star bound (observed ~1e-4)
  bf16   : L-inf <= 8e-2, mean abs <= 6e-3    -- the inherent cost of 8-bit mantissas through 30+
           layers; a CPU simulation of bf16 storage gives L-inf 2.5e-2 at B=4
"""
import ctypes as C

import numpy as np
import pytest
import torch

import geometry_cases as G
from mere_fusion_amd import weights as W
from oracle import glue_ref
from oracle import wav2lip_ref as R

pytestmark = pytest.mark.gpu

TOL_X3 = 1.5e-4          # gate at ~4 x the measured 3.5e-5 (bound: 1e-3); VERDICT r03: the gate used to sit at the bound itself
TOL_BF16_LINF, TOL_BF16_MEAN = 8e-2, 6e-3


def _conv_layer(lib_built, case, precision, x):
    from mere_fusion_amd import _lib
    l = _lib.lib()
    _lib.init_device(0)
    p = G.case_params(case)
    sh, sw = (case["stride"], case["stride"]) if isinstance(case["stride"], int) else case["stride"]
    d = _lib.MfConv2dDesc(cin=case["cin"], cout=case["cout"], kh=case["k"], kw=case["k"], stride_h=sh, stride_w=sw,
                          pad_h=case["pad"], pad_w=case["pad"], transposed=case["transposed"],
                          output_padding=case["outpad"], residual=case["residual"], act=1,
                          in_h=case["h"], in_w=case["w"])
    h = C.c_void_p()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    keep = {k: v.contiguous() for k, v in p.items()}
    _lib.check(l.mf_conv2d_create(C.byref(d), ptr(keep["weight"]), ptr(keep["bias"]), ptr(keep["gamma"]),
                                  ptr(keep["beta"]), ptr(keep["mean"]), ptr(keep["var"]),
                                  _lib.PRECISIONS[precision], C.byref(h)), "conv2d_create")
    oh, ow = C.c_int(), C.c_int()
    _lib.check(l.mf_conv2d_out_shape(h, C.byref(oh), C.byref(ow)))
    xd = x.cuda().contiguous()
    y = torch.empty((x.shape[0], case["cout"], oh.value, ow.value), device="cuda")
    _lib.check(l.mf_conv2d_forward(h, ptr(xd), ptr(y), x.shape[0], None), "conv2d_forward")
    torch.cuda.synchronize()
    l.mf_conv2d_destroy(h)
    return y.cpu()


@pytest.mark.parametrize("case", G.CASES, ids=[c["name"] for c in G.CASES])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_conv_geometry_vs_golden(lib_built, conv_golden, case, precision):
    x = G.case_input(case)
    y = _conv_layer(lib_built, case, precision, x)
    ref = torch.from_numpy(conv_golden[f"y/{case['name']}"])
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    scale = ref.abs().max().item()
    if precision == "bf16x3":
        assert err <= 2e-4 * max(scale, 1.0), (case["name"], err, scale)
    else:
        assert err <= 3e-2 * max(scale, 1.0), (case["name"], err, scale)


@pytest.mark.parametrize("batch", [1, 3])
def test_conv_geometry_ragged_batch(lib_built, batch):
    """M not a multiple of the tile: rows are clamped on load and masked on store."""
    case = G.CASES[2]
    p = G.case_params(case)
    rng = np.random.default_rng(batch)
    x = torch.from_numpy(rng.standard_normal((batch, case["cin"], case["h"], case["w"])).astype(np.float32))
    sd = {"L.conv_block.0.weight": p["weight"], "L.conv_block.0.bias": p["bias"],
          "L.conv_block.1.weight": p["gamma"], "L.conv_block.1.bias": p["beta"],
          "L.conv_block.1.running_mean": p["mean"], "L.conv_block.1.running_var": p["var"]}
    want = R._layer(sd, "L", ("conv", 1, 1, 0, True), x)
    got = _conv_layer(lib_built, case, "bf16x3", x)
    assert (got - want).abs().max().item() <= 2e-4 * want.abs().max().item()


@pytest.mark.parametrize("batch", [2, 1, 16, 5])
def test_generator_bf16x3_vs_oracle(gpu_model_factory, sd0, batch):
    m = gpu_model_factory("bf16x3")
    mel, face, _ = W.make_lip_inputs(batch, batch)
    want = R.wav2lip_forward(sd0, mel, face)
    with torch.no_grad():
        got = m(mel.cuda(), face.cuda()).cpu()
    assert got.shape == want.shape == (batch, 3, 96, 96)
    err = (got - want).abs().max().item()
    assert err <= TOL_X3, err


def test_generator_vs_reference_golden(gpu_model_factory, wav2lip_golden):
    """Against the vectors recorded from the real `wav2lip.models.Wav2Lip`."""
    m = gpu_model_factory("bf16x3")
    mel, face, _ = W.make_lip_inputs(2, 0)
    with torch.no_grad():
        got = m(mel.cuda(), face.cuda()).cpu()
    ref = torch.from_numpy(wav2lip_golden["output"])
    assert (got - ref).abs().max().item() <= TOL_X3
    for k in ["audio_embedding"] + [f"face_encoder_blocks.{i}" for i in range(7)] + [f"face_decoder_blocks.{i}" for i in range(7)]:
        t = m.read_tap(k, 2).cpu()
        sample = t.reshape(-1).numpy()[:: G.TAP_STRIDE][: G.TAP_MAX]
        np.testing.assert_allclose(sample, wav2lip_golden[f"tap_sample/{k}"], rtol=2e-3, atol=2e-3, err_msg=k)
        np.testing.assert_allclose(t.double().abs().sum().item(), wav2lip_golden[f"tap_abssum/{k}"], rtol=1e-4, err_msg=k)


def test_generator_bf16_vs_oracle(gpu_model_factory, sd0):
    m = gpu_model_factory("bf16")
    mel, face, _ = W.make_lip_inputs(4, 0)
    want = R.wav2lip_forward(sd0, mel, face)
    with torch.no_grad():
        got = m(mel.cuda(), face.cuda()).cpu()
    d = (got - want).abs()
    assert d.max().item() <= TOL_BF16_LINF and d.mean().item() <= TOL_BF16_MEAN, (d.max().item(), d.mean().item())


def test_graph_replay_is_deterministic(gpu_model_factory):
    """call 1 runs eagerly, call 2 captures the hipGraph, calls 3+ replay it."""
    m = gpu_model_factory("bf16x3")
    mel, face, _ = W.make_lip_inputs(4, 9)
    mel, face = mel.cuda(), face.cuda()
    outs = []
    with torch.no_grad():
        for _ in range(4):
            outs.append(m(mel, face).cpu())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # a different input through the captured graph must change the output
    mel2, face2, _ = W.make_lip_inputs(4, 10)
    with torch.no_grad():
        o2 = m(mel2.cuda(), face2.cuda()).cpu()
    assert not torch.equal(o2, outs[0])


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [16, 5])
def test_copies_of_an_input_are_bit_identical_at_every_batch_position(gpu_model_factory, batch):
    """A row's result may not depend on where its image sits in the batch: `batch` copies of ONE (mel, face) pair give `batch` identical frames, call after call
    (eager, capture, replay).  The check that caught the gfx950 packed-fp32 erratum in the UNet (tests/test_musetalk_full.py), held on this
    network too (VERDICT r05 item 2 iv): its kernels run their epilogues beside other workgroups' MFMA loops just the same."""
    m = gpu_model_factory("bf16x3")
    mel, face, _ = W.make_lip_inputs(1, 21)
    mel, face = mel.repeat(batch, 1, 1, 1).cuda(), face.repeat(batch, 1, 1, 1).cuda()
    first = None
    with torch.no_grad():
        for call in range(4):
            out = m(mel, face)
            for k in range(1, batch):
                assert torch.equal(out[k], out[0]), (call, k, float((out[k] - out[0]).abs().max()))
            first = out.clone() if first is None else first
            assert torch.equal(out, first), call


def test_linearity_free_properties_at_full_batch(gpu_model_factory, sd0):
    """B=16 (BASELINE config 2): frames are independent -- a batch equals its frames run one by one."""
    m = gpu_model_factory("bf16x3")
    mel, face, _ = W.make_lip_inputs(16, 2)
    mel, face = mel.cuda(), face.cuda()
    with torch.no_grad():
        full = m(mel, face)
        one = torch.cat([m(mel[i:i + 1], face[i:i + 1]) for i in (0, 7, 15)])
    # tile shapes and split-K depend on the batch size, so the fp32 summation order (and with it the
    # rounding of every re-split activation) differs: equality holds to the parity bound, not bitwise
    assert (full[[0, 7, 15]] - one).abs().max().item() <= 2e-4
    assert full.min() >= 0 and full.max() <= 1


def test_forward_u8_matches_glue_oracle(gpu_model_factory, sd0):
    """lipreal.py:115-126 fused on the GPU == oracle glue around the oracle generator."""
    m = gpu_model_factory("bf16x3")
    mel, _, u8 = W.make_lip_inputs(3, 4)
    img, _ = glue_ref.face_batch(u8, [np.zeros((80, 16))] * 3)
    want = glue_ref.frames_from_pred(R.wav2lip_forward(sd0, mel, torch.from_numpy(img)).numpy())
    with torch.no_grad():
        got = m.forward_u8(mel.cuda(), torch.from_numpy(u8).cuda()).cpu().numpy()
    assert got.shape == (3, 96, 96, 3)
    assert np.abs(got - want).max() <= 255 * TOL_X3
    # what process_frames sees after astype(uint8) differs by at most one grey level
    assert np.abs(glue_ref.to_uint8(got).astype(int) - glue_ref.to_uint8(want).astype(int)).max() <= 1


def test_five_dim_inputs(gpu_model_factory):
    """wav2lip.py:92-94,118-120: (B,T,1,80,16) + (B,6,T,96,96) -> (B,3,T,96,96)."""
    m = gpu_model_factory("bf16x3")
    mel, face, _ = W.make_lip_inputs(4, 1)
    mel5 = mel.reshape(2, 2, 1, 80, 16).cuda()
    face5 = face.reshape(2, 2, 6, 96, 96).permute(0, 2, 1, 3, 4).contiguous().cuda()
    with torch.no_grad():
        out5 = m(mel5, face5)
        flat = m(torch.cat([mel5[:, i] for i in range(2)]), torch.cat([face5[:, :, i] for i in range(2)]))
    assert out5.shape == (2, 3, 2, 96, 96)
    assert torch.equal(out5[:, :, 1], flat[2:4])


def test_module_prefix_and_reload(lib_built, sd0):
    from mere_fusion_amd.wav2lip.models import Wav2Lip
    m = Wav2Lip(precision="bf16x3")
    m.load_state_dict(R.strip_module_prefix({"module." + k: v for k, v in sd0.items()}))
    m = m.to("cuda").eval()
    mel, face, _ = W.make_lip_inputs(1, 0)
    with torch.no_grad():
        a = m(mel.cuda(), face.cuda()).cpu()
        m.load_state_dict(W.make_wav2lip_state_dict(1))      # new checkpoint -> handle rebuilt
        b = m(mel.cuda(), face.cuda()).cpu()
    want_b = R.wav2lip_forward(W.make_wav2lip_state_dict(1), mel, face)
    assert (b - want_b).abs().max().item() <= TOL_X3
    assert (a - b).abs().max().item() > 1e-2


def test_lip_session_driver(gpu_model_factory, sd0):
    """Queue-free restatement of run_step + inference(): GPU mel -> chunks -> fused generator."""
    from mere_fusion_amd import lip_driver as D
    from oracle import mel_ref
    m = gpu_model_factory("bf16x3")
    rng = np.random.default_rng(0)
    faces = rng.integers(0, 256, (5, 96, 96, 3), dtype=np.uint8)
    sess = D.LipSession(m, faces)
    fe = D.LipASRFrontend(batch_size=4)
    fe.warm_up()
    new = [(0.1 * rng.standard_normal(320)).astype(np.float32) for _ in range(8)]
    chunks = fe.run_step(new)
    assert chunks.shape == (4, 1, 80, 16)
    wav = np.concatenate([np.zeros(320 * 20, np.float32)] + new)
    mel = mel_ref.melspectrogram(wav)
    ref_chunks, _ = glue_ref.mel_chunks(mel, 28, 10, 10, 50)
    np.testing.assert_allclose(chunks[:, 0].cpu().numpy(), np.stack(ref_chunks), atol=1e-5)
    frames, idx = sess.step(chunks)
    assert idx == [0, 1, 2, 3] and frames.shape == (4, 96, 96, 3)
    frames2, idx2 = sess.step(chunks)
    assert idx2 == [4, 4, 3, 2]
    img, melb = glue_ref.face_batch(faces[idx2], ref_chunks)
    want = glue_ref.frames_from_pred(R.wav2lip_forward(sd0, torch.from_numpy(melb), torch.from_numpy(img)).numpy())
    assert np.abs(frames2.cpu().numpy() - want).max() <= 255 * TOL_X3


def test_eight_sessions_on_eight_streams_batch16(lib_built, sd0):
    """BASELINE.json configs[3], per-GPU shape: 8 concurrent sessions, one hipStream + one generator handle + one face cache each, B = 16
    mel chunks per step (lipreal.py runs one inference loop per session).  Two steps are enqueued on every stream before anything is
    synchronised, so the sessions really overlap; EVERY session's frames are then held to the oracle (mirror-indexed faces included)."""
    from mere_fusion_amd import lip_driver as D
    from mere_fusion_amd.wav2lip.models import Wav2Lip
    S, Bs = 8, 16
    streams = [torch.cuda.Stream() for _ in range(S)]
    sessions, mels, faces = [], [], []
    for s in range(S):
        m = Wav2Lip(precision="bf16x3")
        m.load_state_dict(sd0)
        m = m.to("cuda").eval()
        mel, _, u8 = W.make_lip_inputs(Bs, 300 + s)
        f = np.random.default_rng(50 + s).integers(0, 256, (11 + s, 96, 96, 3), dtype=np.uint8)     # 11..18 cached crops: the walk mirrors
        sessions.append(D.LipSession(m, f))
        mels.append(mel)
        faces.append(f)
    torch.cuda.synchronize()
    outs = [[] for _ in range(S)]
    for step in range(2):
        for s in range(S):
            with torch.cuda.stream(streams[s]), torch.no_grad():
                fr, idx = sessions[s].step(mels[s].cuda(non_blocking=True))
                outs[s].append((fr, idx))
    torch.cuda.synchronize()
    worst = 0.0
    for s in range(S):
        for step in range(2):
            fr, idx = outs[s][step]
            want_idx = [glue_ref.mirror_index(len(faces[s]), step * Bs + i) for i in range(Bs)]
            assert idx == want_idx
            img, melb = glue_ref.face_batch(faces[s][want_idx], [m_[0] for m_ in mels[s].numpy()])
            want = glue_ref.frames_from_pred(R.wav2lip_forward(sd0, torch.from_numpy(melb), torch.from_numpy(img)).numpy())
            err = np.abs(fr.cpu().numpy() - want).max() / 255.0
            worst = max(worst, err)
            assert err <= TOL_X3, (s, step, err)
    print(f"8 sessions x 8 streams x B16, 2 steps each: worst L-inf vs oracle {worst:.3e}")


def test_graphs_survive_workspace_growth(lib_built, sd0):
    """the split count comes from a batch-dependent cost model, so a SMALLER batch can need a LARGER split-K workspace
    than a batch whose hipGraph is already captured.  Replay B = 32, run B = 16 / 8 / 24 (eager + capture), replay B = 32 again, and compare
    with a handle that never uses graphs (MF_NO_GRAPH=1 at create time)."""
    import os
    from mere_fusion_amd.wav2lip.models import Wav2Lip

    def model():
        m = Wav2Lip(precision="bf16x3")
        m.load_state_dict(sd0)
        return m.to("cuda").eval()
    mel, face, _ = W.make_lip_inputs(32, 21)
    mel, face = mel.cuda(), face.cuda()
    g = model()
    with torch.no_grad():
        first = [g(mel, face).cpu() for _ in range(3)][-1]                   # eager, capture, replay
        for b in (16, 8, 24, 16):
            for _ in range(2):
                g(mel[:b], face[:b])
        again = g(mel, face).cpu()
        old = os.environ.get("MF_NO_GRAPH")
        os.environ["MF_NO_GRAPH"] = "1"
        try:
            e = model()
            eager = e(mel, face).cpu()
        finally:
            if old is None:
                del os.environ["MF_NO_GRAPH"]
            else:
                os.environ["MF_NO_GRAPH"] = old
    assert torch.equal(first, again)
    assert torch.equal(eager, again)


@pytest.mark.gpu
def test_reference_process_model_through_the_dropin(lib_built):
    """The reference's process model, unedited: `app.py:549` sets the start method to spawn, a session's `LipReal` is built later (`from lipreal import LipReal`, app.py:339),
    `lipreal.py` imports the drop-in (:25), asks `torch.cuda.is_available()` at import time (:29) and starts the session's worker with `mp.Process(target=inference)` (:170),
    which builds the model, says `.to('cuda')` and runs it.  The spawned worker re-imports everything through the same PYTHONPATH: the drop-in, its placement hook and the
    library must come up in it on their own.  (fork instead of spawn cannot work on ROCm once the parent has asked `is_available()`: tools/fork_probe.py -- which is why
    app.py's own choice matters and the drop-in does not touch it.)"""
    import os
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    script = r'''
import multiprocessing as mp, sys

def inference(q):                                         # lipreal.py:75: the worker loads the model itself
    import torch
    from wav2lip.models import Wav2Lip
    from mere_fusion_amd import weights as W
    device = 'cuda' if torch.cuda.is_available() else 'cpu'
    m = Wav2Lip(); m.load_state_dict(W.make_wav2lip_state_dict(0)); m = m.to(device).eval()
    mel, face, _ = W.make_lip_inputs(2, 0)
    with torch.no_grad():
        q.put(float(m(mel.to(device), face.to(device)).float().mean()))

if __name__ == "__main__":
    mp.set_start_method('spawn')                          # app.py:549
    from wav2lip.models import Wav2Lip                    # lipreal.py:25 (imported when the first session is built, app.py:339)
    import torch
    device = 'cuda' if torch.cuda.is_available() else 'cpu'   # lipreal.py:29
    q = mp.Queue()                                        # lipreal.py:160
    p = mp.Process(target=inference, args=(q,)); p.start()     # lipreal.py:170
    print("WORKER", mp.get_start_method(), q.get(timeout=240)); p.join(30)
'''
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mf_procmodel_probe.py")
    with open(path, "w") as f:
        f.write(script)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "mere-fusion_amd", "dropin")]))
    out = subprocess.run([sys.executable, path], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "WORKER spawn 0." in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]
