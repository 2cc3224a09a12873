"""Measured launch configurations (mf_conv_tune, the production default): on the first forward at a batch size every implicit-GEMM layer times its
tile x split-K x operand-path candidates on the live buffers and keeps the fastest; the second forward captures the graph with them.
The results must stay inside the same parity bounds as with the cost model alone, and must not depend on WHEN the tuning happened."""
import numpy as np
import pytest
import torch

from mere_fusion_amd import weights as W

pytestmark = pytest.mark.gpu


def test_wav2lip_autotuned_matches_reference_golden(lib_built, sd0, wav2lip_golden, autotuned):
    from mere_fusion_amd.wav2lip.models import Wav2Lip
    m = Wav2Lip(precision="bf16x3")
    m.load_state_dict(sd0)
    m = m.to("cuda").eval()
    mel, face, _ = W.make_lip_inputs(2, 0)                     # the inputs tests/golden/make_golden.py recorded the reference on
    mel, face = mel.cuda(), face.cuda()
    with torch.no_grad():
        first = m(mel, face).cpu().numpy()          # eager + tuning + eager
        second = m(mel, face).cpu().numpy()         # capture
        third = m(mel, face).cpu().numpy()          # replay
    want = wav2lip_golden["output"]
    assert np.abs(first - want).max() <= 1e-3
    np.testing.assert_array_equal(first, second)
    np.testing.assert_array_equal(second, third)
    # batch 16 (BASELINE configs[1]) against the oracle
    from oracle import wav2lip_ref
    mel16, face16, _ = W.make_lip_inputs(16, 3)
    with torch.no_grad():
        got = m(mel16.cuda(), face16.cuda()).cpu()
    assert (got - wav2lip_ref.wav2lip_forward(sd0, mel16, face16)).abs().max().item() <= 1e-3


def test_unet_small_autotune_on_off_agree(lib_built, monkeypatch):
    from mere_fusion_amd.musetalk.models.unet import UNet
    from mere_fusion_amd.musetalk.config import unet_config_json
    from oracle import musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd = W.make_musetalk_unet_state_dict(cfg, 0)
    lat, aud = W.make_musetalk_inputs(3, 2)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MF_AUTOTUNE", mode)
        unet = UNet(unet_config_json(cfg["unet"]), usd, max_batch=3)
        runs = [unet.model(lat.cuda(), torch.tensor([0]).cuda(), encoder_hidden_states=unet.pe(aud.cuda())).sample.cpu() for _ in range(3)]
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[1], runs[2])      # eager(+tune), capture, replay: one set of configurations
        outs[mode] = runs[0]
    want = R.unet_forward(usd, cfg["unet"], lat, torch.tensor([0]), R.add_positional_encoding(aud))
    for mode in outs:
        assert (outs[mode] - want).abs().max().item() <= 2e-3
    assert (outs["0"] - outs["1"]).abs().max().item() <= 2e-4                        # different tiles: different fp32 summation order only
