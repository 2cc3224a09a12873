"""The reference's own construction path, from files (VERDICT r02 "what's missing" item 4): `load_all_model()` (musetalk/utils/utils.py:19-25 ->
`VAE(model_path="./models/sd-vae-ft-mse/")` vae.py:24, `UNet(unet_config=..json, model_path=..bin)` unet.py:37-41,
`Audio2Feature(model_path="./models/whisper/tiny.pt")` whisper/__init__.py:108-116) and lipreal.py:33-53 `load_model("./models/wav2lip.pth")` with its
`module.` prefixes -- driven through the drop-in import seam (`mere-fusion_amd/dropin` on sys.path, the imports musereal.py / lipreal.py make) on
synthetic checkpoints written in the on-disk layouts those loaders read.  The objects built from files must behave bit-identically to the ones every
other test builds from in-memory state dicts."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from mere_fusion_amd import weights as W

DROPIN = os.path.join(ROOT, "mere-fusion_amd", "dropin")


def _legacy_attention_keys(vsd):
    """the published sd-vae-ft-mse file carries the OLD diffusers attention names, projections stored as 1x1 convs in some exports"""
    back = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    out = {}
    for k, v in vsd.items():
        for new, old in back.items():
            tag = f".attentions.0.{new}."
            if tag in k:
                k = k.replace(tag, f".attentions.0.{old}.")
                if k.endswith("weight"):
                    v = v.reshape(v.shape[0], v.shape[1], 1, 1)
        out[k] = v
    return out


def _write_models(root, cfg, usd, vsd, wsd, vae_format):
    from safetensors.torch import save_file
    from mere_fusion_amd.musetalk.config import unet_config_json, vae_config_json
    os.makedirs(root / "models" / "whisper"), os.makedirs(root / "models" / "sd-vae-ft-mse"), os.makedirs(root / "models" / "musetalk")
    dims = dict(W.WHISPER_TINY)
    full = {"encoder." + k: v for k, v in wsd.items()}
    full["decoder.token_embedding.weight"] = torch.zeros(4, 4)          # a real checkpoint also holds the decoder: ignored
    torch.save({"dims": dims, "model_state_dict": full}, root / "models" / "whisper" / "tiny.pt")
    vcfg = vae_config_json(cfg["vae"])
    vcfg.update(_class_name="AutoencoderKL", sample_size=256, in_channels=3)       # what a diffusers config.json also carries
    json.dump(vcfg, open(root / "models" / "sd-vae-ft-mse" / "config.json", "w"))
    if vae_format == "safetensors":
        save_file({k: v.contiguous() for k, v in vsd.items()}, str(root / "models" / "sd-vae-ft-mse" / "diffusion_pytorch_model.safetensors"))
        torch.save({"poison": torch.zeros(1)}, root / "models" / "sd-vae-ft-mse" / "diffusion_pytorch_model.bin")   # must NOT be the one that is read
    else:
        torch.save(_legacy_attention_keys(vsd), root / "models" / "sd-vae-ft-mse" / "diffusion_pytorch_model.bin")
    json.dump(unet_config_json(cfg["unet"]), open(root / "models" / "musetalk" / "musetalk.json", "w"))
    torch.save(usd, root / "models" / "musetalk" / "pytorch_model.bin")


def test_weight_file_choice_and_legacy_keys(tmp_path):
    """CPU: `.safetensors` wins over `.bin` (what from_pretrained does), and the legacy attention names map onto the new ones."""
    from safetensors.torch import save_file
    from mere_fusion_amd.musetalk.models.vae import load_diffusers_weights, remap_legacy_attention_keys
    a, b = {"x": torch.ones(2)}, {"x": torch.zeros(2)}
    torch.save(b, tmp_path / "diffusion_pytorch_model.bin")
    assert torch.equal(load_diffusers_weights(str(tmp_path))["x"], b["x"])
    save_file(a, str(tmp_path / "diffusion_pytorch_model.safetensors"))
    assert torch.equal(load_diffusers_weights(str(tmp_path))["x"], a["x"])
    legacy = {"decoder.mid_block.attentions.0.query.weight": torch.zeros(4, 4, 1, 1), "decoder.mid_block.attentions.0.proj_attn.bias": torch.zeros(4),
              "decoder.conv_in.weight": torch.zeros(1)}
    assert sorted(remap_legacy_attention_keys(legacy)) == ["decoder.conv_in.weight", "decoder.mid_block.attentions.0.to_out.0.bias",
                                                           "decoder.mid_block.attentions.0.to_q.weight"]


@pytest.mark.gpu
@pytest.mark.parametrize("vae_format", ["safetensors", "bin_legacy_keys"])
def test_load_all_model_from_files_matches_state_dict_path(lib_built, tmp_path, monkeypatch, vae_format):
    from mere_fusion_amd.musetalk.config import unet_config_json, vae_config_json
    from mere_fusion_amd.musetalk.models.unet import UNet
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature
    from oracle import musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    vsd.update(W.make_musetalk_vae_encoder_state_dict(cfg, 0))          # the published file holds encoder + decoder
    wsd = W.make_whisper_encoder_state_dict(0)
    _write_models(tmp_path, cfg, usd, vsd, wsd, vae_format)
    monkeypatch.chdir(tmp_path)                                         # the loaders use paths relative to the working directory (utils.py:20-23)
    monkeypatch.syspath_prepend(DROPIN)
    for m in [k for k in sys.modules if k == "musetalk" or k.startswith("musetalk.")]:
        monkeypatch.delitem(sys.modules, m)
    from musetalk.utils.utils import load_all_model                     # the import musereal.py:21 makes
    audio_processor, vae, unet, pe = load_all_model()

    lat, aud = W.make_musetalk_inputs(2, 3)
    t0 = torch.tensor([0]).cuda()
    pred_f = unet.model(lat.cuda(), t0, encoder_hidden_states=pe(aud.cuda())).sample
    frames_f = vae.decode_latents(pred_f)
    unet_m = UNet(unet_config_json(cfg["unet"]), usd)
    vae_m = VAE(config=vae_config_json(cfg["vae"]), state_dict=vsd)
    pred_m = unet_m.model(lat.cuda(), t0, encoder_hidden_states=unet_m.pe(aud.cuda())).sample
    assert torch.equal(pred_f, pred_m)
    assert np.array_equal(frames_f, vae_m.decode_latents(pred_m)) and frames_f.dtype == np.uint8 and frames_f.shape == (2, 256, 256, 3)
    wav = W.make_speech_like_wav(11520, 1)
    a2f_m = Audio2Feature(state_dict=wsd, n_head=W.WHISPER_TINY["n_audio_head"])
    assert np.array_equal(audio_processor.audio2feat(wav), a2f_m.audio2feat(wav))
    # avatar preparation reads the encoder half of the same file (mere_musetalk.py:303-304)
    crop = np.random.default_rng(0).integers(0, 256, (256, 256, 3), dtype=np.uint8)
    g1, g2 = torch.Generator(device="cuda").manual_seed(1), torch.Generator(device="cuda").manual_seed(1)
    assert torch.equal(vae.get_latents_for_unet(crop, generator=g1), vae_m.get_latents_for_unet(crop, generator=g2))


@pytest.mark.gpu
def test_wav2lip_pth_with_module_prefixes(lib_built, tmp_path, monkeypatch, sd0):
    """lipreal.py:33-53 verbatim on a `{"state_dict": {"module.<key>": ...}}` file (a DataParallel export), with the drop-in class."""
    monkeypatch.syspath_prepend(DROPIN)
    for m in [k for k in sys.modules if k == "wav2lip" or k.startswith("wav2lip.")]:
        monkeypatch.delitem(sys.modules, m)
    from wav2lip.models import Wav2Lip                                  # lipreal.py:25
    path = tmp_path / "wav2lip.pth"
    torch.save({"state_dict": {"module." + k: v for k, v in sd0.items()}, "global_step": 1}, path)
    model = Wav2Lip()
    s = torch.load(path)["state_dict"]                                  # lipreal.py:33-41
    new_s = {}
    for k, v in s.items():
        new_s[k.replace("module.", "")] = v                             # lipreal.py:48-49
    model.load_state_dict(new_s)
    model = model.to("cuda").eval()                                     # lipreal.py:52-53
    mel, face, _ = W.make_lip_inputs(2, 0)
    ref = Wav2Lip()
    ref.load_state_dict(sd0)
    ref = ref.to("cuda").eval()
    with torch.no_grad():
        assert torch.equal(model(mel.cuda(), face.cuda()), ref(mel.cuda(), face.cuda()))
