"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the Wav2Lip generator.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (mere-fusion_amd/) never does: it fails loudly when the HIP library is absent.

What is restated (paths relative to the reference checkout):
  * Conv2d          wav2lip/models/conv.py:5-19   ReLU(BN(conv(x)) [+ x])
  * Conv2dTranspose wav2lip/models/conv.py:33-44  ReLU(BN(convT(x)))
  * Wav2Lip.forward wav2lip/models/wav2lip.py:87-125 (ctor :12-85): audio encoder -> embedding,
    face encoder -> 7 skips, decoder with cat(x, skip) after every block, output_block + sigmoid.

It is written as a flat layer table + functional torch ops (no nn.Module tree), consuming the
reference's state-dict keys directly, so it is an independent statement of the arithmetic
rather than a copy of the module code.

Pinned by: tests/golden/wav2lip_golden.npz, produced by tests/golden/make_golden.py, which
imports the real `wav2lip.models.Wav2Lip` from /root/reference in the build container and
records its outputs on seeded weights/inputs (the reference has no tests or vectors of its own,
SURVEY 4). tests/test_oracle_golden.py checks this restatement against those vectors.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, conv.py:10

# (prefix, kind, stride, padding, output_padding, residual)
# kind: "conv" = Conv2d+BN+ReLU, "convT" = ConvTranspose2d+BN+ReLU
FACE_ENCODER = [  # wav2lip.py:15-36
    [("conv", 1, 3, 0, False)],
    [("conv", 2, 1, 0, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("conv", 2, 1, 0, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("conv", 2, 1, 0, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("conv", 2, 1, 0, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("conv", 2, 1, 0, False), ("conv", 1, 1, 0, True)],
    [("conv", 1, 0, 0, False), ("conv", 1, 0, 0, False)],
]
AUDIO_ENCODER = [  # wav2lip.py:38-55
    ("conv", 1, 1, 0, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True),
    ("conv", (3, 1), 1, 0, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True),
    ("conv", 3, 1, 0, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True),
    ("conv", (3, 2), 1, 0, False), ("conv", 1, 1, 0, True),
    ("conv", 1, 0, 0, False), ("conv", 1, 0, 0, False),
]
FACE_DECODER = [  # wav2lip.py:57-81
    [("conv", 1, 0, 0, False)],
    [("convT", 1, 0, 0, False), ("conv", 1, 1, 0, True)],
    [("convT", 2, 1, 1, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("convT", 2, 1, 1, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("convT", 2, 1, 1, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("convT", 2, 1, 1, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
    [("convT", 2, 1, 1, False), ("conv", 1, 1, 0, True), ("conv", 1, 1, 0, True)],
]


def _layer(sd, prefix, spec, x):
    kind, stride, pad, outpad, residual = spec
    w = sd[f"{prefix}.conv_block.0.weight"].to(x.dtype)
    b = sd[f"{prefix}.conv_block.0.bias"].to(x.dtype)
    if kind == "conv":
        y = F.conv2d(x, w, b, stride=stride, padding=pad)
    else:
        y = F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=outpad)
    g = sd[f"{prefix}.conv_block.1.weight"].to(x.dtype)
    beta = sd[f"{prefix}.conv_block.1.bias"].to(x.dtype)
    mu = sd[f"{prefix}.conv_block.1.running_mean"].to(x.dtype)
    var = sd[f"{prefix}.conv_block.1.running_var"].to(x.dtype)
    # eval-mode BatchNorm2d: (y - mu) / sqrt(var + eps) * g + beta
    y = (y - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    y = y * g[None, :, None, None] + beta[None, :, None, None]
    if residual:  # conv.py:17-18
        y = y + x
    return torch.relu(y)


@torch.no_grad()
def wav2lip_forward(sd, audio_sequences, face_sequences, taps=None, dtype=torch.float32):
    """sd: state dict (keys as wav2lip.py's module tree, optional 'module.' prefix stripped
    by the caller as lipreal.py:48-49 does). audio [B,1,80,16], face [B,6,96,96] -> [B,3,96,96].
    `taps`, if a dict, receives intermediate activations keyed by block name."""
    a = audio_sequences.to(dtype)
    x = face_sequences.to(dtype)
    for i, spec in enumerate(AUDIO_ENCODER):
        a = _layer(sd, f"audio_encoder.{i}", spec, a)
    if taps is not None:
        taps["audio_embedding"] = a
    feats = []
    for bi, blk in enumerate(FACE_ENCODER):
        for i, spec in enumerate(blk):
            x = _layer(sd, f"face_encoder_blocks.{bi}.{i}", spec, x)
        feats.append(x)
        if taps is not None:
            taps[f"face_encoder_blocks.{bi}"] = x
    x = a
    for bi, blk in enumerate(FACE_DECODER):
        for i, spec in enumerate(blk):
            x = _layer(sd, f"face_decoder_blocks.{bi}.{i}", spec, x)
        if taps is not None:
            taps[f"face_decoder_blocks.{bi}"] = x
        x = torch.cat((x, feats.pop()), dim=1)  # wav2lip.py:108
    x = _layer(sd, "output_block.0", ("conv", 1, 1, 0, False), x)
    x = F.conv2d(x, sd["output_block.1.weight"].to(dtype), sd["output_block.1.bias"].to(dtype))
    if taps is not None:
        taps["logits"] = x
    return torch.sigmoid(x)


def strip_module_prefix(state_dict):
    """lipreal.py:46-49."""
    return {k.replace("module.", ""): v for k, v in state_dict.items()}
