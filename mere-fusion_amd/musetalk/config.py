"""Product-side configuration of the MuseTalk step (BASELINE.json configs[2]).

The reference reads `./models/musetalk/musetalk.json` and `./models/sd-vae-ft-mse/config.json` (musetalk/utils/utils.py:67-73), which do
not ship with it; this is the public MuseTalk v1 / sd-vae-ft-mse architecture [upstream-knowledge, SURVEY Appendix C] in the key layout
`mere_fusion_amd.weights.make_musetalk_*` and the handles' config structs take.  bench.py, the tests and smoke() build the product from THIS
table; the checker keeps its own copy (tests/test_musetalk.py asserts that the two agree)."""
import ctypes as C

MUSETALK_V1 = dict(
    unet=dict(in_channels=8, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
              cross_attention_dim=384, attention_heads=8, norm_num_groups=32,
              down_attn=(True, True, True, False), up_attn=(False, True, True, True)),
    vae=dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
             norm_num_groups=32, scaling_factor=0.18215),
)


def unet_config_json(u, sample_size=32):
    """The dict `UNet(unet_config=...)` takes (the JSON's keys) from the table above."""
    return dict(in_channels=u["in_channels"], out_channels=u["out_channels"], block_out_channels=list(u["block_out_channels"]),
                layers_per_block=u["layers_per_block"], cross_attention_dim=u["cross_attention_dim"],
                attention_head_dim=u["attention_heads"], norm_num_groups=u["norm_num_groups"], down_attn=u["down_attn"],
                up_attn=u["up_attn"], sample_size=sample_size)


def vae_config_json(v):
    out = dict(v)
    out["block_out_channels"] = list(v["block_out_channels"])
    return out


def algorithmic_flops_per_frame(unet, vae):
    """2 x MACs of every convolution / Linear / attention product of one frame, summed over the handles' own op lists
    (mf_unet_op_info / mf_vae_op_info: `mf_conv_flops` = 2 * sites * cin * cout * taps, attention 4 * Tq * Tk * C; GroupNorm, SiLU,
    softmax and layout kernels count zero).  Returns (unet_flops, vae_flops)."""
    from .. import _lib
    l = _lib.lib()
    out = []
    for h, nops, info in ((unet.model._h, l.mf_unet_num_ops, l.mf_unet_op_info), (vae._h, l.mf_vae_num_ops, l.mf_vae_op_info)):
        tot = 0.0
        for i in range(nops(h)):
            nm, kn, fl = C.create_string_buffer(160), C.create_string_buffer(96), C.c_double()
            _lib.check(info(h, i, nm, 160, kn, 96, C.byref(fl)))
            tot += fl.value
        out.append(tot)
    return tuple(out)
