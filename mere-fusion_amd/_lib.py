"""ctypes binding of include/merefusion.h (the C ABI of libmerefusion_hip.so).

There is no CPU fallback anywhere in this package: if the library is missing or a call fails,
a RuntimeError carrying mf_last_error() is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmerefusion_hip.so")

MF_PREC_BF16 = 0
MF_PREC_BF16X3 = 1
MF_PREC_F16Q = 2
PRECISIONS = {"bf16": MF_PREC_BF16, "bf16x3": MF_PREC_BF16X3, "f16q": MF_PREC_F16Q}


class MfTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class MfConv2dDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "cin", "cout", "kh", "kw", "stride_h", "stride_w", "pad_h", "pad_w", "transposed",
        "output_padding", "residual", "act", "in_h", "in_w", "upsample", "pad_hi")]


class MfUnetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("n_blocks", C.c_int),
                ("block_out_channels", C.c_int * 4), ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int),
                ("attention_heads", C.c_int), ("norm_num_groups", C.c_int), ("down_attn", C.c_int * 4),
                ("up_attn", C.c_int * 4), ("sample_size", C.c_int), ("ctx_len", C.c_int)]


class MfNerfFieldConfig(C.Structure):
    _fields_ = [("bound", C.c_float), ("num_levels", C.c_int), ("level_dim", C.c_int), ("base_resolution", C.c_int),
                ("log2_per_level_scale", C.c_float), ("offsets", C.c_int * 33), ("audio_dim", C.c_int), ("geo_feat_dim", C.c_int),
                ("hidden_dim", C.c_int), ("individual_dim", C.c_int), ("exp_eye", C.c_int)]


class MfNerfTorsoConfig(C.Structure):
    _fields_ = [("torso_shrink", C.c_float), ("num_levels", C.c_int), ("level_dim", C.c_int), ("base_resolution", C.c_int),
                ("log2_per_level_scale", C.c_float), ("offsets", C.c_int * 33), ("individual_dim", C.c_int), ("grid_size", C.c_int)]


class MfVaeConfig(C.Structure):
    _fields_ = [("latent_channels", C.c_int), ("out_channels", C.c_int), ("n_blocks", C.c_int),
                ("block_out_channels", C.c_int * 4), ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int),
                ("sample_size", C.c_int), ("scaling_factor", C.c_float)]


class MfWav2Vec2Config(C.Structure):
    _fields_ = [("hidden", C.c_int), ("n_layer", C.c_int), ("n_head", C.c_int), ("ffn", C.c_int), ("vocab", C.c_int), ("n_conv", C.c_int),
                ("conv_dim", C.c_int * 8), ("conv_kernel", C.c_int * 8), ("conv_stride", C.c_int * 8), ("conv_bias", C.c_int),
                ("feat_norm_layer", C.c_int), ("stable_ln", C.c_int), ("pos_k", C.c_int), ("pos_groups", C.c_int), ("layer_norm_eps", C.c_float),
                ("do_normalize", C.c_int), ("out_hidden", C.c_int)]


class MfPasteJob(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("frame_index", "x1", "y1", "x2", "y2", "cx1", "cy1", "cx2", "cy2")] + [("mask", C.c_void_p)]


def tensor_array(state_dict):
    """(ctypes array of MfTensor, keep-alive list) for a {name: fp32 cpu tensor} dict."""
    import torch
    items = [(k.encode(), v.detach().to("cpu", torch.float32).contiguous()) for k, v in state_dict.items()
             if torch.is_tensor(v) and v.is_floating_point()]
    arr = (MfTensor * len(items))()
    for i, (k, v) in enumerate(items):
        arr[i].name, arr[i].data, arr[i].ndim = k, v.data_ptr(), v.dim()
        for d in range(min(v.dim(), 4)):
            arr[i].shape[d] = v.shape[d]
    return arr, items


# every symbol include/merefusion.h declares: (restype, argtypes)
SIGNATURES = {
    "mf_init": (C.c_int, [C.c_int]),
    "mf_last_error": (C.c_char_p, []),
    "mf_abi_version": (C.c_int, []),
    "mf_wav2lip_create": (C.c_int, [C.POINTER(MfTensor), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_wav2lip_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_wav2lip_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_wav2lip_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_wav2lip_num_launches": (C.c_int, [C.c_void_p, C.c_int]),
    "mf_wav2lip_launch_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                        C.POINTER(C.c_double)]),
    "mf_wav2lip_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.c_void_p]),
    "mf_wav2lip_tune": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mf_wav2lip_destroy": (None, [C.c_void_p]),
    "mf_conv2d_create": (C.c_int, [C.POINTER(MfConv2dDesc)] + [C.c_void_p] * 6 + [C.c_int, C.POINTER(C.c_void_p)]),
    "mf_conv2d_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_conv2d_forward_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_conv2d_out_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mf_conv2d_time": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "mf_conv2d_destroy": (None, [C.c_void_p]),
    "mf_whisper_create": (C.c_int, [C.POINTER(MfTensor), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_whisper_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mf_whisper_log_mel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_whisper_encode_audio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_whisper_set_batch": (C.c_int, [C.c_void_p, C.c_int]),
    "mf_whisper_encode_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_whisper_destroy": (None, [C.c_void_p]),
    "mf_wav2vec2_create": (C.c_int, [C.POINTER(MfWav2Vec2Config), C.POINTER(MfTensor), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_wav2vec2_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mf_wav2vec2_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_wav2vec2_destroy": (None, [C.c_void_p]),
    "mf_net_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_net_buffer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mf_net_conv": (C.c_int, [C.c_void_p, C.POINTER(MfConv2dDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "mf_net_maxpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mf_net_l2norm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float]),
    "mf_net_global_avgpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mf_net_scale_add": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mf_net_upsample_nearest": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mf_net_num_ops": (C.c_int, [C.c_void_p]),
    "mf_net_flops_per_item": (C.c_double, [C.c_void_p]),
    "mf_net_set_input": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mf_net_run": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mf_net_tune": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mf_net_get_output": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_net_get_output_bilinear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mf_s3fd_maxout_bg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mf_net_destroy": (None, [C.c_void_p]),
    "mf_probe_mfma_ceiling": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "mf_unet_create": (C.c_int, [C.POINTER(MfUnetConfig), C.POINTER(MfTensor), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_unet_num_ops": (C.c_int, [C.c_void_p]),
    "mf_unet_op_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double)]),
    "mf_unet_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "mf_unet_tune": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mf_unet_destroy": (None, [C.c_void_p]),
    "mf_vae_create": (C.c_int, [C.POINTER(MfVaeConfig), C.POINTER(MfTensor), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_vae_decode_latents": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_vae_num_ops": (C.c_int, [C.c_void_p]),
    "mf_vae_op_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double)]),
    "mf_vae_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "mf_vae_tune": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mf_vae_destroy": (None, [C.c_void_p]),
    "mf_vae_encoder_create": (C.c_int, [C.POINTER(MfVaeConfig), C.POINTER(MfTensor), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_vae_encoder_image_size": (C.c_int, [C.c_void_p]),
    "mf_vae_encoder_destroy": (None, [C.c_void_p]),
    "mf_melspec": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_melspec_frames": (C.c_int, [C.c_int]),
    "mf_attention_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "mf_near_far_from_aabb": (C.c_int, [C.c_void_p] * 3 + [C.c_uint32, C.c_float] + [C.c_void_p] * 3),
    "mf_march_rays": (C.c_int, [C.c_uint32, C.c_uint32] + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
                      + [C.c_void_p] * 8),
    "mf_composite_rays_triplane": (C.c_int, [C.c_uint32, C.c_uint32, C.c_float] + [C.c_void_p] * 15),
    "mf_grid_encode_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_uint32] * 4 + [C.c_float, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "mf_sh_encode_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "mf_freq_encode_forward": (C.c_int, [C.c_void_p] + [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p]),
    "mf_nerf_finish": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mf_nerf_field_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mf_nerf_field_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_int] + [C.c_void_p] * 6),
    "mf_nerf_field_destroy": (None, [C.c_void_p]),
    "mf_nerf_resize_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mf_nerf_head_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mf_nerf_head_render": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                      C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 5),
    "mf_nerf_head_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 5),
    "mf_nerf_head_set_eye": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mf_nerf_head_sums": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4),
    "mf_nerf_head_plan_rounds": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mf_nerf_head_set_rounds": (C.c_int, [C.c_void_p, C.c_int]),
    "mf_nerf_head_last_rounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mf_nerf_head_ctl_snapshot": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "mf_nerf_head_destroy": (None, [C.c_void_p]),
    "mf_nerf_torso_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mf_nerf_torso_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 4),
    "mf_nerf_torso_destroy": (None, [C.c_void_p]),
    "mf_audio_encoder_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mf_audio_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_audio_encoder_forward_smooth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mf_audio_encoder_destroy": (None, [C.c_void_p]),
    "mf_gather_rows_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p]),
    "mf_whisper_feature_chunks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_paste_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(MfPasteJob), C.c_int,
                                  C.c_void_p, C.c_void_p]),
    "mf_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mf_host_unregister": (C.c_int, [C.c_void_p]),
    "mf_copy_d2h_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mf_copy_d2h_2d_async": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "mf_stream_synchronize": (C.c_int, [C.c_void_p]),
    "mf_resize_linear_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is not built. Run `python -m mere_fusion_amd.build` (needs hipcc); "
                "this package has no CPU fallback.")
        # torch first: its wheel carries its own HIP / HSA runtime, and device buffers are torch's.  Loading this library
        # before torch would bring a second runtime (/opt/rocm) into the process, which then sees no device.
        import torch  # noqa: F401
        # MF_LIB_PATH: an alternative build of the same library (kernel ablation studies, tools/halo_ablate.sh)
        l = C.CDLL(os.environ.get("MF_LIB_PATH") or LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"merefusion {what} failed (status {rc}): {msg}")


_inited = set()
_device_inited = False          # read by placement.py: once HIP is up in this process, HIP_VISIBLE_DEVICES can no longer choose its GPU


def init_device(index):
    global _device_inited
    if index not in _inited:
        check(lib().mf_init(int(index)), "mf_init")
        _inited.add(index)
        _device_inited = True
