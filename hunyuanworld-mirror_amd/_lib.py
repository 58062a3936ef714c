"""ctypes binding of libwm_hip.so (the C ABI declared in include/wm_hip.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, importing
this module's ``lib()`` raises.  ``import torch`` must come first so that this library binds to the
HIP runtime / RCCL already loaded by torch-ROCm (same sonames) and device pointers are shared.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads libamdhip64.so.7 / librccl.so.1 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
# WM_HIP_LIB: a diagnostic build of the same library (tools only, e.g. the in-kernel stamp build of csrc/Makefile "stamps")
LIB_PATH = os.environ.get("WM_HIP_LIB") or os.path.join(_HERE, "libwm_hip.so")

WM_DT = {"bf16": 0, "f16": 1, "fp16": 1, "float16": 1, "bfloat16": 0}
EPI_F32, EPI_T16, EPI_GELU_T16, EPI_RESID = 0, 1, 2, 3


class wm_config(C.Structure):
    _fields_ = [
        ("img_size", C.c_int32), ("patch_size", C.c_int32), ("embed_dim", C.c_int32), ("gs_dim", C.c_int32),
        ("enable_cond", C.c_int32), ("enable_cam", C.c_int32), ("enable_pts", C.c_int32), ("enable_depth", C.c_int32),
        ("enable_norm", C.c_int32), ("enable_gs", C.c_int32),
        ("depth", C.c_int32), ("num_heads", C.c_int32), ("mlp_ratio", C.c_int32), ("num_register_tokens", C.c_int32),
        ("intermediate_idxs", C.c_int32 * 4),
        ("rope_freq", C.c_float),
        ("dino_depth", C.c_int32), ("dino_heads", C.c_int32),
        ("cam_trunk_depth", C.c_int32), ("cam_heads", C.c_int32), ("cam_steps", C.c_int32),
        ("dpt_features", C.c_int32),
        ("dpt_out_channels", C.c_int32 * 4),
        ("backbone_dtype", C.c_int32), ("head_dtype", C.c_int32),
    ]


_FP = C.c_void_p  # device float*


class wm_outputs(C.Structure):
    _fields_ = [(n, _FP) for n in (
        "camera_params", "camera_poses", "camera_intrs", "depth", "depth_conf", "pts3d", "pts3d_conf",
        "normals", "normals_conf", "gs_depth", "gs_depth_conf", "splat_means", "splat_quats", "splat_scales",
        "splat_opacities", "splat_sh", "splat_weights")] + [("taps", _FP * 4)]


EXPORTS = [
    "wm_create", "wm_destroy", "wm_last_error", "wm_set_weight", "wm_finalize_weights", "wm_host_resample_pos",
    "wm_workspace_bytes", "wm_reserve", "wm_set_workspace", "wm_missing_name", "wm_share_weights", "wm_forward", "wm_forward_sharded", "wm_rccl_unique_id", "wm_comm_init_rccl",
    "wm_local_group_create", "wm_local_group_destroy", "wm_comm_init_local", "wm_allgather", "wm_profile_enable", "wm_profile_read",
    "wm_op_gemm", "wm_op_gemm_resid_ln", "wm_op_gemm_qkv", "wm_op_attention", "wm_op_layernorm", "wm_op_qkv_post", "wm_op_conv", "wm_op_bilinear",
    "wm_op_linear_f32", "wm_host_to_16", "wm_set_tuning", "wm_op_attention_split", "wm_op_attention_ex", "wm_op_attention_flag_count", "wm_op_gs_splat", "wm_op_conv3x3_up", "wm_depth_to_world", "wm_confidence_mask", "wm_confidence_mask_workspace_bytes", "wm_preprocess_image", "wm_preprocess_image_size",
    "wm_preprocess_image_workspace_bytes", "wm_rasterize_splats", "wm_rasterize_workspace_bytes", "wm_prune_gs", "wm_prune_gs_workspace_bytes", "wm_op_up_conv_n32", "wm_op_conv3x3_gemm16", "wm_op_conv_ex", "wm_op_upconv3x3_tap", "wm_op_tconv", "wm_op_upconv_gather",
]

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.wm_create.argtypes = [C.POINTER(wm_config), i32, C.POINTER(vp)]
    L.wm_destroy.argtypes = [vp]
    L.wm_destroy.restype = None
    L.wm_last_error.argtypes = [vp]
    L.wm_last_error.restype = C.c_char_p
    L.wm_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]
    L.wm_finalize_weights.argtypes = [vp, C.POINTER(i32)]
    L.wm_host_resample_pos.argtypes = [vp, i32, i32, i32, i32, vp]
    L.wm_host_resample_pos.restype = None
    L.wm_workspace_bytes.argtypes = [vp, i32, i32, i32, i32]
    L.wm_workspace_bytes.restype = C.c_size_t
    L.wm_reserve.argtypes = [vp, i32, i32, i32, i32]
    L.wm_set_workspace.argtypes = [vp, vp, C.c_size_t]
    L.wm_share_weights.argtypes = [vp, vp]
    L.wm_missing_name.argtypes = [vp, i32]
    L.wm_missing_name.restype = C.c_char_p
    L.wm_forward.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, C.POINTER(C.c_int32), C.POINTER(wm_outputs), vp]
    L.wm_forward_sharded.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(C.c_int32),
                                     C.POINTER(wm_outputs), vp]
    L.wm_rccl_unique_id.argtypes = [vp]
    L.wm_comm_init_rccl.argtypes = [vp, vp, i32, i32]
    L.wm_local_group_create.argtypes = [i32]
    L.wm_local_group_create.restype = vp
    L.wm_local_group_destroy.argtypes = [vp]
    L.wm_local_group_destroy.restype = None
    L.wm_comm_init_local.argtypes = [vp, vp, i32]
    L.wm_allgather.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.wm_profile_enable.argtypes = [vp, i32]
    L.wm_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.wm_op_gemm.argtypes = [i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.wm_op_gemm_resid_ln.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, C.POINTER(i32), vp]
    L.wm_op_gemm_qkv.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]
    L.wm_op_attention.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.wm_op_attention_split.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.wm_op_attention_ex.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.wm_op_attention_flag_count.argtypes = [i32, i32, i32]
    L.wm_op_attention_flag_count.restype = C.c_size_t
    L.wm_op_layernorm.argtypes = [vp, vp, vp, vp, i32, i32, f32, i32, i32, vp]
    L.wm_op_qkv_post.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]
    L.wm_op_conv.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.wm_op_bilinear.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.wm_op_linear_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.wm_host_to_16.argtypes = [vp, vp, C.c_size_t, i32]
    L.wm_host_to_16.restype = None
    L.wm_depth_to_world.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]
    L.wm_depth_to_world.restype = i32
    L.wm_confidence_mask_workspace_bytes.argtypes = [C.c_size_t]
    L.wm_confidence_mask_workspace_bytes.restype = C.c_size_t
    L.wm_confidence_mask.argtypes = [vp, C.c_size_t, f32, vp, vp, C.c_size_t, vp]
    L.wm_confidence_mask.restype = i32
    L.wm_op_up_conv_n32.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    L.wm_op_up_conv_n32.restype = i32
    L.wm_op_conv3x3_gemm16.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.wm_op_conv3x3_gemm16.restype = i32
    L.wm_op_conv_ex.argtypes = [i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.wm_op_conv_ex.restype = i32
    L.wm_op_upconv3x3_tap.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    L.wm_op_upconv3x3_tap.restype = i32
    L.wm_op_tconv.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    L.wm_op_tconv.restype = i32
    L.wm_op_upconv_gather.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.wm_op_upconv_gather.restype = i32
    L.wm_prune_gs_workspace_bytes.argtypes = [C.c_size_t]
    L.wm_prune_gs_workspace_bytes.restype = C.c_size_t
    L.wm_prune_gs.argtypes = [vp, vp, vp, vp, vp, vp, i32, f32, vp, vp, vp, vp, vp, C.POINTER(i32), vp, C.c_size_t, vp]
    L.wm_prune_gs.restype = i32
    L.wm_rasterize_workspace_bytes.argtypes = [i32, i32, i32, i32, C.c_size_t]
    L.wm_rasterize_workspace_bytes.restype = C.c_size_t
    L.wm_rasterize_splats.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, C.c_size_t, C.c_size_t,
                                      C.POINTER(C.c_ulonglong), vp]
    L.wm_rasterize_splats.restype = i32
    L.wm_preprocess_image_size.argtypes = [i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.wm_preprocess_image_size.restype = i32
    L.wm_preprocess_image_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.wm_preprocess_image_workspace_bytes.restype = C.c_size_t
    L.wm_preprocess_image.argtypes = [vp, i32, i32, i32, i32, vp, vp, C.c_size_t, vp]
    L.wm_preprocess_image.restype = i32
    L.wm_set_tuning.argtypes = [C.c_char_p, i32]
    L.wm_set_tuning.restype = i32
    _lib = L
    return L


def make_config(cfg, backbone_dtype="bf16", head_dtype="f16") -> wm_config:
    c = wm_config()
    for k in ("img_size", "patch_size", "embed_dim", "gs_dim", "depth", "num_heads", "mlp_ratio",
              "num_register_tokens", "dino_depth", "dino_heads", "cam_trunk_depth", "cam_heads", "cam_steps",
              "dpt_features"):
        setattr(c, k, int(getattr(cfg, k)))
    for k in ("enable_cond", "enable_cam", "enable_pts", "enable_depth", "enable_norm", "enable_gs"):
        setattr(c, k, 1 if getattr(cfg, k) else 0)
    c.rope_freq = float(cfg.rope_freq)
    for i in range(4):
        c.intermediate_idxs[i] = int(cfg.intermediate_idxs[i])
        c.dpt_out_channels[i] = int(cfg.dpt_out_channels[i])
    c.backbone_dtype = WM_DT[backbone_dtype]
    c.head_dtype = WM_DT[head_dtype]
    return c


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
