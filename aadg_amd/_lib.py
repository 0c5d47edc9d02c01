"""ctypes binding of libaadg_hip.so (include/aadg_hip.h) for torch device tensors.

torch is used only as plumbing here: device memory, the current HIP stream, and dtype/shape checks.
There is deliberately NO CPU fallback: every wrapper raises if the shared library or a GPU is
missing, so a silent eager path can never stand in for the HIP kernels.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaadg_hip.so")      # tests / kernel A/B scripts may assign another path before load() (scripts/ab/hook)
MAX_OPS = 4

# mirror of `aadg_unit` (include/aadg_hip.h); 140 bytes, no padding
UNIT_DTYPE = np.dtype([
    ("src", "<i4"), ("n_ops", "<i4"),
    ("op", "<i4", (MAX_OPS,)), ("iarg", "<i4", (MAX_OPS,)), ("farg", "<f4", (MAX_OPS,)),
    ("rect", "<i4", (MAX_OPS, 4)),
    ("scaled_w", "<i4"), ("scaled_h", "<i4"), ("pad", "<i4"), ("crop_x", "<i4"), ("crop_y", "<i4"),
], align=False)
assert UNIT_DTYPE.itemsize == 140

DATASET_OPTIC, DATASET_VESSEL = 0, 1

# every symbol include/aadg_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "aadg_abi_version",
    "aadg_aug_u8_workspace_bytes", "aadg_aug_u8_forward", "aadg_aug_u8_forward_ex", "aadg_aug_u8_forward_ex2", "aadg_op_u8",
    "aadg_pool_histograms_u8",
    "aadg_aug_u8_plan",
    "aadg_sinkhorn_workspace_bytes", "aadg_sinkhorn_divergence_f32", "aadg_sinkhorn_rewards_f32", "aadg_sinkhorn_rewards_norm_f32",
    "aadg_normalize_rewards_f32",
    "aadg_seg_loss_workspace_bytes", "aadg_seg_bce_dice_f32", "aadg_seg_bce_dice_scaled_f32",
    "aadg_fop_workspace_bytes", "aadg_fop_f32",
    "aadg_upsample_bilinear2d", "aadg_upsample_bilinear2d_strided", "aadg_upsample_bilinear2d_backward_supported", "aadg_upsample_bilinear2d_backward",
    "aadg_upsample_bilinear2d_backward_workspace_bytes", "aadg_upsample_bilinear2d_backward_strided",
    "aadg_bn_workspace_bytes", "aadg_bn_mask_bytes", "aadg_bn_forward", "aadg_bn_backward",
    "aadg_bn_sync_forward", "aadg_bn_sync_backward",
    "aadg_layernorm_supported", "aadg_layernorm_workspace_bytes", "aadg_layernorm_forward", "aadg_layernorm_backward",
    "aadg_dwconv3x3_gelu_nhwc_supported", "aadg_dwconv3x3_gelu_nhwc_forward", "aadg_dwconv3x3_gelu_nhwc_backward",
    "aadg_dwconv3x3_supported", "aadg_dwconv3x3_workspace_bytes", "aadg_dwconv3x3", "aadg_dwconv3x3_wgrad",
    "aadg_bn_relu_maxpool_supported", "aadg_bn_relu_maxpool_forward", "aadg_bn_relu_maxpool_backward",
    "aadg_conv1x1_nchw_supported", "aadg_conv1x1_nchw_bf16",
    "aadg_stem_conv7x7_supported", "aadg_stem_conv7x7_workspace_bytes", "aadg_stem_conv7x7_bf16", "aadg_stem_conv7x7_wgrad_bf16",
    "aadg_subsample2x2_supported", "aadg_subsample2x2", "aadg_subsample2x2_backward",
    "aadg_maxpool3x3s2_supported", "aadg_maxpool3x3s2_index_bytes", "aadg_maxpool3x3s2_forward", "aadg_maxpool3x3s2_backward",
    "aadg_conv1x1_wgrad_supported", "aadg_conv1x1_wgrad_bf16", "aadg_conv3x3_wgrad_supported", "aadg_conv3x3_wgrad_bf16",
    "aadg_conv3x3_nchw_supported", "aadg_conv3x3_nchw_bf16", "aadg_conv3x3s2_wgrad_supported", "aadg_conv3x3s2_wgrad_bf16",
    "aadg_conv3x3s2_dgrad_supported", "aadg_conv3x3s2_dgrad_bf16", "aadg_conv3x3s2_nchw_supported", "aadg_conv3x3s2_nchw_bf16",
    "aadg_controller_supported", "aadg_controller_workspace_bytes", "aadg_controller_sample_f32",
    "aadg_controller_ppo_update_f32",
    "aadg_embed_prologue_f32", "aadg_embed_prologue_norm_f32",
    "aadg_upsample_sum_backward_all_supported", "aadg_upsample_sum_backward_all",
    "aadg_upsample_sum", "aadg_upsample_sum_backward",
    "aadg_weight_layouts_bf16",
    "aadg_weight_layouts_split_bf16", "aadg_conv1x1_nchw_f32x3", "aadg_conv1x1_wgrad_f32x3", "aadg_conv3x3_nchw_f32x3",
    "aadg_conv3x3_wgrad_f32x3", "aadg_conv3x3s2_nchw_f32x3", "aadg_conv3x3s2_dgrad_f32x3", "aadg_conv3x3s2_wgrad_f32x3",
    "aadg_stem_conv7x7_f32x3", "aadg_stem_conv7x7_wgrad_f32x3", "aadg_sinkhorn_divergence_phases_f32",
    "aadg_conv1x1_nchw_f32x3_stats", "aadg_conv3x3_nchw_f32x3_stats", "aadg_conv3x3_f32x3_stats_supported",
    "aadg_bn_finalize_f32", "aadg_conv1x1_f32x3_pre_supported", "aadg_conv1x1_nchw_f32x3_pre", "aadg_conv1x1_wgrad_f32x3_pre",
    "aadg_conv3x3_nchw_f32x3_pre", "aadg_conv3x3_wgrad_f32x3_pre", "aadg_conv1x1_wgrad_f32x3_pre_supported", "aadg_bn_forward_res_affine_f32", "aadg_bn_backward_res_bn_f32",
    "aadg_bn_sync_backward_res_bn_f32",
]

_lib = None
_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t


def load():
    """Load libaadg_hip.so; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libaadg_hip.so is missing (%s). Build it with `python -m aadg_amd.build` or "
            "`__graft_entry__.build()`; aadg_amd has no CPU fallback." % LIB_PATH)
    lib = _c.CDLL(LIB_PATH)
    lib.aadg_abi_version.restype = _i
    lib.aadg_aug_u8_workspace_bytes.restype = _sz
    lib.aadg_aug_u8_workspace_bytes.argtypes = [_i, _i, _i, _i]
    lib.aadg_aug_u8_forward.restype = _i
    lib.aadg_aug_u8_forward.argtypes = [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]
    lib.aadg_aug_u8_forward_ex.restype = _i
    lib.aadg_aug_u8_forward_ex.argtypes = lib.aadg_aug_u8_forward.argtypes + [_i, _i, _vp, _vp]
    lib.aadg_aug_u8_forward_ex2.restype = _i
    lib.aadg_aug_u8_forward_ex2.argtypes = lib.aadg_aug_u8_forward_ex.argtypes + [_vp]
    lib.aadg_pool_histograms_u8.restype = _i
    lib.aadg_pool_histograms_u8.argtypes = [_vp, _i, _i, _i, _vp, _vp]
    lib.aadg_aug_u8_plan.restype = _i
    lib.aadg_aug_u8_plan.argtypes = [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]
    lib.aadg_op_u8.restype = _i
    lib.aadg_op_u8.argtypes = [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp]
    lib.aadg_sinkhorn_workspace_bytes.restype = _sz
    lib.aadg_sinkhorn_workspace_bytes.argtypes = [_i, _i, _i]
    lib.aadg_sinkhorn_divergence_f32.restype = _i
    lib.aadg_sinkhorn_divergence_f32.argtypes = [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _sz, _vp]
    lib.aadg_sinkhorn_rewards_f32.restype = _i
    lib.aadg_sinkhorn_rewards_f32.argtypes = [_vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _sz, _vp]
    lib.aadg_sinkhorn_rewards_norm_f32.restype = _i
    lib.aadg_sinkhorn_rewards_norm_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _sz, _vp]
    lib.aadg_normalize_rewards_f32.restype = _i
    lib.aadg_normalize_rewards_f32.argtypes = [_vp, _i, _vp, _vp]
    if hasattr(lib, "aadg_seg_bce_dice_f32"):
        lib.aadg_seg_loss_workspace_bytes.restype = _sz
        lib.aadg_seg_loss_workspace_bytes.argtypes = [_i, _i, _i]
        lib.aadg_seg_bce_dice_f32.restype = _i
        lib.aadg_seg_bce_dice_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]
        lib.aadg_seg_bce_dice_scaled_f32.restype = _i
        lib.aadg_seg_bce_dice_scaled_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _vp, _vp, _sz, _vp]
    if hasattr(lib, "aadg_fop_f32"):
        lib.aadg_fop_workspace_bytes.restype = _sz
        lib.aadg_fop_workspace_bytes.argtypes = [_i, _i, _i]
        lib.aadg_fop_f32.restype = _i
        lib.aadg_fop_f32.argtypes = [_i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]
    lib.aadg_upsample_bilinear2d.restype = _i
    lib.aadg_upsample_bilinear2d.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.aadg_upsample_bilinear2d_strided.restype = _i
    lib.aadg_upsample_bilinear2d_strided.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _i, _c.c_longlong, _i, _vp]
    lib.aadg_upsample_bilinear2d_backward_supported.restype = _i
    lib.aadg_upsample_bilinear2d_backward_supported.argtypes = [_i, _i, _i, _i]
    lib.aadg_upsample_bilinear2d_backward.restype = _i
    lib.aadg_upsample_bilinear2d_backward.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]
    lib.aadg_upsample_bilinear2d_backward_strided.restype = _i
    lib.aadg_upsample_bilinear2d_backward_strided.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _i, _c.c_longlong, _i, _vp, _sz, _vp]
    lib.aadg_upsample_bilinear2d_backward_workspace_bytes.restype = _sz
    lib.aadg_upsample_bilinear2d_backward_workspace_bytes.argtypes = [_i, _i]
    lib.aadg_bn_workspace_bytes.restype = _sz
    lib.aadg_bn_workspace_bytes.argtypes = [_i]
    lib.aadg_bn_forward.restype = _i
    lib.aadg_bn_mask_bytes.restype = _sz
    lib.aadg_bn_mask_bytes.argtypes = [_i, _i, _i, _i]
    lib.aadg_bn_forward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _c.c_longlong, _vp]
    lib.aadg_bn_backward.restype = _i
    lib.aadg_bn_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _c.c_longlong, _vp]
    lib.aadg_dwconv3x3_gelu_nhwc_supported.restype = _i
    lib.aadg_dwconv3x3_gelu_nhwc_supported.argtypes = [_i, _i, _i, _i, _i]
    lib.aadg_dwconv3x3_gelu_nhwc_forward.restype = _i
    lib.aadg_dwconv3x3_gelu_nhwc_forward.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_dwconv3x3_gelu_nhwc_backward.restype = _i
    lib.aadg_dwconv3x3_gelu_nhwc_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_layernorm_supported.restype = _i
    lib.aadg_layernorm_supported.argtypes = [_i, _i, _i]
    lib.aadg_layernorm_workspace_bytes.restype = _sz
    lib.aadg_layernorm_workspace_bytes.argtypes = [_i, _i]
    lib.aadg_layernorm_forward.restype = _i
    lib.aadg_layernorm_forward.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]
    lib.aadg_layernorm_backward.restype = _i
    lib.aadg_layernorm_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]
    lib.aadg_bn_sync_forward.restype = _i
    lib.aadg_bn_sync_forward.argtypes = [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                         _c.c_longlong, _vp]
    lib.aadg_bn_sync_backward.restype = _i
    lib.aadg_bn_sync_backward.argtypes = [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                          _vp, _vp, _vp, _sz, _c.c_longlong, _vp]
    lib.aadg_dwconv3x3_supported.restype = _i
    lib.aadg_dwconv3x3_supported.argtypes = [_i, _i, _i, _i]
    lib.aadg_dwconv3x3_workspace_bytes.restype = _sz
    lib.aadg_dwconv3x3_workspace_bytes.argtypes = [_i]
    lib.aadg_dwconv3x3.restype = _i
    lib.aadg_dwconv3x3.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]
    lib.aadg_dwconv3x3_wgrad.restype = _i
    lib.aadg_dwconv3x3_wgrad.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]
    lib.aadg_bn_relu_maxpool_supported.restype = _i
    lib.aadg_bn_relu_maxpool_supported.argtypes = [_i, _i, _i]
    lib.aadg_bn_relu_maxpool_forward.restype = _i
    lib.aadg_bn_relu_maxpool_forward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]
    lib.aadg_bn_relu_maxpool_backward.restype = _i
    lib.aadg_bn_relu_maxpool_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]
    lib.aadg_conv1x1_nchw_supported.restype = _i
    lib.aadg_conv1x1_nchw_supported.argtypes = [_i, _i, _i]
    lib.aadg_conv1x1_nchw_bf16.restype = _i
    lib.aadg_conv1x1_nchw_bf16.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_stem_conv7x7_supported.restype = _i
    lib.aadg_stem_conv7x7_supported.argtypes = [_i, _i]
    lib.aadg_stem_conv7x7_workspace_bytes.restype = ctypes.c_size_t
    lib.aadg_stem_conv7x7_workspace_bytes.argtypes = []
    lib.aadg_stem_conv7x7_bf16.restype = _i
    lib.aadg_stem_conv7x7_bf16.argtypes = [_vp, _i, _vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]
    lib.aadg_stem_conv7x7_wgrad_bf16.restype = _i
    lib.aadg_stem_conv7x7_wgrad_bf16.argtypes = [_vp, _i, _vp, _vp, _i, _i, _i, _vp]
    lib.aadg_subsample2x2_supported.restype = _i
    lib.aadg_subsample2x2_supported.argtypes = [_i, _i, _i]
    lib.aadg_subsample2x2.restype = _i
    lib.aadg_subsample2x2.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_subsample2x2_backward.restype = _i
    lib.aadg_subsample2x2_backward.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_maxpool3x3s2_supported.restype = _i
    lib.aadg_maxpool3x3s2_supported.argtypes = [_i, _i]
    lib.aadg_maxpool3x3s2_forward.restype = _i
    lib.aadg_maxpool3x3s2_index_bytes.restype = ctypes.c_size_t
    lib.aadg_maxpool3x3s2_index_bytes.argtypes = [_i, _i, _i]
    lib.aadg_maxpool3x3s2_forward.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_maxpool3x3s2_backward.restype = _i
    lib.aadg_maxpool3x3s2_backward.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_conv1x1_wgrad_supported.restype = _i
    lib.aadg_conv1x1_wgrad_supported.argtypes = [_i, _i, _i]
    lib.aadg_conv1x1_wgrad_bf16.restype = _i
    lib.aadg_conv1x1_wgrad_bf16.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_conv3x3_wgrad_supported.restype = _i
    lib.aadg_conv3x3_wgrad_supported.argtypes = [_i, _i, _i, _i, _i]
    lib.aadg_conv3x3_wgrad_bf16.restype = _i
    lib.aadg_conv3x3_wgrad_bf16.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.aadg_conv3x3s2_wgrad_supported.restype = _i
    lib.aadg_conv3x3s2_wgrad_supported.argtypes = [_i, _i, _i, _i]
    lib.aadg_conv3x3s2_wgrad_bf16.restype = _i
    lib.aadg_conv3x3s2_wgrad_bf16.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_conv3x3s2_dgrad_supported.restype = _i
    lib.aadg_conv3x3s2_dgrad_supported.argtypes = [_i, _i, _i, _i]
    lib.aadg_conv3x3s2_dgrad_bf16.restype = _i
    lib.aadg_conv3x3s2_dgrad_bf16.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_weight_layouts_bf16.restype = _i
    lib.aadg_weight_layouts_bf16.argtypes = [_vp, _vp, _i, _vp]
    lib.aadg_conv3x3s2_nchw_supported.restype = _i
    lib.aadg_conv3x3s2_nchw_supported.argtypes = [_i, _i, _i, _i]
    lib.aadg_conv3x3s2_nchw_bf16.restype = _i
    lib.aadg_conv3x3s2_nchw_bf16.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_conv3x3_nchw_supported.restype = _i
    lib.aadg_conv3x3_nchw_supported.argtypes = [_i, _i, _i, _i, _i]
    lib.aadg_conv3x3_nchw_bf16.restype = _i
    lib.aadg_conv3x3_nchw_bf16.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.aadg_controller_supported.restype = _i
    lib.aadg_controller_supported.argtypes = [_i] * 7
    lib.aadg_controller_workspace_bytes.restype = _sz
    lib.aadg_controller_workspace_bytes.argtypes = [_i] * 7
    lib.aadg_controller_sample_f32.restype = _i
    lib.aadg_controller_sample_f32.argtypes = [_vp] + [_i] * 7 + [_f] + [_vp] * 7 + [_sz, _vp]
    lib.aadg_controller_ppo_update_f32.restype = _i
    lib.aadg_controller_ppo_update_f32.argtypes = [_vp] * 3 + [_i] * 7 + [_f] + [_vp] * 3 + [_f, _i, _i, _f, _f, _f, _f, _vp, _vp, _sz, _vp]
    lib.aadg_embed_prologue_f32.restype = _i
    lib.aadg_embed_prologue_f32.argtypes = [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp]
    lib.aadg_upsample_sum.restype = _i
    lib.aadg_upsample_sum.argtypes = [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_upsample_sum_backward.restype = _i
    lib.aadg_upsample_sum_backward.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.aadg_upsample_sum_backward_all_supported.restype = _i
    lib.aadg_upsample_sum_backward_all_supported.argtypes = [_i, _i, _vp, _vp, _i]
    lib.aadg_upsample_sum_backward_all.restype = _i
    lib.aadg_upsample_sum_backward_all.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_embed_prologue_norm_f32.restype = _i
    lib.aadg_embed_prologue_norm_f32.argtypes = [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]
    lib.aadg_weight_layouts_split_bf16.restype = _i
    lib.aadg_weight_layouts_split_bf16.argtypes = [_vp, _vp, _i, _vp]
    lib.aadg_conv1x1_nchw_f32x3.restype = _i
    lib.aadg_conv1x1_nchw_f32x3.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_conv1x1_nchw_f32x3_stats.restype = _i
    lib.aadg_conv1x1_nchw_f32x3_stats.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]
    lib.aadg_conv1x1_wgrad_f32x3.restype = _i
    lib.aadg_conv1x1_wgrad_f32x3.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp]
    lib.aadg_bn_finalize_f32.restype = _i
    lib.aadg_bn_finalize_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]
    lib.aadg_conv1x1_f32x3_pre_supported.restype = _i
    lib.aadg_conv1x1_f32x3_pre_supported.argtypes = [_i, _i, _i]
    lib.aadg_conv1x1_nchw_f32x3_pre.restype = _i
    lib.aadg_conv1x1_nchw_f32x3_pre.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]
    lib.aadg_conv1x1_wgrad_f32x3_pre.restype = _i
    lib.aadg_conv1x1_wgrad_f32x3_pre.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.aadg_bn_forward_res_affine_f32.restype = _i
    lib.aadg_bn_forward_res_affine_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]
    lib.aadg_bn_backward_res_bn_f32.restype = _i
    lib.aadg_bn_backward_res_bn_f32.argtypes = [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _i, _i, _i, _vp, _sz, _vp, _sz, _c.c_longlong, _vp]
    lib.aadg_bn_sync_backward_res_bn_f32.restype = _i
    lib.aadg_bn_sync_backward_res_bn_f32.argtypes = [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                     _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _sz, _c.c_longlong, _vp]
    lib.aadg_conv1x1_wgrad_f32x3_pre_supported.restype = _i
    lib.aadg_conv1x1_wgrad_f32x3_pre_supported.argtypes = [_i, _i, _i, _i]
    lib.aadg_conv3x3_nchw_f32x3_pre.restype = _i
    lib.aadg_conv3x3_nchw_f32x3_pre.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]
    lib.aadg_conv3x3_wgrad_f32x3_pre.restype = _i
    lib.aadg_conv3x3_wgrad_f32x3_pre.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]
    lib.aadg_conv3x3_nchw_f32x3.restype = _i
    lib.aadg_conv3x3_nchw_f32x3.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.aadg_conv3x3_nchw_f32x3_stats.restype = _i
    lib.aadg_conv3x3_nchw_f32x3_stats.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]
    lib.aadg_conv3x3_f32x3_stats_supported.restype = _i
    lib.aadg_conv3x3_f32x3_stats_supported.argtypes = [_i, _i, _i, _i, _i]
    lib.aadg_conv3x3_wgrad_f32x3.restype = _i
    lib.aadg_conv3x3_wgrad_f32x3.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    lib.aadg_sinkhorn_divergence_phases_f32.restype = _i
    lib.aadg_sinkhorn_divergence_phases_f32.argtypes = [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _sz, _i, _vp]
    lib.aadg_stem_conv7x7_f32x3.restype = _i
    lib.aadg_stem_conv7x7_f32x3.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]
    lib.aadg_stem_conv7x7_wgrad_f32x3.restype = _i
    lib.aadg_stem_conv7x7_wgrad_f32x3.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp]
    lib.aadg_conv3x3s2_nchw_f32x3.restype = _i
    lib.aadg_conv3x3s2_nchw_f32x3.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_conv3x3s2_dgrad_f32x3.restype = _i
    lib.aadg_conv3x3s2_dgrad_f32x3.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    lib.aadg_conv3x3s2_wgrad_f32x3.restype = _i
    lib.aadg_conv3x3s2_wgrad_f32x3.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    if lib.aadg_abi_version() != 11:
        raise RuntimeError("libaadg_hip.so ABI version mismatch")
    _lib = lib
    return lib


class AadgError(RuntimeError):
    pass


def _check(rc, what):
    if rc == 0:
        return
    if rc == -1:
        raise AadgError("%s: bad argument" % what)
    if rc == -2:
        raise AadgError("%s: workspace too small" % what)
    if rc == -3:
        raise AadgError("%s: unsupported size for this kernel" % what)
    raise AadgError("%s: HIP error %d" % (what, rc))


def _require_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AadgError("aadg_amd kernels need GPU tensors (got %s); there is no CPU path" % t.device)
        if not t.is_contiguous():
            raise AadgError("aadg_amd kernels need contiguous tensors")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the hipStream_t torch currently launches on (every aadg_* entry point is given it).  torch.cuda.current_stream() builds a
    Stream object per call (~8 us, ~10 calls per hot-path step); the raw getter returns the handle itself."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_ws_cache = {}
_zws_cache = {}


def _zeroed_workspace(nbytes, device, tag):
    """Scratch whose CONTENT is part of a kernel's contract: zero-filled when handed out for the first time, and every call leaves it
    zero-filled (aadg_seg_bce_dice_*: integer accumulators and an arrival counter).  One buffer per (device, stream, tag, size): two
    streams must not share accumulators."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(), tag, int(nbytes))
    buf = _zws_cache.get(key)
    if buf is None:
        buf = _zws_cache[key] = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
    return buf



def workspace(nbytes, device, tag="default"):
    """Caller-owned scratch (the library itself never allocates). Cached per (device, tag)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
def units_to_device(units, device):
    """numpy UNIT_DTYPE[N] -> uint8 device tensor [N,140]."""
    units = np.ascontiguousarray(units, dtype=UNIT_DTYPE)
    host = torch.from_numpy(units.view(np.uint8).reshape(units.shape[0], UNIT_DTYPE.itemsize))
    return host.to(device, non_blocking=False)


def validate_units(units, P, Hs, Ws):
    """Host-side argument checks the kernels rely on (raises like the reference's asserts would)."""
    units = np.asarray(units)
    if units.dtype != UNIT_DTYPE:
        raise AadgError("units must have UNIT_DTYPE")
    if units.shape[0] == 0:
        raise AadgError("empty unit list")
    if (units["src"] < 0).any() or (units["src"] >= P).any():
        raise AadgError("unit.src out of range")
    if (units["n_ops"] < 0).any() or (units["n_ops"] > MAX_OPS).any():
        raise AadgError("unit.n_ops out of range (CONTROLLER.L <= %d)" % MAX_OPS)
    if (units["scaled_w"] * 3 < Ws).any() or (units["scaled_h"] * 3 < Hs).any() or \
            (units["scaled_w"] < 1).any() or (units["scaled_h"] < 1).any():
        raise AadgError("scale factor below 1/3 is not supported by the 8-tap resampler")
    for k in range(MAX_OPS):
        live = units["n_ops"] > k
        ops = units["op"][:, k][live]
        if ((ops < 0) | (ops > 9)).any():
            raise AadgError("unknown op id")
        r = units["rect"][:, k][live & (units["op"][:, k] == 9)]
        if r.size and ((r[:, 0] < 0).any() or (r[:, 1] < 0).any() or (r[:, 2] >= Ws).any() or (r[:, 3] >= Hs).any()):
            raise AadgError("cutout rectangle must be clipped to the image")
        b = units["iarg"][:, k][live & (units["op"][:, k] == 4)]
        if b.size and ((b < 0).any() or (b > 8).any()):
            raise AadgError("posterize bits out of range")
    return int(units["n_ops"].max())


def launch_plan(units, Hs, Ws, crop):
    """(classes, stats_mask, order, counts, stat_lists, late, n_stat_stencil, n_generic_wonly) for aadg_aug_u8_forward_ex2 -- mirrors unit_flow() in
    csrc/aug_u8.hip.  n_generic_wonly (ABI 9): the first ones of the generic run shrink the width only and chain no stencil (one-pass tile).
    order: unit indices grouped by tile class (plain up-scaling, up-scaling with a Sharpness stencil, generic, staged);
    counts = (n_plain, n_sharp, n_generic, n_generic_sharp: the last ones of the generic run chain a Sharpness stencil); stat_lists[k]: the units whose k-th op needs a pixel pass for its image statistics;
    late: the units with such an op in a slot k >= 1."""
    n_ops = units["n_ops"]
    live = np.arange(MAX_OPS)[None, :] < n_ops[:, None]
    sharp = ((units["op"] == 8) & (units["farg"] != np.float32(1.0)) & live).sum(axis=1)
    ok = sharp <= 2
    if (Ws & 3) or (crop & 3):
        ok[:] = False
    up = ok & (units["scaled_w"] >= Ws) & (units["scaled_h"] >= Hs)
    generic = ok & ~up & (2 * units["scaled_w"] >= Ws) & (2 * units["scaled_h"] >= Hs)
    staged = ~(up | generic)
    classes = (1 if up.any() else 0) | (2 if staged.any() else 0) | (4 if generic.any() else 0)
    needs = np.isin(units["op"], (0, 2, 5)) & live
    # statistics by push-forward (csrc/aug_u8.hip: stats_by_pushforward): AutoContrast / Equalize in slot k >= 1 behind per-channel
    # byte maps only, fused-flow units -- k_lut derives that stage's histogram from the RAW image's, no pixel pass
    lut_class = np.isin(units["op"], (0, 1, 2, 3, 4, 5, 7))
    prefix_lut = np.ones_like(live)
    for k in range(1, MAX_OPS):
        prefix_lut[:, k] = prefix_lut[:, k - 1] & lut_class[:, k - 1]
    push = live & np.isin(units["op"], (0, 2)) & prefix_lut & (up | generic)[:, None]
    push[:, 0] = False
    pixel_pass = needs & ~push
    pixel_pass[:, 0] |= push.any(axis=1)                   # the raw histogram is the source of every push-forward
    stats_mask = 0
    for k in range(MAX_OPS):
        if pixel_pass[:, k].any():
            stats_mask |= 1 << k
    # tile classes in list order: up-scaling plain / with a stencil, down-scaling ("generic") plain / with a stencil, staged
    # (among the generic units without a stencil those that shrink the width only come first: k_fused3w's list, ABI 9)
    wonly = generic & (sharp == 0) & (units["scaled_h"] >= Hs) & (Ws >= 8)
    cls = np.where(up & (sharp == 0), 0, np.where(up, 1, np.where(wonly, 2, np.where(generic & (sharp == 0), 3, np.where(generic, 4, 5)))))
    order = np.argsort(cls, kind="stable").astype(np.int32)
    counts = (int((cls == 0).sum()), int((cls == 1).sum()), int(((cls == 2) | (cls == 3) | (cls == 4)).sum()), int((cls == 4).sum()))
    # work lists of the histogram kernels; slot k's list starts with the units that have a Sharpness stencil among ops [0, k) (ABI 7:
    # aadg_aug_lists.n_stat_stencil -- their tiles get a workgroup each), both parts in ascending unit order
    stencil = (units["op"] == 8) & (units["farg"] != np.float32(1.0)) & live
    stat_lists, n_stencil = [], []
    for k in range(MAX_OPS):
        idx = np.nonzero(pixel_pass[:, k])[0]
        before = stencil[idx, :k].any(axis=1) if k > 0 else np.zeros(idx.size, bool)
        stat_lists.append(np.concatenate([idx[before], idx[~before]]).astype(np.int32))
        n_stencil.append(int(before.sum()))
    # "late" units: a slot k >= 1 needs a pixel pass (include/aadg_hip.h: aadg_aug_lists.late_units)
    late = np.nonzero(pixel_pass[:, 1:].any(axis=1))[0].astype(np.int32)
    return classes, stats_mask, order, counts, stat_lists, late, n_stencil, int((cls == 2).sum())


def launch_hints(units, Hs, Ws, crop):
    """launch_plan without the late list: (classes, stats_mask, order, counts, stat_lists)."""
    return launch_plan(units, Hs, Ws, crop)[:5]


class AugLists(ctypes.Structure):
    """mirror of `aadg_aug_lists` (include/aadg_hip.h): host struct of device index arrays"""
    _fields_ = [("order", ctypes.c_void_p), ("n_plain", ctypes.c_int32), ("n_sharp", ctypes.c_int32), ("n_generic", ctypes.c_int32),
                ("stat_units", ctypes.c_void_p * MAX_OPS), ("n_stat", ctypes.c_int32 * MAX_OPS), ("pool_hist", ctypes.c_void_p),
                ("late_units", ctypes.c_void_p), ("n_late", ctypes.c_int32), ("n_generic_sharp", ctypes.c_int32),
                ("n_stat_stencil", ctypes.c_int32 * MAX_OPS), ("gen_chunk", ctypes.c_int32), ("n_generic_wonly", ctypes.c_int32)]


HIST_STRIDE = 772      # AADG_HIST_STRIDE


def pool_histograms(pool):
    """uint32 [P, HIST_STRIDE] statistics of the source pool (uint8 [P,H,W,3], device): per channel histogram + sum of L.
    The policy ops run on the raw source image, so these serve every unit / batch that draws the image: compute once per
    resident pool and pass as `pool_hist` to aug_u8_forward (recompute after writing to the pool)."""
    lib = load()
    _require_cuda(pool)
    if pool.dtype != torch.uint8 or pool.dim() != 4 or pool.shape[3] != 3 or not pool.is_contiguous():
        raise AadgError("pool must be contiguous uint8 [P,H,W,3]")
    P, Hs, Ws, _ = pool.shape
    hist = torch.empty((P, HIST_STRIDE), dtype=torch.int32, device=pool.device)
    _check(lib.aadg_pool_histograms_u8(pool.data_ptr(), P, Hs, Ws, hist.data_ptr(), _stream()), "aadg_pool_histograms_u8")
    return hist


# optional (start, stop) torch.cuda.Event pair recorded around the dominant kernel of the next
# aug_u8_forward call(s); used by bench.py to time that kernel live on the launch stream
PROFILE_EVENTS = None
# optional list: every aug_u8_forward call appends the op mix of its units (bench.py: the tile kernel's duration follows it)
PROFILE_MIX = None
# optional (start, stop) torch.cuda.Event pair recorded on the current stream around the WHOLE library call (all its kernels)
PROFILE_CALL_EVENTS = None
_pinned = {}


_REC = UNIT_DTYPE.itemsize + 4 * (2 + MAX_OPS)   # staging bytes per unit: the record + its slot in the class-order list and in each
                                                 # stage's statistics work list


def _pinned_units(n):
    buf = _pinned.get("units")
    if buf is None or buf.numel() < n * _REC:
        buf = torch.empty(max(n, 256) * _REC, dtype=torch.uint8).pin_memory()
        _pinned["units"] = buf
    return buf


def aug_u8_forward(pool, masks, units, crop, dataset, out_img=None, out_lbl=None, pool_hist=None, gen_chunk=0):
    """pool u8 [P,Hs,Ws,3], masks u8 [P,Hs,Ws] (device), units numpy UNIT_DTYPE[N]; pool_hist: pool_histograms(pool) or None
    (None: the statistics passes of the call read the source images themselves); gen_chunk: aadg_aug_lists.gen_chunk (0 = default).
    Returns (aug_images f32 [N,3,crop,crop], aug_labels f32 [N,K,crop,crop]) on the device."""
    lib = load()
    _require_cuda(pool, masks)
    if pool.dtype != torch.uint8 or masks.dtype != torch.uint8 or pool.dim() != 4 or pool.shape[3] != 3:
        raise AadgError("pool must be uint8 [P,H,W,3] and masks uint8 [P,H,W]")
    P, Hs, Ws, _ = pool.shape
    if tuple(masks.shape) != (P, Hs, Ws):
        raise AadgError("masks shape must match pool")
    units = np.asarray(units)
    if units.dtype != UNIT_DTYPE:
        raise AadgError("units must have UNIT_DTYPE")
    N = units.shape[0]
    if N == 0:
        raise AadgError("empty unit list")
    K = 2 if dataset == DATASET_OPTIC else 1
    dev = pool.device
    if out_img is None:
        out_img = torch.empty((N, 3, crop, crop), dtype=torch.float32, device=dev)
    if out_lbl is None:
        out_lbl = torch.empty((N, K, crop, crop), dtype=torch.float32, device=dev)
    _require_cuda(out_img, out_lbl)
    # units: host records -> pinned staging -> async H2D on the launch stream (no host sync)
    units = np.ascontiguousarray(units)
    stage = _pinned_units(N)
    ready = _pinned.get("units_ready")
    if ready is not None:
        ready.synchronize()          # previous copy out of the staging buffer has completed
    nb_units = N * UNIT_DTYPE.itemsize                       # a multiple of 4: the int32 lists behind it are aligned
    host = stage[:N * _REC].numpy()
    host[:nb_units] = units.view(np.uint8).reshape(-1)
    # validation + work lists (tile-class order, per-slot statistics lists, late list) by the library's host-side planner, written
    # straight into the staging buffer behind the records: [order N][stat_units MAX_OPS x N][late N] int32
    base = stage.data_ptr()
    summary = (ctypes.c_int32 * (9 + 2 * MAX_OPS))()
    rc = lib.aadg_aug_u8_plan(base, N, P, Hs, Ws, crop, base + nb_units, base + nb_units + 4 * N, base + nb_units + 4 * N * (1 + MAX_OPS), summary)
    if rc != 0:
        validate_units(units, P, Hs, Ws)                     # raises with the reason
        _check(rc, "aadg_aug_u8_plan")
    n_plain, n_sharp, n_generic, n_generic_sharp, n_late, classes, stats_mask, max_ops = summary[:8]
    d_units = torch.empty(N * _REC, dtype=torch.uint8, device=dev)
    lists = AugLists()
    lists.order = d_units.data_ptr() + nb_units
    lists.n_plain, lists.n_sharp, lists.n_generic, lists.n_generic_sharp = n_plain, n_sharp, n_generic, n_generic_sharp
    lists.n_generic_wonly = summary[8 + 2 * MAX_OPS]
    lists.gen_chunk = int(gen_chunk)
    for k in range(MAX_OPS):
        lists.stat_units[k] = d_units.data_ptr() + nb_units + 4 * N * (1 + k)
        lists.n_stat[k] = summary[8 + k]
        lists.n_stat_stencil[k] = summary[8 + MAX_OPS + k]
    if pool_hist is not None:
        if pool_hist.dtype != torch.int32 or tuple(pool_hist.shape) != (P, HIST_STRIDE) or not pool_hist.is_cuda:
            raise AadgError("pool_hist must be pool_histograms(pool): int32 [P, %d] on the device" % HIST_STRIDE)
        lists.pool_hist = pool_hist.data_ptr()
        lists.late_units = d_units.data_ptr() + nb_units + 4 * N * (1 + MAX_OPS)
        lists.n_late = n_late
    d_units.copy_(stage[:N * _REC], non_blocking=True)          # records + work lists: one H2D copy
    ready = torch.cuda.Event()
    ready.record()
    _pinned["units_ready"] = ready
    nb = lib.aadg_aug_u8_workspace_bytes(N, Hs, Ws, crop)
    ws = workspace(nb, dev, "aug")
    ev0 = ev1 = 0
    if PROFILE_MIX is not None:
        live = np.arange(MAX_OPS)[None, :] < units["n_ops"][:, None]
        PROFILE_MIX.append({"units": int(N), "ops": int(live.sum()), "sharpness_ops": int(((units["op"] == 8) & live).sum()),
                            "sharpness_units": int(n_sharp), "stat_ops": int(sum(summary[8:8 + MAX_OPS])), "late_units": int(n_late),
                            "upscaled": int(((units["scaled_w"] != Ws) | (units["scaled_h"] != Hs)).sum())})
    if PROFILE_EVENTS is not None:
        ev0, ev1 = PROFILE_EVENTS[0].cuda_event, PROFILE_EVENTS[1].cuda_event
    if PROFILE_CALL_EVENTS is not None:
        PROFILE_CALL_EVENTS[0].record()
    rc = lib.aadg_aug_u8_forward_ex2(pool.data_ptr(), masks.data_ptr(), P, Hs, Ws, d_units.data_ptr(), N, max_ops, crop,
                                     dataset, out_img.data_ptr(), out_lbl.data_ptr(), ws.data_ptr(), ws.numel(), _stream(),
                                     classes, stats_mask, ev0, ev1, ctypes.byref(lists))
    if PROFILE_CALL_EVENTS is not None:
        PROFILE_CALL_EVENTS[1].record()
    _check(rc, "aadg_aug_u8_forward")
    d_units.record_stream(torch.cuda.current_stream())
    return out_img, out_lbl


def op_u8(img, op, iarg=0, farg=0.0, rect=None):
    """One registry op on a uint8 HWC device image."""
    lib = load()
    _require_cuda(img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise AadgError("img must be uint8 [H,W,3]")
    H, W, _ = img.shape
    out = torch.empty_like(img)
    r = (ctypes.c_int32 * 4)(*(rect if rect is not None else (0, 0, -1, -1)))
    nb = lib.aadg_aug_u8_workspace_bytes(1, H, W, 0)
    ws = workspace(nb, img.device, "op")
    rc = lib.aadg_op_u8(img.data_ptr(), out.data_ptr(), H, W, int(op), int(iarg), float(farg),
                        ctypes.cast(r, _vp), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_op_u8")
    return out


# ------------------------------------------------------------------------------------------------
def sinkhorn_rewards(fe, D, B, M, blur=0.05, scaling=0.5, rewards=None, row_norm=None):
    """fe f32 [D*B*M, E] in collate order (row (b*D+d)*M + j). rewards[j] += sum_pairs S(x_d1, x_d2).
    row_norm (optional, f32 [D*B*M]): |fe[n]| from embed_prologue(..., want_norm=True); the kernel then skips its norm pass."""
    lib = load()
    _require_cuda(fe, rewards, row_norm)
    if fe.dtype != torch.float32 or fe.dim() != 2 or fe.shape[0] != D * B * M:
        raise AadgError("fe must be float32 [D*B*M, E]")
    if rewards is None:
        rewards = torch.zeros(M, dtype=torch.float32, device=fe.device)
    P = D * (D - 1) // 2
    nb = lib.aadg_sinkhorn_workspace_bytes(M * P, B, fe.shape[1])
    ws = workspace(nb, fe.device, "sinkhorn")
    if row_norm is not None:
        if row_norm.dtype != torch.float32 or row_norm.numel() != fe.shape[0]:
            raise AadgError("row_norm must be float32 [D*B*M]")
        rc = lib.aadg_sinkhorn_rewards_norm_f32(fe.data_ptr(), row_norm.data_ptr(), D, B, M, fe.shape[1], blur, scaling,
                                                rewards.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    else:
        rc = lib.aadg_sinkhorn_rewards_f32(fe.data_ptr(), D, B, M, fe.shape[1], blur, scaling, rewards.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_sinkhorn_rewards_f32")
    return rewards


def sinkhorn_divergence(feat, cloud_rows, cloud_off, prob_xy, max_cloud, blur=0.05, scaling=0.5):
    """General form: index tables (int32 device tensors) into feat f32 [rows, E]; returns f32 [n_prob]."""
    lib = load()
    _require_cuda(feat, cloud_rows, cloud_off, prob_xy)
    if feat.dtype != torch.float32 or feat.dim() != 2:
        raise AadgError("feat must be float32 [rows, E]")
    for t in (cloud_rows, cloud_off, prob_xy):
        if t.dtype != torch.int32:
            raise AadgError("index tables must be int32")
    n_prob = prob_xy.numel() // 2
    out = torch.empty(n_prob, dtype=torch.float32, device=feat.device)
    nb = lib.aadg_sinkhorn_workspace_bytes(n_prob, int(max_cloud), feat.shape[1])
    ws = workspace(nb, feat.device, "sinkhorn")
    rc = lib.aadg_sinkhorn_divergence_f32(feat.data_ptr(), feat.stride(0), feat.shape[1], cloud_rows.data_ptr(),
                                          cloud_off.data_ptr(), prob_xy.data_ptr(), n_prob, int(max_cloud), blur,
                                          scaling, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_sinkhorn_divergence_f32")
    return out


def sinkhorn_divergence_phases(feat, cloud_rows, cloud_off, prob_xy, max_cloud, phases, blur=0.05, scaling=0.5, out=None):
    """Measurement (bench.py): the large-cloud path in halves -- phases 1 = cost build into the workspace, 2 = sweeps over it + result,
    3 = both."""
    lib = load()
    _require_cuda(feat, cloud_rows, cloud_off, prob_xy)
    n_prob = prob_xy.numel() // 2
    if out is None:
        out = torch.empty(n_prob, dtype=torch.float32, device=feat.device)
    nb = lib.aadg_sinkhorn_workspace_bytes(n_prob, int(max_cloud), feat.shape[1])
    ws = workspace(nb, feat.device, "sinkhorn")
    _check(lib.aadg_sinkhorn_divergence_phases_f32(feat.data_ptr(), feat.stride(0), feat.shape[1], cloud_rows.data_ptr(), cloud_off.data_ptr(),
                                                   prob_xy.data_ptr(), n_prob, int(max_cloud), blur, scaling, out.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), int(phases), _stream()), "aadg_sinkhorn_divergence_phases_f32")
    return out


def normalize_rewards(rewards):
    lib = load()
    _require_cuda(rewards)
    out = torch.empty_like(rewards)
    rc = lib.aadg_normalize_rewards_f32(rewards.data_ptr(), rewards.numel(), out.data_ptr(), _stream())
    _check(rc, "aadg_normalize_rewards_f32")
    return out


# ------------------------------------------------------------------------------------------------
def seg_bce_dice(logits, labels, M, want_grad=False, grad_scale=1.0):
    """logits/labels f32 [N,K,H,W] -> (bce [M], dice [K], grad or None); grad = d(grad_scale * mean_j bce_j)/d logits."""
    lib = load()
    _require_cuda(logits, labels)
    if logits.dtype != torch.float32 or labels.dtype != torch.float32 or logits.shape != labels.shape or logits.dim() < 3:
        raise AadgError("logits and labels must be float32 tensors of the same [N,K,...] shape")
    N, K = logits.shape[:2]
    HW = logits[0, 0].numel()
    if N % M:
        raise AadgError("N must be a multiple of M")
    bce = torch.empty(M, dtype=torch.float32, device=logits.device)
    dice = torch.empty(K, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    nb = lib.aadg_seg_loss_workspace_bytes(N, K, HW)
    ws = _zeroed_workspace(nb, logits.device, "segloss")         # accumulators + arrival counter: zero on entry, left zeroed by the kernel
    rc = lib.aadg_seg_bce_dice_scaled_f32(logits.data_ptr(), labels.data_ptr(), N, K, HW, M, float(grad_scale), bce.data_ptr(),
                                          dice.data_ptr(), _ptr(grad), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_seg_bce_dice_scaled_f32")
    return bce, dice, grad


class _PolicyBCE(torch.autograd.Function):
    """loss = mean_j BCE(sigmoid(z)[j::M], y[j::M]); forward and backward share one fused pass."""

    @staticmethod
    def forward(ctx, logits, labels, M):
        bce, dice, grad = seg_bce_dice(logits.contiguous(), labels.contiguous(), M, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(dice)
        return bce.mean(), bce.detach(), dice

    @staticmethod
    def backward(ctx, g_loss, g_bce, g_dice):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None


def policy_bce_loss(logits, labels, M):
    """Drop-in for search_dg.py:140-142 (+ the Dice monitor of :164-165): returns (seg_loss, bce[M], dice[K])."""
    return _PolicyBCE.apply(logits, labels, M)


def policy_bce_backward(logits, labels, M, scale=1.0):
    """The segmentation loss of search_dg.py:140-142 AND its backward pass in one call: computes scale * mean_j BCE_j, the Dice monitor
    and d loss / d logits in ONE fused pass over logits / labels, then starts the backward pass of the graph behind `logits` from
    that gradient (`logits.backward(grad)`).  Equivalent to `(scale * policy_bce_loss(...)[0]).backward()` without autograd's
    `grad * d loss` product -- a read + write of the whole [N,K,H,W] gradient (0.6 GB at 144 x 2 x 512 x 512) that multiplied it by
    a scalar the kernel can apply itself.  Returns (scale * loss (detached), bce [M], dice [K])."""
    z = logits if logits.dtype == torch.float32 else logits.float()         # a differentiable cast under autocast
    z = z if z.is_contiguous() else z.contiguous()
    bce, dice, grad = seg_bce_dice(z.detach(), labels.contiguous(), M, want_grad=True, grad_scale=scale)
    if z.requires_grad:
        z.backward(grad)
    return bce.mean() * scale, bce, dice


# ------------------------------------------------------------------------------------------------
FOP = {name: i for i, name in enumerate([
    "invert", "solarize", "posterize", "gray", "contrast", "auto_contrast", "saturate", "brightness", "hue",
    "sample_pairing", "equalize", "sharpness", "gaussian_blur3x3", "shear_x", "shear_y", "translate_x",
    "translate_y", "rotate", "hflip", "vflip"])}


def fop(name, img, mag=None, kernel=None, perm=None):
    """One float tensor op of data/functional.py on a [B,3,H,W] float32 GPU tensor (output clamped to [0,1])."""
    lib = load()
    _require_cuda(img, mag, kernel, perm)
    if img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3:
        raise AadgError("img must be float32 [B,3,H,W]")
    B, C, H, W = img.shape
    out = torch.empty_like(img)
    mag_n = 0
    if mag is not None:
        mag = mag.to(torch.float32).reshape(-1).contiguous()
        mag_n = mag.numel()
    if kernel is not None:
        kernel = kernel.to(torch.float32).reshape(-1).contiguous()
        if kernel.numel() != 9:
            raise AadgError("kernel must be 3x3")
    if perm is not None:
        perm = perm.to(torch.int32).contiguous()
    nb = lib.aadg_fop_workspace_bytes(B, H, W)
    ws = workspace(nb, img.device, "fop")
    rc = lib.aadg_fop_f32(FOP[name], img.data_ptr(), out.data_ptr(), _ptr(mag), mag_n, _ptr(kernel), _ptr(perm),
                          B, C, H, W, ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_fop_f32(%s)" % name)
    return out


# ------------------------------------------------------------------------------------------------
class _UpsampleBilinearAC(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) with the HIP forward and (gathered) backward kernels."""

    @staticmethod
    def forward(ctx, x, size):
        lib = load()
        _require_cuda(x)
        if x.dtype not in (torch.float32, torch.bfloat16) or x.dim() != 4:
            raise AadgError("upsample: expected a float32/bfloat16 NCHW tensor")
        x = x.contiguous()
        N, C, h, w = x.shape
        H, W = int(size[0]), int(size[1])
        out = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device)
        rc = lib.aadg_upsample_bilinear2d(x.data_ptr(), out.data_ptr(), N * C, h, w, H, W,
                                          0 if x.dtype == torch.float32 else 1, _stream())
        _check(rc, "aadg_upsample_bilinear2d")
        ctx.in_shape = (N, C, h, w)
        ctx.out_size = (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        N, C, h, w = ctx.in_shape
        H, W = ctx.out_size
        # a channel slice of a wider tensor (the gradient of the decoder's concatenation) is read in place
        sliced = g.dim() == 4 and g.stride()[1:] == (H * W, W, 1) and g.stride(0) >= C * H * W
        if not sliced:
            g = g.contiguous()
        if lib.aadg_upsample_bilinear2d_backward_supported(h, w, H, W):
            gi = torch.empty(ctx.in_shape, dtype=g.dtype, device=g.device)
            ws = torch.empty(lib.aadg_upsample_bilinear2d_backward_workspace_bytes(h, w), dtype=torch.uint8, device=g.device)
            rc = lib.aadg_upsample_bilinear2d_backward_strided(g.data_ptr(), gi.data_ptr(), N, C, h, w, H, W, g.stride(0),
                                                               0 if g.dtype == torch.float32 else 1, ws.data_ptr(), ws.numel(),
                                                               _stream())
            _check(rc, "aadg_upsample_bilinear2d_backward_strided")
        else:       # very large factors: the LDS tile does not hold the contributing rectangle
            gi = torch.ops.aten.upsample_bilinear2d_backward(g.contiguous(), list(ctx.out_size), list(ctx.in_shape), True, None, None)
        return gi, None


def upsample_bilinear_ac(x, size):
    return _UpsampleBilinearAC.apply(x, tuple(size))


class _UpsampleSum(torch.autograd.Function):
    """full + sum_i F.interpolate(low_i, full.shape[-2:], mode='bilinear', align_corners=False) in one pass (csrc/upsample_sum.hip)."""

    @staticmethod
    def forward(ctx, full, *lows):
        lib = load()
        N, C, H, W = full.shape
        out = torch.empty_like(full)
        ptrs = (ctypes.c_void_p * len(lows))(*[t.data_ptr() for t in lows])
        hs = (ctypes.c_int * len(lows))(*[t.shape[2] for t in lows])
        ws_ = (ctypes.c_int * len(lows))(*[t.shape[3] for t in lows])
        _check(lib.aadg_upsample_sum(full.data_ptr(), ptrs, hs, ws_, len(lows), out.data_ptr(), N * C, H, W, _BN_DTYPES[full.dtype], _stream()),
               "aadg_upsample_sum")
        ctx.low_shapes = [tuple(t.shape) for t in lows]
        ctx.out_hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        g = g.contiguous()
        H, W = ctx.out_hw
        grads = [torch.empty(shp, dtype=g.dtype, device=g.device) for shp in ctx.low_shapes]
        n = len(grads)
        hs = (ctypes.c_int * max(n, 1))(*[shp[2] for shp in ctx.low_shapes])
        ws_ = (ctypes.c_int * max(n, 1))(*[shp[3] for shp in ctx.low_shapes])
        if n and lib.aadg_upsample_sum_backward_all_supported(H, W, hs, ws_, n):
            # all levels in one pass over g (the per-level kernel reads the whole gradient once per level)
            ptrs = (ctypes.c_void_p * n)(*[d.data_ptr() for d in grads])
            _check(lib.aadg_upsample_sum_backward_all(g.data_ptr(), ptrs, hs, ws_, n, g.shape[0] * g.shape[1], H, W, _BN_DTYPES[g.dtype],
                                                      _stream()), "aadg_upsample_sum_backward_all")
            return (g,) + tuple(grads)
        grads = []
        for shp in ctx.low_shapes:
            d = torch.empty(shp, dtype=g.dtype, device=g.device)
            _check(lib.aadg_upsample_sum_backward(g.data_ptr(), d.data_ptr(), shp[0] * shp[1], shp[2], shp[3], H, W, _BN_DTYPES[g.dtype],
                                                  _stream()), "aadg_upsample_sum_backward")
            grads.append(d)
        return (g,) + tuple(grads)


def upsample_sum(full, lows):
    """full [N,C,H,W] + the bilinear (align_corners=False) resizes of `lows` ([N,C,h_i,w_i], at most 3) to H x W; float32 / bfloat16."""
    tensors = [full] + list(lows)
    _require_cuda(*tensors)
    if full.dim() != 4 or full.dtype not in _BN_DTYPES or len(lows) > 3 or any(t.dtype != full.dtype or t.dim() != 4 or
                                                                              t.shape[:2] != full.shape[:2] for t in lows):
        raise AadgError("upsample_sum: expected NCHW float32/bfloat16 tensors of one dtype with equal N and C (at most 3 low maps)")
    return _UpsampleSum.apply(full, *lows)


# ------------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
_BN_DTYPES = {torch.float32: 0, torch.bfloat16: 1}
_bn_ws_cache = {}


def _bn_ws(C, device):
    """Per-(device, stream) scratch for the per-channel partials; reused across layers (stream-ordered)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    need = load().aadg_bn_workspace_bytes(C)
    ws = _bn_ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
        _bn_ws_cache[key] = ws
    return ws


class _BatchNormAct(torch.autograd.Function):
    """act(batch_norm(x) [+ residual]) with the HIP streaming kernels (csrc/batchnorm.hip); training mode.
    handles = k > 1 returns the output k times (k tensors on one storage), one per consumer -- e.g. the next bottleneck's
    first convolution and its residual branch, or the five ASPP branches: the consumers then deliver separate gradients,
    which the backward kernel sums while reading them instead of autograd running elementwise adds over the full activation."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, momentum, eps, act, handles, out=None, presums=None,
                res_scale=None, res_shift=None):
        lib = load()
        N, C, H, W = x.shape
        # res_scale / res_shift: `residual` is the first output of batch_norm_lazy(..., act=ACT_NONE) -- the RAW output of the projection
        # shortcut's convolution, normalised while this kernel reads it (float32, with presums)
        # presums: float64 [2C + 1] totals of x (sum, sum of squares per channel, element count) the PRODUCING convolution left behind
        # (aadg_conv1x1_nchw_f32x3_stats): the statistics pass over x is not run
        # out: a channel slice of a concatenation buffer (concat_slices): the result is written there, image stride = the buffer's
        y = torch.empty_like(x) if out is None else out
        y_stride = 0 if out is None else out.stride(0)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bn_ws(C, x.device)
        mask = None
        if residual is not None and act != ACT_NONE:
            nb = lib.aadg_bn_mask_bytes(N, C, H * W, _BN_DTYPES[x.dtype])
            if nb and (x.data_ptr() | residual.data_ptr() | y.data_ptr()) % 16 == 0:
                mask = torch.empty(nb, dtype=torch.uint8, device=x.device)
        if res_scale is not None:
            if presums is None or residual is None or out is not None or x.dtype != torch.float32:
                raise AadgError("batch_norm_act: res_affine needs float32 tensors, a residual, presums and no `out`")
            rc = lib.aadg_bn_forward_res_affine_f32(x.data_ptr(), residual.data_ptr(), res_scale.data_ptr(), res_shift.data_ptr(), y.data_ptr(),
                                                    _ptr(mask), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var), momentum, eps,
                                                    act, N, C, H * W, mean.data_ptr(), invstd.data_ptr(), presums.data_ptr(), ws.data_ptr(),
                                                    ws.numel(), _stream())
        elif presums is not None:
            rc = lib.aadg_bn_sync_forward(2, x.data_ptr(), _ptr(residual), y.data_ptr(), _ptr(mask), _ptr(weight), _ptr(bias),
                                          _ptr(running_mean), _ptr(running_var), momentum, eps, act, N, C, H * W, _BN_DTYPES[x.dtype],
                                          mean.data_ptr(), invstd.data_ptr(), presums.data_ptr(), ws.data_ptr(), ws.numel(), y_stride,
                                          _stream())
        else:
            rc = lib.aadg_bn_forward(x.data_ptr(), _ptr(residual), y.data_ptr(), _ptr(mask), _ptr(weight), _ptr(bias),
                                     _ptr(running_mean), _ptr(running_var), momentum, eps, act, 1, N, C, H * W, _BN_DTYPES[x.dtype],
                                     mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), y_stride, _stream())
        _check(rc, "aadg_bn_forward")
        if out is not None:
            ctx.mark_dirty(out)
        ctx.act = act
        ctx.has_res = residual is not None
        # the activation mask is re-derived from x (no residual), or taken from the bit mask the forward wrote (fused residual;
        # 1/16 of the output's bytes) or, where that is not available, from the stored output
        ctx.save_for_backward(x, y if (ctx.has_res and mask is None) else None, mask, weight, bias, mean, invstd)
        if handles > 1:
            return (y,) + tuple(y.view_as(y) for _ in range(handles - 1))
        return y

    @staticmethod
    def backward(ctx, *grads):
        lib = load()
        x, y, mask, weight, bias, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        dy, extra, pconst, dy_stride = _bn_prepare_grads(grads, x, ctx.has_res)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dw = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bn_ws(C, x.device)
        rc = lib.aadg_bn_backward(x.data_ptr(), _ptr(y), _ptr(mask), dy.data_ptr(), _ptr_array(extra) if extra else None, len(extra),
                                  _ptr(pconst), _ptr(weight), _ptr(bias), mean.data_ptr(), invstd.data_ptr(), ctx.act, dx.data_ptr(),
                                  _ptr(dres), dw.data_ptr(), db.data_ptr(), N, C, H * W, _BN_DTYPES[x.dtype], ws.data_ptr(),
                                  ws.numel(), dy_stride, _stream())
        _check(rc, "aadg_bn_backward")
        return (dx, dres, dw if weight is not None else None, db if bias is not None else None, None, None, None, None, None, None, None, None,
                None, None)


def _bn_prepare_grads(grads, x, has_res):
    """The gradients of a BatchNorm output's handles as the backward kernels take them: (dy, extra gradients to be summed while
    reading, per-plane constant gradient or None, image stride of dy when it is a channel slice of a wider tensor else 0)."""
    N, C, H, W = x.shape
    grads = [g for g in grads if g is not None]                   # unused handles deliver no gradient
    # a gradient that is a broadcast over each plane (the backward of a global average pool of this output) travels as one
    # float per plane instead of a materialised activation-sized tensor
    pconst = None
    if has_res and len(grads) > 1:
        flat = [g for g in grads if g.dim() == 4 and g.stride(2) == 0 and g.stride(3) == 0 and g.shape[2] * g.shape[3] > 1]
        if flat:
            grads = [g for g in grads if not any(g is f for f in flat)]
            pconst = flat[0][:, :, 0, 0].float()
            for f in flat[1:]:
                pconst = pconst + f[:, :, 0, 0].float()
            pconst = pconst.contiguous()
    # a single gradient that is a channel slice of a wider one (the backward of a concatenation) is read in place
    dy_stride = 0
    if len(grads) == 1 and not grads[0].is_contiguous() and tuple(grads[0].stride()[1:]) == (H * W, W, 1) and \
            grads[0].stride(0) >= C * H * W and grads[0].stride(0) % 8 == 0 and grads[0].data_ptr() % 16 == 0:
        dy_stride = grads[0].stride(0)
    else:
        grads = [g.contiguous() for g in grads]
    if not grads:
        grads = [torch.zeros_like(x)]
    if len(grads) > 1 and (not has_res or len(grads) > 1 + BN_MAX_EXTRA):
        # the fused sum rides on the materialised masked gradient of the residual case
        total = grads[0]
        for g in grads[1:]:
            total = total + g
        grads = [total]
    return grads[0], grads[1:], pconst, dy_stride


# ---- synchronised statistics (data-parallel ranks): the all-reduce sits between the statistics and the elementwise kernels ----
BN_SYNC_REDUCE = None      # callable(float64 device tensor) -> None: in-place SUM over the ranks; None = all-reduce on the small-collectives
                           # process group (aadg_amd/distributed.py: small_group -- never queued behind a DDP gradient bucket)


BN_SYNC_COLLECTIVES = [0]  # all-reduces issued by the BatchNorm layers since the counter was last cleared (bench.py: collectives_per_step)


def _bn_sync_reduce(t):
    BN_SYNC_COLLECTIVES[0] += 1
    if BN_SYNC_REDUCE is not None:
        return BN_SYNC_REDUCE(t)
    from . import distributed as adist
    adist.small_all_reduce(t, kind="batchnorm_statistics_all_reduce")


class _SyncBatchNormAct(torch.autograd.Function):
    """_BatchNormAct with the per-channel sums all-reduced over the ranks (aadg_bn_sync_forward / _backward): every rank
    normalises with the statistics of the GLOBAL batch, as the reference's single-GPU batch does.  Two small
    all-reduces per layer and step ([2C + 1] and [2C] float64)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, momentum, eps, act, handles, out=None, presums=None):
        lib = load()
        N, C, H, W = x.shape
        y = torch.empty_like(x) if out is None else out
        y_stride = 0 if out is None else out.stride(0)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        # presums: this rank's float64 sums from the producing convolution's epilogue: phase 1 (the local statistics pass) is not run
        sums = presums if presums is not None else torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
        ws = _bn_ws(C, x.device)
        mask = None
        if residual is not None and act != ACT_NONE:
            nb = lib.aadg_bn_mask_bytes(N, C, H * W, _BN_DTYPES[x.dtype])
            if nb and (x.data_ptr() | residual.data_ptr() | y.data_ptr()) % 16 == 0:
                mask = torch.empty(nb, dtype=torch.uint8, device=x.device)
        args = (x.data_ptr(), _ptr(residual), y.data_ptr(), _ptr(mask), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                momentum, eps, act, N, C, H * W, _BN_DTYPES[x.dtype], mean.data_ptr(), invstd.data_ptr(), sums.data_ptr(), ws.data_ptr(),
                ws.numel(), y_stride, _stream())
        if presums is None:
            _check(lib.aadg_bn_sync_forward(1, *args), "aadg_bn_sync_forward(1)")
        _bn_sync_reduce(sums)
        _check(lib.aadg_bn_sync_forward(2, *args), "aadg_bn_sync_forward(2)")
        if out is not None:
            ctx.mark_dirty(out)
        ctx.act = act
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y if (ctx.has_res and mask is None) else None, mask, weight, bias, mean, invstd, sums)
        if handles > 1:
            return (y,) + tuple(y.view_as(y) for _ in range(handles - 1))
        return y

    @staticmethod
    def backward(ctx, *grads):
        lib = load()
        x, y, mask, weight, bias, mean, invstd, fsums = ctx.saved_tensors
        N, C, H, W = x.shape
        dy, extra, pconst, dy_stride = _bn_prepare_grads(grads, x, ctx.has_res)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dw = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        ws = _bn_ws(C, x.device)
        extra_arr = _ptr_array(extra) if extra else None
        args = (x.data_ptr(), _ptr(y), _ptr(mask), dy.data_ptr(), extra_arr, len(extra), _ptr(pconst), _ptr(weight), _ptr(bias),
                mean.data_ptr(), invstd.data_ptr(), ctx.act, dx.data_ptr(), _ptr(dres), dw.data_ptr(), db.data_ptr(), N, C, H * W,
                _BN_DTYPES[x.dtype], sums.data_ptr(), fsums.data_ptr() + 16 * C, ws.data_ptr(), ws.numel(), dy_stride, _stream())
        _check(lib.aadg_bn_sync_backward(1, *args), "aadg_bn_sync_backward(1)")
        _bn_sync_reduce(sums)
        _check(lib.aadg_bn_sync_backward(2, *args), "aadg_bn_sync_backward(2)")
        return (dx, dres, dw if weight is not None else None, db if bias is not None else None, None, None, None, None, None, None, None, None)


class _SyncBatchNormActGroup(torch.autograd.Function):
    """k INDEPENDENT BatchNorm (+ activation) layers of one step -- the branches of the ASPP head -- with ONE all-reduce per direction
    instead of k: statistics kernels of all members, one all-reduce of the concatenated float64 sums, elementwise kernels of all
    members; the backward receives the k output gradients together (one autograd node) and does the same with [sum g, sum g x^].
    Members have no fused residual and one consumer.  apply(meta, x_0, weight_0, bias_0, rm_0, rv_0, out_0 | None, x_1, ...) with
    meta = [(momentum, eps, act), ...]."""

    @staticmethod
    def forward(ctx, meta, *t):
        lib = load()
        k = len(meta)
        mem = [t[6 * i:6 * i + 6] for i in range(k)]
        dev = mem[0][0].device
        Cs = [m[0].shape[1] for m in mem]
        offs = [0]
        for C in Cs:
            offs.append(offs[-1] + 2 * C + 1)
        # a member's totals from its producer's epilogue (meta[i][3], optional): its statistics pass is not run
        pres = [m[3] if len(m) > 3 else None for m in meta]
        if all(p is not None for p in pres):
            sums = torch.cat(pres)
        else:
            sums = torch.empty(offs[-1], dtype=torch.float64, device=dev)
            for i, p in enumerate(pres):
                if p is not None:
                    sums[offs[i]:offs[i + 1]].copy_(p)
        calls, ys, saved = [], [], []
        for i, ((x, weight, bias, rm, rv, out), (momentum, eps, act)) in enumerate(zip(mem, [m[:3] for m in meta])):
            N, C, H, W = x.shape
            y = torch.empty_like(x) if out is None else out
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
            ws = workspace(lib.aadg_bn_workspace_bytes(C), dev, "bn_group_%d" % i)       # one scratch per member: their kernels interleave
            calls.append((x.data_ptr(), None, y.data_ptr(), None, _ptr(weight), _ptr(bias), _ptr(rm), _ptr(rv), float(momentum), float(eps),
                          int(act), N, C, H * W, _BN_DTYPES[x.dtype], mean.data_ptr(), invstd.data_ptr(), sums.data_ptr() + 8 * offs[i],
                          ws.data_ptr(), ws.numel(), 0 if out is None else out.stride(0), _stream()))
            ys.append(y)
            saved += [x, weight, bias, mean, invstd]
        for a, p in zip(calls, pres):
            if p is None:
                _check(lib.aadg_bn_sync_forward(1, *a), "aadg_bn_sync_forward(1)")
        _bn_sync_reduce(sums)
        for a in calls:
            _check(lib.aadg_bn_sync_forward(2, *a), "aadg_bn_sync_forward(2)")
        dirty = [m[5] for m in mem if m[5] is not None]
        if dirty:
            ctx.mark_dirty(*dirty)
        ctx.meta, ctx.offs = [m[:3] for m in meta], offs
        ctx.save_for_backward(sums, *saved)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *grads):
        lib = load()
        fsums, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        k = len(ctx.meta)
        dev = fsums.device
        Cs = [saved[5 * i].shape[1] for i in range(k)]
        boffs = [0]
        for C in Cs:
            boffs.append(boffs[-1] + 2 * C)
        bsums = torch.empty(boffs[-1], dtype=torch.float64, device=dev)
        calls, outs, keep = [], [], []
        for i in range(k):
            x, weight, bias, mean, invstd = saved[5 * i:5 * i + 5]
            N, C, H, W = x.shape
            dy, extra, pconst, dy_stride = _bn_prepare_grads((grads[i],), x, False)
            keep.append(dy)
            dx = torch.empty_like(x)
            dw = torch.empty(C, dtype=torch.float32, device=dev)
            db = torch.empty(C, dtype=torch.float32, device=dev)
            ws = workspace(lib.aadg_bn_workspace_bytes(C), dev, "bn_group_%d" % i)
            calls.append((x.data_ptr(), None, None, dy.data_ptr(), None, 0, None, _ptr(weight), _ptr(bias), mean.data_ptr(), invstd.data_ptr(),
                          int(ctx.meta[i][2]), dx.data_ptr(), None, dw.data_ptr(), db.data_ptr(), N, C, H * W, _BN_DTYPES[x.dtype],
                          bsums.data_ptr() + 8 * boffs[i], fsums.data_ptr() + 8 * (ctx.offs[i] + 2 * C), ws.data_ptr(), ws.numel(), dy_stride,
                          _stream()))
            outs += [dx, dw if weight is not None else None, db if bias is not None else None, None, None, None]
        for a in calls:
            _check(lib.aadg_bn_sync_backward(1, *a), "aadg_bn_sync_backward(1)")
        _bn_sync_reduce(bsums)
        for a in calls:
            _check(lib.aadg_bn_sync_backward(2, *a), "aadg_bn_sync_backward(2)")
        return (None,) + tuple(outs)


class _SyncBatchNormShortcutPair(torch.autograd.Function):
    """relu(bn_main(a) + bn_short(b)): the tail of a bottleneck whose shortcut is a projection (1x1 convolution + BatchNorm).  The two
    layers read independent tensors in the forward, and in the backward both receive the SAME masked gradient (the main layer's phase 1
    writes it as `dres`), so their sums travel in ONE all-reduce per direction instead of two.  Same kernels as two _SyncBatchNormAct
    nodes; the shortcut's normalised output is the main layer's fused residual and is not kept for the backward."""

    @staticmethod
    def forward(ctx, a, b, wa, ba, rma, rva, wb, bb, rmb, rvb, mom_a, eps_a, mom_b, eps_b, act, handles, pre_a=None, pre_b=None):
        """pre_a / pre_b (both or neither): this rank's float64 [2C + 1] totals of a / b from their producers' epilogues -- phase 1 (a
        statistics pass over each tensor) is then not run"""
        lib = load()
        N, C, H, W = a.shape
        dev, dt = a.device, _BN_DTYPES[a.dtype]
        y, idt = torch.empty_like(a), torch.empty_like(a)
        stat = [torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4)]        # mean_a, invstd_a, mean_b, invstd_b
        have = pre_a is not None and pre_b is not None
        sums = torch.cat([pre_a, pre_b]) if have else torch.empty(2 * (2 * C + 1), dtype=torch.float64, device=dev)
        ws_a = workspace(lib.aadg_bn_workspace_bytes(C), dev, "bn_group_0")
        ws_b = workspace(lib.aadg_bn_workspace_bytes(C), dev, "bn_group_1")
        mask = None
        nb = lib.aadg_bn_mask_bytes(N, C, H * W, dt)
        if nb and (a.data_ptr() | idt.data_ptr() | y.data_ptr()) % 16 == 0:
            mask = torch.empty(nb, dtype=torch.uint8, device=dev)
        call_b = (b.data_ptr(), None, idt.data_ptr(), None, _ptr(wb), _ptr(bb), _ptr(rmb), _ptr(rvb), mom_b, eps_b, ACT_NONE, N, C, H * W, dt,
                  stat[2].data_ptr(), stat[3].data_ptr(), sums.data_ptr() + 8 * (2 * C + 1), ws_b.data_ptr(), ws_b.numel(), 0, _stream())
        call_a = (a.data_ptr(), idt.data_ptr(), y.data_ptr(), _ptr(mask), _ptr(wa), _ptr(ba), _ptr(rma), _ptr(rva), mom_a, eps_a, act, N, C, H * W, dt,
                  stat[0].data_ptr(), stat[1].data_ptr(), sums.data_ptr(), ws_a.data_ptr(), ws_a.numel(), 0, _stream())
        if not have:
            _check(lib.aadg_bn_sync_forward(1, *call_a), "aadg_bn_sync_forward(1)")
            _check(lib.aadg_bn_sync_forward(1, *call_b), "aadg_bn_sync_forward(1)")
        _bn_sync_reduce(sums)
        _check(lib.aadg_bn_sync_forward(2, *call_b), "aadg_bn_sync_forward(2)")          # the shortcut first: it is the main layer's residual
        _check(lib.aadg_bn_sync_forward(2, *call_a), "aadg_bn_sync_forward(2)")
        ctx.act = act
        ctx.save_for_backward(a, b, y if mask is None else None, mask, wa, ba, wb, bb, sums, *stat)
        if handles > 1:
            return (y,) + tuple(y.view_as(y) for _ in range(handles - 1))
        return y

    @staticmethod
    def backward(ctx, *grads):
        lib = load()
        a, b, y, mask, wa, ba, wb, bb, fsums, mean_a, invstd_a, mean_b, invstd_b = ctx.saved_tensors
        N, C, H, W = a.shape
        dev, dt = a.device, _BN_DTYPES[a.dtype]
        dy, extra, pconst, dy_stride = _bn_prepare_grads(grads, a, True)
        da, db_, g = torch.empty_like(a), torch.empty_like(b), torch.empty_like(a)
        par = [torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4)]         # dweight_a, dbias_a, dweight_b, dbias_b
        sums = torch.empty(4 * C, dtype=torch.float64, device=dev)
        ws_a = workspace(lib.aadg_bn_workspace_bytes(C), dev, "bn_group_0")
        ws_b = workspace(lib.aadg_bn_workspace_bytes(C), dev, "bn_group_1")
        extra_arr = _ptr_array(extra) if extra else None
        call_a = (a.data_ptr(), _ptr(y), _ptr(mask), dy.data_ptr(), extra_arr, len(extra), _ptr(pconst), _ptr(wa), _ptr(ba), mean_a.data_ptr(),
                  invstd_a.data_ptr(), ctx.act, da.data_ptr(), g.data_ptr(), par[0].data_ptr(), par[1].data_ptr(), N, C, H * W, dt,
                  sums.data_ptr(), fsums.data_ptr() + 16 * C, ws_a.data_ptr(), ws_a.numel(), dy_stride, _stream())
        call_b = (b.data_ptr(), None, None, g.data_ptr(), None, 0, None, _ptr(wb), _ptr(bb), mean_b.data_ptr(), invstd_b.data_ptr(), ACT_NONE,
                  db_.data_ptr(), None, par[2].data_ptr(), par[3].data_ptr(), N, C, H * W, dt, sums.data_ptr() + 16 * C,
                  fsums.data_ptr() + 8 * (2 * C + 1) + 16 * C, ws_b.data_ptr(), ws_b.numel(), 0, _stream())
        _check(lib.aadg_bn_sync_backward(1, *call_a), "aadg_bn_sync_backward(1)")         # writes g = the masked, summed gradient
        _check(lib.aadg_bn_sync_backward(1, *call_b), "aadg_bn_sync_backward(1)")
        _bn_sync_reduce(sums)
        _check(lib.aadg_bn_sync_backward(2, *call_a), "aadg_bn_sync_backward(2)")
        _check(lib.aadg_bn_sync_backward(2, *call_b), "aadg_bn_sync_backward(2)")
        return (da, db_, par[0] if wa is not None else None, par[1] if ba is not None else None, None, None,
                par[2] if wb is not None else None, par[3] if bb is not None else None, None, None, None, None, None, None, None, None, None, None)


def sync_batch_norm_shortcut_pair(a, main, b, short, act=ACT_RELU, handles=1):
    """act(bn_main(a) + bn_short(b)) in training mode with synchronised statistics and one all-reduce per direction for the two layers;
    main / short = (weight, bias, running_mean, running_var, momentum, eps).  handles as batch_norm_act."""
    _require_cuda(a, b)
    if a.shape != b.shape or a.dtype != b.dtype or not bn_act_supported(a, b):
        raise AadgError("sync_batch_norm_shortcut_pair: expected two contiguous NCHW float32/bfloat16 tensors of one shape")
    pa, pb = getattr(a, '_aadg_bn_sums', None), getattr(b, '_aadg_bn_sums', None)      # the producing convolutions' epilogue totals
    C = a.shape[1]
    if not all(p is not None and p.dtype == torch.float64 and p.numel() == 2 * C + 1 for p in (pa, pb)):
        pa = pb = None
    return _SyncBatchNormShortcutPair.apply(a, b, main[0], main[1], main[2], main[3], short[0], short[1], short[2], short[3],
                                            float(main[4]), float(main[5]), float(short[4]), float(short[5]), int(act), int(handles), pa, pb)


def sync_batch_norm_act_group(members):
    """members: [(x, weight, bias, running_mean, running_var, momentum, eps, act, out | None), ...] -- independent training-mode
    BatchNorm (+ activation) layers whose statistics travel in ONE all-reduce per direction.  Returns the outputs in order."""
    meta, flat = [], []
    for (x, weight, bias, rm, rv, momentum, eps, act, out) in members:
        _require_cuda(x)
        if not bn_act_supported(x):
            raise AadgError("sync_batch_norm_act_group: expected contiguous NCHW float32/bfloat16 tensors")
        if out is not None and (out.shape != x.shape or out.dtype != x.dtype or tuple(out.stride()[1:]) != tuple(x.stride()[1:]) or
                                out.data_ptr() % 16 or out.stride(0) % 8):
            raise AadgError("sync_batch_norm_act_group: `out` must be a channel slice of a contiguous NCHW buffer of the same dtype")
        pre = getattr(x, '_aadg_bn_sums', None)
        if pre is not None and not (pre.dtype == torch.float64 and pre.numel() == 2 * x.shape[1] + 1):
            pre = None
        meta.append((float(momentum), float(eps), int(act), pre))
        flat += [x, weight, bias, rm, rv, out]
    return _SyncBatchNormActGroup.apply(meta, *flat)


BN_MAX_EXTRA = 6


def concat_slices(N, channels, H, W, dtype, device):
    """A contiguous [N, sum(channels), H, W] buffer and one tensor per part aliasing its channel slice.  The parts are plain
    aliases of the buffer's storage (not autograd views of it): producers write into them (batch_norm_act(..., out=part)) and
    concat_from_slices(buffer, parts) is then the concatenation without a copy."""
    buf = torch.empty((N, sum(channels), H, W), dtype=dtype, device=device)
    parts, off = [], 0
    for c in channels:
        parts.append(torch.empty(0, dtype=dtype, device=device).set_(buf.untyped_storage(), buf.storage_offset() + off * H * W,
                                                                    (N, c, H, W), buf.stride()))
        off += c
    return buf, parts


class _ConcatFromSlices(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.sizes = [p.shape[1] for p in parts]
        return buf.detach().view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for c in ctx.sizes:
            outs.append(g[:, o:o + c])
            o += c
        return (None,) + tuple(outs)


def concat_from_slices(buf, parts):
    """torch.cat(parts, 1) where every part already lives in its slice of `buf` (concat_slices): no copy forward, channel-slice
    views of the gradient backward (the BatchNorm backward kernels read them in place)."""
    return _ConcatFromSlices.apply(buf, *parts)



def bn_act_supported(x, residual=None):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _BN_DTYPES and x.is_contiguous() and
            (residual is None or (residual.is_contiguous() and residual.dtype == x.dtype and residual.shape == x.shape)))


def batch_norm_act(x, weight, bias, running_mean, running_var, training, momentum, eps, act=ACT_NONE, residual=None, dual=False,
                   handles=None, out=None, sync=False, presums=None, res_affine=None):
    """act(F.batch_norm(x, ...) [+ residual]) on NCHW float32 / bfloat16 GPU tensors.  handles = k > 1 (training only; dual =
    True means k = 2) returns the output as a tuple of k tensors on one storage, one per consumer, see _BatchNormAct."""
    handles = int(handles) if handles else (2 if dual else 1)
    _require_cuda(x, residual)
    if not bn_act_supported(x, residual):
        raise AadgError("batch_norm_act: expected contiguous NCHW float32/bfloat16 tensors")
    if training:
        if out is not None and (handles != 1 or out.shape != x.shape or out.dtype != x.dtype or tuple(out.stride()[1:]) != tuple(x.stride()[1:]) or
                                out.data_ptr() % 16 or out.stride(0) % 8):
            raise AadgError("batch_norm_act: `out` must be a channel slice of a contiguous NCHW buffer of the same dtype")
        fn = _SyncBatchNormAct if sync else _BatchNormAct
        if presums is not None and (presums.dtype != torch.float64 or presums.numel() != 2 * x.shape[1] + 1 or not presums.is_cuda):
            raise AadgError("batch_norm_act: presums must be the float64 [2C + 1] totals of x")
        if res_affine is not None:
            # `residual` = the first output of batch_norm_lazy(..., act=ACT_NONE) of a projection shortcut, normalised on load
            if sync:
                raise AadgError("batch_norm_act: res_affine is a per-device path")
            return fn.apply(x, residual, weight, bias, running_mean, running_var, float(momentum), float(eps), int(act), handles, out, presums,
                            res_affine[0], res_affine[1])
        return fn.apply(x, residual, weight, bias, running_mean, running_var, float(momentum), float(eps), int(act), handles, out, presums)
    lib = load()
    N, C, H, W = x.shape
    if x.requires_grad or (residual is not None and residual.requires_grad):
        raise AadgError("batch_norm_act: inference mode is forward-only")
    y = torch.empty_like(x)
    ws = _bn_ws(C, x.device)
    rc = lib.aadg_bn_forward(x.data_ptr(), _ptr(residual), y.data_ptr(), None, _ptr(weight), _ptr(bias), running_mean.data_ptr(),
                             running_var.data_ptr(), 0.0, float(eps), int(act), 0, N, C, H * W, _BN_DTYPES[x.dtype],
                             None, None, ws.data_ptr(), ws.numel(), 0, _stream())
    _check(rc, "aadg_bn_forward")
    return y


# ------------------------------------------------------------------------------------------------
_POOL_BWD_FUSED_F32 = True      # (scripts/ab/pool_bwd_ab.sh switches it off for the comparison)


class _BNReluMaxPool(torch.autograd.Function):
    """max_pool2d(relu(batch_norm(x)), 3, 2, 1), training mode, in one pass over x (csrc/batchnorm.hip k_bn_relu_maxpool): the
    normalised map is never materialised.  Backward (bfloat16): two passes over x that rebuild the pooling gradient from
    (index, dy) on the fly; float32: the pooling gather followed by the ordinary BatchNorm backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        lib = load()
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, W // 2
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device)
        idx = torch.empty(N * C * Ho * Wo, dtype=torch.uint8, device=x.device)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bn_ws(C, x.device)
        _check(lib.aadg_bn_relu_maxpool_forward(x.data_ptr(), y.data_ptr(), idx.data_ptr(), _ptr(weight), _ptr(bias), _ptr(running_mean),
                                                _ptr(running_var), momentum, eps, N, C, H, W, _BN_DTYPES[x.dtype], mean.data_ptr(),
                                                invstd.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "aadg_bn_relu_maxpool_forward")
        ctx.save_for_backward(x, idx, weight, bias, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dyp):
        lib = load()
        x, idx, weight, bias, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        dyp = dyp.contiguous()
        if x.dtype == torch.bfloat16 or (_POOL_BWD_FUSED_F32 and W % 8 == 0 and dyp.data_ptr() % 8 == 0):
            # both BatchNorm backward passes rebuild the pooling gradient from (index, dyp): nothing activation-sized in between
            # (float32 since round 5: 4-column vectors)
            dx = torch.empty_like(x)
            dw = torch.empty(C, dtype=torch.float32, device=x.device)
            db = torch.empty(C, dtype=torch.float32, device=x.device)
            ws = _bn_ws(C, x.device)
            _check(lib.aadg_bn_relu_maxpool_backward(x.data_ptr(), idx.data_ptr(), dyp.data_ptr(), _ptr(weight), _ptr(bias), mean.data_ptr(),
                                                     invstd.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, C, H, W,
                                                     _BN_DTYPES[x.dtype], ws.data_ptr(), ws.numel(), _stream()), "aadg_bn_relu_maxpool_backward")
            return dx, dw if weight is not None else None, db if bias is not None else None, None, None, None, None
        dy = torch.empty_like(x)
        _check(lib.aadg_maxpool3x3s2_backward(idx.data_ptr(), dyp.data_ptr(), dy.data_ptr(), N * C, H, W, _BN_DTYPES[x.dtype], _stream()),
               "aadg_maxpool3x3s2_backward")
        dx = torch.empty_like(x)
        dw = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bn_ws(C, x.device)
        _check(lib.aadg_bn_backward(x.data_ptr(), None, None, dy.data_ptr(), None, 0, None, _ptr(weight), _ptr(bias), mean.data_ptr(),
                                    invstd.data_ptr(), ACT_RELU, dx.data_ptr(), None, dw.data_ptr(), db.data_ptr(), N, C, H * W,
                                    _BN_DTYPES[x.dtype], ws.data_ptr(), ws.numel(), 0, _stream()), "aadg_bn_backward")
        return dx, dw if weight is not None else None, db if bias is not None else None, None, None, None, None


def bn_relu_maxpool_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _BN_DTYPES and x.is_contiguous() and x.data_ptr() % 16 == 0 and
            (x.shape[2] * x.shape[3]) % (8 if x.dtype == torch.bfloat16 else 4) == 0 and
            bool(load().aadg_bn_relu_maxpool_supported(x.shape[2], x.shape[3], _BN_DTYPES[x.dtype])))


def bn_relu_maxpool(x, weight, bias, running_mean, running_var, momentum, eps):
    """Training-mode max_pool2d(relu(batch_norm(x)), 3, 2, 1) on a contiguous NCHW float32 / bfloat16 GPU tensor."""
    _require_cuda(x)
    if not bn_relu_maxpool_supported(x):
        raise AadgError("bn_relu_maxpool: unsupported shape / dtype / layout")
    return _BNReluMaxPool.apply(x, weight, bias, running_mean, running_var, float(momentum), float(eps))


# ------------------------------------------------------------------------------------------------
class _DepthwiseConv3x3(torch.autograd.Function):
    """F.conv2d(x, weight, stride=1, padding=d, dilation=d, groups=C) with the HIP kernels (csrc/depthwise.hip).
    `weight` is the float32 master copy [C,1,3,3]; activations float32 or bfloat16."""

    @staticmethod
    def forward(ctx, x, weight, dilation):
        lib = load()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        rc = lib.aadg_dwconv3x3(x.data_ptr(), weight.data_ptr(), y.data_ptr(), N, C, H, W, dilation, 0,
                                _BN_DTYPES[x.dtype], _stream())
        _check(rc, "aadg_dwconv3x3")
        ctx.dilation = dilation
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        x, weight = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            rc = lib.aadg_dwconv3x3(dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), N, C, H, W, ctx.dilation, 1,
                                    _BN_DTYPES[x.dtype], _stream())
            _check(rc, "aadg_dwconv3x3(flip)")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            ws = _bn_ws(C, x.device)
            need = lib.aadg_dwconv3x3_workspace_bytes(C)
            if ws.numel() < need:
                ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            rc = lib.aadg_dwconv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, C, H, W, ctx.dilation,
                                          _BN_DTYPES[x.dtype], ws.data_ptr(), ws.numel(), _stream())
            _check(rc, "aadg_dwconv3x3_wgrad")
        return dx, dw, None


def dwconv3x3_supported(x, weight, dilation):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _BN_DTYPES and x.is_contiguous() and weight.dtype == torch.float32 and
            weight.is_contiguous() and tuple(weight.shape) == (x.shape[1], 1, 3, 3) and
            bool(load().aadg_dwconv3x3_supported(x.shape[2], x.shape[3], int(dilation), _BN_DTYPES[x.dtype])))


def dwconv3x3(x, weight, dilation=1):
    _require_cuda(x, weight)
    if not dwconv3x3_supported(x, weight, dilation):
        raise AadgError("dwconv3x3: unsupported shape / dtype / layout")
    return _DepthwiseConv3x3.apply(x, weight, int(dilation))


# ------------------------------------------------------------------------------------------------
class _MaxPool3x3s2(torch.autograd.Function):
    """F.max_pool2d(x, 3, 2, 1) with the HIP kernels (csrc/maxpool.hip).  The forward stores the arg-max as one byte per
    output (its position in the 3x3 window); the backward gathers from that and dy alone, so x is not kept alive."""

    @staticmethod
    def forward(ctx, x):
        lib = load()
        N, C, H, W = x.shape
        y = torch.empty((N, C, (H - 1) // 2 + 1, W // 2), dtype=x.dtype, device=x.device)
        idx = None
        if x.requires_grad:
            idx = torch.empty(lib.aadg_maxpool3x3s2_index_bytes(N * C, H, W), dtype=torch.uint8, device=x.device)
        rc = lib.aadg_maxpool3x3s2_forward(x.data_ptr(), y.data_ptr(), _ptr(idx), N * C, H, W, _BN_DTYPES[x.dtype], _stream())
        _check(rc, "aadg_maxpool3x3s2_forward")
        ctx.save_for_backward(idx)
        ctx.in_shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        idx, = ctx.saved_tensors
        N, C, H, W = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        rc = lib.aadg_maxpool3x3s2_backward(idx.data_ptr(), dy.data_ptr(), dx.data_ptr(), N * C, H, W, _BN_DTYPES[dy.dtype], _stream())
        _check(rc, "aadg_maxpool3x3s2_backward")
        return dx


def maxpool3x3s2_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _BN_DTYPES and x.is_contiguous() and
            bool(load().aadg_maxpool3x3s2_supported(x.shape[2], x.shape[3])))


def maxpool3x3s2(x):
    _require_cuda(x)
    if not maxpool3x3s2_supported(x):
        raise AadgError("maxpool3x3s2: unsupported shape / dtype / layout")
    return _MaxPool3x3s2.apply(x)


# ------------------------------------------------------------------------------------------------
def conv1x1_nchw(a, x):
    """out [N, M, H, W] = a [M, K] (bfloat16) applied to the channels of x [N, K, H, W] (bfloat16): the matrix-core kernel of
    csrc/conv1x1_fwd.hip (LDS transpose reads: no layout change of the NCHW activations)."""
    _require_cuda(a, x)
    N, K, H, W = x.shape
    M = a.shape[0]
    if (a.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or a.dim() != 2 or a.shape[1] != K or
            not load().aadg_conv1x1_nchw_supported(M, K, H * W)):
        raise AadgError("conv1x1_nchw: unsupported shape / dtype / layout")
    out = torch.empty((N, M, H, W), dtype=x.dtype, device=x.device)
    _check(load().aadg_conv1x1_nchw_bf16(a.data_ptr(), x.data_ptr(), out.data_ptr(), N, M, K, H * W, _stream()), "aadg_conv1x1_nchw_bf16")
    return out


# ------------------------------------------------------------------------------------------------
class _StemConv7x7(torch.autograd.Function):
    """conv2d(bfloat16(x [N,3,H,W]), weight [64,3,7,7] float32 master, stride 2, padding 3) -> bfloat16 with the MFMA kernels of
    csrc/stem_conv.hip, forward and weight gradient.  x may still be float32 (the augmentation kernel's output): it is rounded
    while it is loaded.  An input gradient, if ever asked for, is the library's."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = load()
        N, C, H, W = x.shape
        y = torch.empty((N, 64, H // 2, W // 2), dtype=torch.bfloat16, device=x.device)
        ws = workspace(lib.aadg_stem_conv7x7_workspace_bytes(), x.device, "stem")
        _check(lib.aadg_stem_conv7x7_bf16(x.data_ptr(), _BN_DTYPES[x.dtype], weight.data_ptr(), y.data_ptr(), N, H, W, ws.data_ptr(),
                                          ws.numel(), _stream()), "aadg_stem_conv7x7_bf16")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if not ctx.needs_input_grad[0]:                      # the image itself needs no gradient: weight gradient on the matrix cores
            N, C, H, W = x.shape
            dw = torch.empty_like(weight)
            _check(load().aadg_stem_conv7x7_wgrad_bf16(x.data_ptr(), _BN_DTYPES[x.dtype], dy.data_ptr(), dw.data_ptr(), N, H, W, _stream()),
                   "aadg_stem_conv7x7_wgrad_bf16")
            return None, dw
        xb = x.to(torch.bfloat16)
        dx, dw, _ = torch.ops.aten.convolution_backward(dy, xb, weight.to(torch.bfloat16), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                        [True, True, False])
        return dx.to(x.dtype), dw.to(weight.dtype)


class _StemConv7x7X3(torch.autograd.Function):
    """conv2d(x [N,3,H,W] float32, weight [64,3,7,7] float32, stride 2, padding 3) -> float32 at float32 precision (the X3 instantiations
    of csrc/stem_conv.hip: forward and weight gradient).  The image needs no gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = load()
        N, C, H, W = x.shape
        y = torch.empty((N, 64, H // 2, W // 2), dtype=torch.float32, device=x.device)
        ws = workspace(lib.aadg_stem_conv7x7_workspace_bytes(), x.device, "stem")
        _check(lib.aadg_stem_conv7x7_f32x3(x.data_ptr(), weight.data_ptr(), y.data_ptr(), N, H, W, ws.data_ptr(), ws.numel(), _stream()),
               "aadg_stem_conv7x7_f32x3")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[1]:
            N, C, H, W = x.shape
            dw = torch.empty_like(weight)
            _check(load().aadg_stem_conv7x7_wgrad_f32x3(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, W, _stream()),
                   "aadg_stem_conv7x7_wgrad_f32x3")
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, weight, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        return dx, dw


def stem_conv7x7_x3(x, weight):
    _require_cuda(x, weight)
    if not (stem_conv7x7_supported(x, weight) and x.dtype == torch.float32):
        raise AadgError("stem_conv7x7_x3: unsupported shape / dtype / layout")
    return _StemConv7x7X3.apply(x, weight)


def stem_conv7x7_supported(x, weight):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _BN_DTYPES and x.is_contiguous() and x.shape[1] == 3 and
            tuple(weight.shape) == (64, 3, 7, 7) and weight.dtype == torch.float32 and weight.is_contiguous() and
            bool(load().aadg_stem_conv7x7_supported(x.shape[2], x.shape[3])))


def stem_conv7x7(x, weight):
    _require_cuda(x, weight)
    if not stem_conv7x7_supported(x, weight):
        raise AadgError("stem_conv7x7: unsupported shape / dtype / layout")
    return _StemConv7x7.apply(x, weight)


# ------------------------------------------------------------------------------------------------
class _Subsample2x2(torch.autograd.Function):
    """x[:, :, ::2, ::2] as a contiguous tensor (csrc/subsample.hip); the backward writes the whole input gradient in one pass."""

    @staticmethod
    def forward(ctx, x):
        lib = load()
        N, C, H, W = x.shape
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        _check(lib.aadg_subsample2x2(x.data_ptr(), y.data_ptr(), N * C, H, W, _BN_DTYPES[x.dtype], _stream()), "aadg_subsample2x2")
        ctx.in_shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        N, C, H, W = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        _check(lib.aadg_subsample2x2_backward(dy.data_ptr(), dx.data_ptr(), N * C, H, W, _BN_DTYPES[dy.dtype], _stream()),
               "aadg_subsample2x2_backward")
        return dx


def subsample2x2_supported(x):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _BN_DTYPES and x.is_contiguous() and
            bool(load().aadg_subsample2x2_supported(x.shape[2], x.shape[3], _BN_DTYPES[x.dtype])))


def subsample2x2(x):
    _require_cuda(x)
    if not subsample2x2_supported(x):
        raise AadgError("subsample2x2: unsupported shape / dtype / layout")
    return _Subsample2x2.apply(x)


# ------------------------------------------------------------------------------------------------
def conv1x1_wgrad(dy, x):
    """dW [Co, Ci] float32 of a 1x1 / stride-1 convolution from NCHW bfloat16 dy [N,Co,H,W] and x [N,Ci,H,W]."""
    lib = load()
    _require_cuda(dy, x)
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not (dy.is_contiguous() and x.is_contiguous()):
        raise AadgError("conv1x1_wgrad: expected contiguous NCHW bfloat16 tensors")
    N, Co, H, W = dy.shape
    Ci = x.shape[1]
    if x.shape[0] != N or x.shape[2:] != dy.shape[2:]:
        raise AadgError("conv1x1_wgrad: shape mismatch")
    dw = torch.empty((Co, Ci), dtype=torch.float32, device=x.device)
    rc = lib.aadg_conv1x1_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, Co, Ci, H * W, _stream())
    _check(rc, "aadg_conv1x1_wgrad_bf16")
    return dw


# images per call up to which the own 1x1 kernel also takes the mid-sized GEMMs (a per-rank batch of an 8- or 4-GPU run)
CONV1X1_SMALL_BATCH = 40


def _own_gemm_1x1(M, K, HW, N=None):
    """Shapes (out channels M, reduction K, N images) on which the matrix-core kernel of csrc/conv1x1_fwd.hip beats the library
    GEMM on an MI355X (a round-2 timing script at NB = 144 / 72 / 36 / 18).  At the full batch: the bandwidth-bound
    ones -- few output channels, or a short reduction -- plus the 1024 -> 256 / 304 -> 256 layers; the compute-bound late layers
    stay with hipBLASLt.  At a per-rank batch (N <= 40 images: one GEMM per image, hipBLASLt's 256 x 256 tiles leave the chip
    half empty) everything but the largest weights (M K >= 2^20: 512 <-> 2048, 1024 <-> 2048): 2.37 + 2.69 -> 2.0 + 2.0 ms of
    forward + input gradient per step at 18 images."""
    if not load().aadg_conv1x1_nchw_supported(M, K, HW):
        return False
    if N is not None and N <= CONV1X1_SMALL_BATCH and M * K < (1 << 20):
        return True
    return M <= 128 or (M <= 320 and (K <= 128 or K in (304, 1024)))


# ------------------------------------------------------------------------------------------------
# Weight gradients BESIDE the backward chain.  The input gradient of a convolution feeds the next BatchNorm backward of the chain; its
# weight gradient feeds nothing until the optimizer step.  With set_wgrad_stream(True) the convolution Functions below launch their
# weight-gradient kernels (matrix-core bound, 147 KB of LDS: one workgroup per CU) on a second HIP stream, where they overlap the
# chain's BatchNorm passes (HBM bound, no LDS) instead of standing in line with them, and hand the result over at the END of the
# backward pass: a callback queued on the autograd engine makes the launch stream wait for the side stream and puts the gradients
# into `.grad` (what AccumulateGrad would have done).  The gradients therefore bypass autograd's accumulation hooks: NOT for modules
# wrapped in torch's DistributedDataParallel (its reducer listens to those hooks) and not for torch.autograd.grad(); the package's own
# data-parallel wrapper (aadg_amd/reducer.py) takes them over on the side stream instead.  Off by default.
_WG = {"on": False, "stream": None, "pending": []}
_EXP = {}          # timing experiments (scripts/r6/*): never set by the product


def set_wgrad_stream(flag):
    """Weight-gradient kernels of the own convolutions on a side stream, gradients written to `.grad` at the end of the backward pass
    (not under torch's DistributedDataParallel; aadg_amd.reducer.GradReducer is built for it).  Returns the previous setting."""
    old = _WG["on"]
    _WG["on"] = bool(flag)
    return old


def wgrad_stream_enabled():
    return _WG["on"]


def _flush_wgrads():
    side = _WG["stream"]
    pending, _WG["pending"] = _WG["pending"], []
    if not pending:
        return
    task = torch._C._current_graph_task_id()
    main = torch.cuda.current_stream()
    main.wait_stream(side)
    for stamp, weight, dw in pending:
        if stamp != task:
            continue            # left behind by a backward pass that raised before its callbacks ran: not this pass's gradient
        dw.record_stream(main)
        if dw.shape != weight.shape or dw.stride() != weight.stride():
            dw = dw.reshape(weight.shape).contiguous()           # the layout AccumulateGrad would have given it
        if weight.grad is None:
            weight.grad = dw
        else:
            weight.grad.add_(dw)


def _wgrad_beside(weight, fn, *reads):
    """dw = fn() for the parameter `weight`, reading the tensors `reads` (produced on the current stream).  Side stream off (or `weight`
    is no leaf that accumulates into .grad): runs fn() here and returns dw.  On: launches fn() on the side stream and returns None --
    the gradient reaches weight.grad in _flush_wgrads() when the backward pass ends, or, for a parameter of a data-parallel replica
    (aadg_amd/reducer.py: `weight._aadg_grad_sink`), goes into its gradient bucket ON the side stream, from where the bucket's
    all-reduce is issued as soon as its last member is in."""
    if not (_WG["on"] and weight.is_leaf and weight.requires_grad and weight.is_cuda):
        return fn()
    if _WG["stream"] is None:
        _WG["stream"] = torch.cuda.Stream(device=weight.device)
    side = _WG["stream"]
    sink = getattr(weight, "_aadg_grad_sink", None)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dw = fn()
        if sink is not None:
            sink.deliver(weight, dw)
    for t in reads:
        t.record_stream(side)                 # the caching allocator must not hand these out again before the side stream has read them
    if sink is not None:
        return None
    _WG["pending"].append((torch._C._current_graph_task_id(), weight, dw))
    # one callback per deferred gradient (the first to run delivers everything pending, the others find nothing)
    torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrads)
    return None


class _Conv1x1(torch.autograd.Function):
    """1x1 / stride-1 convolution without bias on NCHW bfloat16 activations.  Forward and input gradient: the matrix-core
    kernel of csrc/conv1x1_fwd.hip where it is the faster one (_own_gemm_1x1), else the library GEMMs; weight gradient: the
    MFMA kernel of csrc/conv1x1_wgrad.hip.  `weight` is the float32 master copy."""

    @staticmethod
    def forward(ctx, x, weight):
        wq = cast_weight(weight, x.dtype)
        ctx.save_for_backward(x, wq)
        ctx.wparam = weight
        ctx.wt = _ShadowRef(weight, "bwd")               # [1, Ci, Co] of the tracked shadow (this step's weights)
        Co, Ci = wq.shape[0], wq.shape[1]
        if _own_gemm_1x1(Co, Ci, x.shape[2] * x.shape[3], x.shape[0]):
            return conv1x1_nchw(wq.view(Co, Ci), x)
        return torch.ops.aten.convolution(x, wq, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        # ONE check for both branches: the saved cast `wq` aliases the tracked shadow too, so the library branch must not run on a
        # buffer a later forward has rewritten either (raises AadgError; None = untracked weight)
        wt = ctx.wt.get()
        if ctx.needs_input_grad[0]:
            Co, Ci = wq.shape[0], wq.shape[1]
            if _own_gemm_1x1(Ci, Co, dy.shape[2] * dy.shape[3], dy.shape[0]):
                dx = conv1x1_nchw(wt[0] if wt is not None else wq.view(Co, Ci).t().contiguous(), dy)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, wq, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = _wgrad_beside(ctx.wparam, lambda: conv1x1_wgrad(dy, x).view(wq.shape), dy, x)
        return dx, dw


def conv1x1_supported(x, weight):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_contiguous() and weight.dtype == torch.float32 and
            bool(load().aadg_conv1x1_wgrad_supported(weight.shape[0], weight.shape[1], x.shape[2] * x.shape[3])))


def conv1x1(x, weight):
    _require_cuda(x, weight)
    if not conv1x1_supported(x, weight):
        raise AadgError("conv1x1: unsupported shape / dtype / layout")
    return _Conv1x1.apply(x, weight)


# ------------------------------------------------------------------------------------------------
# bfloat16 shadows of the float32 master weights of the own convolution modules, in the layouts their kernels read.  Every
# `weight.to(bfloat16)` is a 5 us launch (66 per step) and every tap-major / transposed copy (`permute(...).contiguous()`, one per
# convolution and direction) another 5-7 us (~70 per step) -- a cost that does not shrink with the per-rank batch.  A model registered
# with `track_bf16_weights` refreshes ALL of them in ONE launch (csrc/weight_layouts.hip) from a forward pre-hook whenever a master
# weight changed since the last forward; the autograd functions pick them up through `cast_weight` / `weight_layout` and fall back
# to the per-call copies for a weight that is not tracked (or whose shadow is stale).
class WlItem(ctypes.Structure):
    """mirror of `aadg_wl_item` (include/aadg_hip.h)"""
    _fields_ = [("w", ctypes.c_void_p), ("plain", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("bwd", ctypes.c_void_p),
                ("Co", ctypes.c_int32), ("Ci", ctypes.c_int32), ("taps", ctypes.c_int32), ("flip", ctypes.c_int32)]


# Ownership and validity (round 4: no table keyed by id(), no global optimizer hook, no version / epoch bookkeeping).  A shadow hangs on
# its parameter (`weight._aadg_shadow`) and belongs to the `_WeightLayouts` of ONE model.  The model's forward pre-hook rebuilds ALL
# shadows from the master weights -- unconditionally: 80 us per forward, and the only rule that is right for every way a weight can
# change (torch's fused optimizers do not even bump the version counter) -- and opens the scope in which they are trusted; the forward
# post-hook closes it.  Outside a tracked model's forward, `cast_weight` / `weight_layout` fall back to per-call copies.  What an
# autograd function keeps for its backward is a `_ShadowRef`: the buffers stay valid until the owner's NEXT refresh overwrites them
# (`generation`), whatever optimizer steps happen in between.
class _Shadow(object):
    __slots__ = ("owner", "plain", "fwd", "bwd", "ptr", "flip", "split")


def _shadow_of(weight):
    e = getattr(weight, "_aadg_shadow", None)
    if e is not None and e.owner.active and e.ptr == weight.data_ptr():
        return e
    return None


def cast_weight(weight, dtype):
    e = _shadow_of(weight) if dtype == torch.bfloat16 else None
    return e.plain if (e is not None and not e.split) else weight.to(dtype)


def weight_layout(weight, which):
    """The tracked bfloat16 copy of `weight` [Co, Ci, kh, kw] in layout 'fwd' ([taps, Co, Ci]) or 'bwd' ([taps, Ci, Co]; taps mirrored
    for a stride-1 3x3 convolution), or None when the weight is not tracked or the call is not inside its model's forward."""
    e = _shadow_of(weight)
    if e is None or e.split:
        return None
    return e.fwd if which == "fwd" else e.bwd


def split_weight(w):
    """float32 tensor -> [2, ...] bfloat16: hi = bf16(w), lo = bf16(w - hi) -- the per-call form of aadg_weight_layouts_split_bf16
    for a weight that is not tracked"""
    hi = w.to(torch.bfloat16)
    return torch.stack([hi, (w - hi.float()).to(torch.bfloat16)]).contiguous()


def split_layout(weight, which):
    """The tracked (hi, lo) bfloat16 halves of the float32 `weight` [Co, Ci, kh, kw] for the f32x3 kernels -- 'plain' [2, Co, Ci, kh, kw],
    'fwd' [2, taps, Co, Ci] or 'bwd' [2, taps, Ci, Co] (taps mirrored for a stride-1 3x3) -- or None when the weight is not tracked in
    split mode / the call is not inside its model's forward."""
    e = _shadow_of(weight)
    if e is None or not e.split:
        return None
    return {"plain": e.plain, "fwd": e.fwd, "bwd": e.bwd}[which]


class _ShadowRef(object):
    """What an autograd function keeps of a shadow layout between forward and backward: the buffer and the owner's refresh generation
    it was written in.  The buffers are overwritten in place by the owner's next refresh (= the model's next forward), and the saved
    bfloat16 cast aliases the shadow too, so a backward that runs after a LATER forward of the same model (deferred backward,
    activation checkpointing, two forwards before one backward) cannot be served: `get()` fails loudly (AadgError) instead of
    computing with the newer weights.  `get()` returns None only for an untracked weight (the caller builds the layout itself)."""
    __slots__ = ("entry", "tensor", "generation")

    def __init__(self, weight, which, split=False):
        e = _shadow_of(weight)
        self.entry, self.tensor, self.generation = None, None, -1
        if e is not None and e.split == split:
            self.entry, self.tensor, self.generation = e, (e.fwd if which == "fwd" else e.bwd), e.owner.generation

    def get(self):
        e = self.entry
        if e is None:
            return None                     # untracked weight: the caller builds the layout from its own saved cast
        if e.owner.generation != self.generation:
            raise AadgError("backward of a tracked convolution after a later forward of its model: the bfloat16 weight shadows were "
                            "rebuilt in place and hold that forward's weights.  Run each backward before the model's next forward, "
                            "or build the model without track_bf16_weights")
        return self.tensor


class _WeightLayouts(object):
    """All tracked weights of one model: shadows, the device item / tile tables of aadg_weight_layouts_bf16, one launch per refresh."""

    def __init__(self, entries, split=False):
        self.entries = entries              # [(weight, flip)]
        self.split = bool(split)            # (hi, lo) halves for the f32x3 kernels instead of one bfloat16 cast
        self.items = self.tiles = None
        self.n_tiles = 0
        self.generation = 0                 # refreshes so far: what a _ShadowRef compares
        self.active = False                 # inside the model's forward: the shadows hold the current master weights

    def _build(self):
        dev = self.entries[0][0].device
        items = (WlItem * len(self.entries))()
        tiles = []
        for i, (w, flip) in enumerate(self.entries):
            Co, Ci, taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            e = _Shadow()
            e.owner, e.ptr, e.flip, e.split = self, w.data_ptr(), flip, self.split
            lead = (2,) if self.split else ()          # split: the hi plane, then the lo plane
            e.plain = torch.empty(lead + tuple(w.shape), dtype=torch.bfloat16, device=dev)
            e.fwd = torch.empty(lead + (taps, Co, Ci), dtype=torch.bfloat16, device=dev) if taps > 1 else e.plain.view(lead + (1, Co, Ci))
            e.bwd = torch.empty(lead + (taps, Ci, Co), dtype=torch.bfloat16, device=dev)
            w._aadg_shadow = e
            items[i] = WlItem(w.data_ptr(), e.plain.data_ptr(), e.fwd.data_ptr() if taps > 1 else None, e.bwd.data_ptr(), Co, Ci, taps, flip)
            tiles += [(i, o0, c0) for o0 in range(0, Co, 32) for c0 in range(0, Ci, 256 if taps == 1 else 32)]
        raw = np.frombuffer(bytes(items), dtype=np.uint8).copy()
        self.items = torch.from_numpy(raw).to(dev)
        self.tiles = torch.tensor(tiles, dtype=torch.int32).to(dev)
        self.n_tiles = len(tiles)

    def refresh(self):
        """all shadows <- the master weights as they are now (one launch); opens the scope in which they are served"""
        stale = self.items is None
        if not stale:
            for w, _ in self.entries:
                e = getattr(w, "_aadg_shadow", None)
                if e is None or e.owner is not self or e.ptr != w.data_ptr() or e.plain.device != w.device:
                    stale = True                   # storage moved (.to(), load into new tensors) or taken over by another tracker
                    break
        if stale:
            self._build()
        fn = load().aadg_weight_layouts_split_bf16 if self.split else load().aadg_weight_layouts_bf16
        _check(fn(self.items.data_ptr(), self.tiles.data_ptr(), self.n_tiles, _stream()), "aadg_weight_layouts_bf16")
        self.generation += 1
        self.active = True

    def close(self):
        self.active = False


def track_bf16_weights(model, module_types, split=False):
    """Registers the float32 weights of `model`'s modules of the given types (1x1 / 3x3 convolutions: weight [Co, Ci, k, k], k*k <= 9)
    for the batched bfloat16 casts / re-layouts (CUDA models only).  A 3x3 module with stride 1 gets the mirrored-tap 'bwd' layout
    (its input gradient is the forward kernel on dY), any other the plain transposed one.  split = True: the layouts are the
    (hi, lo) bfloat16 halves the f32x3 kernels read (float32 activations, float32-grade products).
    Rule that comes with tracking: ONE forward per backward.  Every forward of the model rebuilds the shadows in place; what the autograd
    functions of a forward keep are references into them, so each backward has to run before the model's NEXT forward (a second forward
    in between -- gradient accumulation over two forwards, an eval / no_grad pass -- makes the earlier backward raise AadgError, on the own
    and on the library branches alike, rather than compute with the newer weights).  Models that need another order stay untracked."""
    if getattr(model, "_aadg_weight_layouts", None) is not None:
        raise AadgError("track_bf16_weights: this model's weights are tracked already")
    entries = []
    for m in model.modules():
        if isinstance(m, module_types) and m.weight.dtype == torch.float32 and m.weight.is_cuda and m.weight.dim() == 4 and \
                m.weight.shape[2] * m.weight.shape[3] <= 9:
            stride = m.stride[0] if isinstance(m.stride, (tuple, list)) else m.stride
            entries.append((m.weight, 1 if (m.weight.shape[2] == 3 and stride == 1) else 0))
    if entries:
        wl = _WeightLayouts(entries, split=split)
        model.register_forward_pre_hook(lambda mod, args: wl.refresh())
        model.register_forward_hook(lambda mod, args, out: wl.close(), always_call=True)
        model._aadg_weight_layouts = wl
    return len(entries)


def refresh_bf16_weights(model):
    """The pre-hook's work, callable directly by a caller that runs sub-modules of a tracked model on their own: rebuilds the shadows
    and leaves them trusted until `release_bf16_weights(model)` (or the model's next full forward)."""
    wl = getattr(model, "_aadg_weight_layouts", None)
    if wl is not None:
        wl.refresh()


def release_bf16_weights(model):
    wl = getattr(model, "_aadg_weight_layouts", None)
    if wl is not None:
        wl.close()


def conv3x3_wgrad(dy, x, dilation=1):
    """dW [Co, Ci, 3, 3] float32 of a 3x3 / stride-1 / padding = dilation convolution from NCHW bfloat16 dy [N,Co,H,W], x [N,Ci,H,W]."""
    lib = load()
    _require_cuda(dy, x)
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not (dy.is_contiguous() and x.is_contiguous()):
        raise AadgError("conv3x3_wgrad: expected contiguous NCHW bfloat16 tensors")
    N, Co, H, W = dy.shape
    Ci = x.shape[1]
    if x.shape[0] != N or x.shape[2:] != dy.shape[2:]:
        raise AadgError("conv3x3_wgrad: shape mismatch")
    dw9 = torch.empty((9, Co, Ci), dtype=torch.float32, device=x.device)
    rc = lib.aadg_conv3x3_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw9.data_ptr(), N, Co, Ci, H, W, int(dilation), _stream())
    _check(rc, "aadg_conv3x3_wgrad_bf16")
    return dw9.permute(1, 2, 0).reshape(Co, Ci, 3, 3)


def conv3x3s2_wgrad(dy, x):
    """dW [Co, Ci, 3, 3] float32 of a 3x3 / stride-2 / padding-1 convolution from NCHW bfloat16 dy [N,Co,Ho,Wo], x [N,Ci,2Ho,2Wo]."""
    lib = load()
    _require_cuda(dy, x)
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not (dy.is_contiguous() and x.is_contiguous()):
        raise AadgError("conv3x3s2_wgrad: expected contiguous NCHW bfloat16 tensors")
    N, Co, Ho, Wo = dy.shape
    Ci = x.shape[1]
    if x.shape[0] != N or x.shape[2] != 2 * Ho or x.shape[3] != 2 * Wo:
        raise AadgError("conv3x3s2_wgrad: shape mismatch")
    dw9 = torch.empty((9, Co, Ci), dtype=torch.float32, device=x.device)
    _check(lib.aadg_conv3x3s2_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw9.data_ptr(), N, Co, Ci, Ho, Wo, _stream()), "aadg_conv3x3s2_wgrad_bf16")
    return dw9.permute(1, 2, 0).reshape(Co, Ci, 3, 3)


def conv3x3s2_dgrad(a9t, dy):
    """dx [N, C, 2Ho, 2Wo] bfloat16 of a 3x3 / stride-2 / padding-1 convolution from dy [N, M, Ho, Wo] and the tap-major weights
    a9t [9, C, M] (a9t[kh*3+kw][c][m] = weight[m][c][kh][kw])."""
    lib = load()
    _require_cuda(a9t, dy)
    if a9t.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16 or not (a9t.is_contiguous() and dy.is_contiguous()) or a9t.dim() != 3:
        raise AadgError("conv3x3s2_dgrad: expected contiguous bfloat16 a9t [9,C,M] and NCHW dy")
    N, M, Ho, Wo = dy.shape
    C = a9t.shape[1]
    if a9t.shape[0] != 9 or a9t.shape[2] != M:
        raise AadgError("conv3x3s2_dgrad: shape mismatch")
    dx = torch.empty((N, C, 2 * Ho, 2 * Wo), dtype=torch.bfloat16, device=dy.device)
    _check(lib.aadg_conv3x3s2_dgrad_bf16(a9t.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, C, M, Ho, Wo, _stream()), "aadg_conv3x3s2_dgrad_bf16")
    return dx


def conv3x3s2_nchw(a9, x):
    """out [N, M, H/2, W/2] bfloat16 = 3x3 / stride-2 / padding-1 convolution of x [N, K, H, W] with tap-major weights a9 [9, M, K]."""
    lib = load()
    _require_cuda(a9, x)
    if a9.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not (a9.is_contiguous() and x.is_contiguous()) or a9.dim() != 3:
        raise AadgError("conv3x3s2_nchw: expected contiguous bfloat16 a9 [9,M,K] and NCHW x")
    N, K, H, W = x.shape
    M = a9.shape[1]
    if a9.shape[0] != 9 or a9.shape[2] != K or H % 2 or W % 2:
        raise AadgError("conv3x3s2_nchw: shape mismatch")
    out = torch.empty((N, M, H // 2, W // 2), dtype=torch.bfloat16, device=x.device)
    _check(lib.aadg_conv3x3s2_nchw_bf16(a9.data_ptr(), x.data_ptr(), out.data_ptr(), N, M, K, H // 2, W // 2, _stream()), "aadg_conv3x3s2_nchw_bf16")
    return out


class _Conv3x3S2(torch.autograd.Function):
    """3x3 / stride-2 / padding-1 convolution without bias on NCHW bfloat16 activations: forward k_conv3x3_s2 (csrc/conv3x3_s2_fwd.hip),
    input gradient k_dgrad3x3_s2 (csrc/conv3x3_s2_dgrad.hip), weight gradient k_wgrad3x3_s2 (csrc/conv3x3_wgrad.hip).  `weight` is
    the float32 master copy."""

    @staticmethod
    def forward(ctx, x, weight):
        wq = cast_weight(weight, x.dtype)
        ctx.save_for_backward(x, wq)
        ctx.wparam = weight
        ctx.a9t = _ShadowRef(weight, "bwd")              # tap-major transposed shadow (this step's weights)
        Co, Ci = wq.shape[0], wq.shape[1]
        if load().aadg_conv3x3s2_nchw_supported(Co, Ci, x.shape[2] // 2, x.shape[3] // 2):
            a9 = weight_layout(weight, "fwd")
            return conv3x3s2_nchw(a9 if a9 is not None else wq.permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous(), x)
        return torch.ops.aten.convolution(x, wq, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        a9t = ctx.a9t.get()                              # checked for both branches (see _Conv1x1.backward)
        if ctx.needs_input_grad[0]:
            M, C = wq.shape[0], wq.shape[1]
            if load().aadg_conv3x3s2_dgrad_supported(C, M, dy.shape[2], dy.shape[3]):
                dx = conv3x3s2_dgrad(a9t if a9t is not None else wq.permute(2, 3, 1, 0).reshape(9, C, M).contiguous(), dy)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, wq, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = _wgrad_beside(ctx.wparam, lambda: conv3x3s2_wgrad(dy, x), dy, x)
        return dx, dw


def conv3x3s2_supported(x, weight):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_contiguous() and weight.dtype == torch.float32 and
            tuple(weight.shape[2:]) == (3, 3) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and
            bool(load().aadg_conv3x3s2_wgrad_supported(weight.shape[0], weight.shape[1], x.shape[2] // 2, x.shape[3] // 2)))


def conv3x3s2(x, weight):
    _require_cuda(x, weight)
    if not conv3x3s2_supported(x, weight):
        raise AadgError("conv3x3s2: unsupported shape / dtype / layout")
    return _Conv3x3S2.apply(x, weight)


def conv3x3_nchw(a9, x, dilation=1):
    """out [N, M, H, W] bfloat16 = 3x3 convolution (stride 1, padding = dilation) of x [N, K, H, W] with tap-major weights a9 [9, M, K]."""
    lib = load()
    _require_cuda(a9, x)
    if a9.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or not (a9.is_contiguous() and x.is_contiguous()) or a9.dim() != 3:
        raise AadgError("conv3x3_nchw: expected contiguous bfloat16 a9 [9,M,K] and NCHW x")
    N, K, H, W = x.shape
    M = a9.shape[1]
    if a9.shape[0] != 9 or a9.shape[2] != K:
        raise AadgError("conv3x3_nchw: shape mismatch")
    out = torch.empty((N, M, H, W), dtype=torch.bfloat16, device=x.device)
    rc = lib.aadg_conv3x3_nchw_bf16(a9.data_ptr(), x.data_ptr(), out.data_ptr(), N, M, K, H, W, int(dilation), _stream())
    _check(rc, "aadg_conv3x3_nchw_bf16")
    return out


# images per call up to which the own forward / input-gradient kernel replaces the library's (None: always); the weight gradient
# kernel is used at every batch size
CONV3X3_OWN_FWD_MAX_N = None


def _own_conv3x3_fwd(x, M, K, dilation):
    if CONV3X3_OWN_FWD_MAX_N is not None and x.shape[0] > CONV3X3_OWN_FWD_MAX_N:
        return False
    return bool(load().aadg_conv3x3_nchw_supported(M, K, x.shape[2], x.shape[3], int(dilation)))


class _Conv3x3(torch.autograd.Function):
    """3x3 / stride-1 / padding = dilation convolution without bias on NCHW bfloat16 activations: forward and input gradient are
    the library's, the weight gradient is the MFMA kernel of csrc/conv3x3_wgrad.hip.  `weight` is the float32 master copy."""

    @staticmethod
    def forward(ctx, x, weight, dilation):
        wq = cast_weight(weight, x.dtype)
        ctx.save_for_backward(x, wq)
        ctx.wparam = weight
        ctx.a9t = _ShadowRef(weight, "bwd")              # tap-major transposed shadow (this step's weights)
        ctx.dilation = dilation
        Co, Ci = wq.shape[0], wq.shape[1]
        if _own_conv3x3_fwd(x, Co, Ci, dilation):
            a9 = weight_layout(weight, "fwd")
            return conv3x3_nchw(a9 if a9 is not None else wq.permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous(), x, dilation)
        return torch.ops.aten.convolution(x, wq, None, [1, 1], [dilation, dilation], [dilation, dilation], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        d = ctx.dilation
        dy = dy.contiguous()
        dx = dw = None
        a9t = ctx.a9t.get()                              # checked for both branches (see _Conv1x1.backward)
        if ctx.needs_input_grad[0]:
            Co, Ci = wq.shape[0], wq.shape[1]
            if _own_conv3x3_fwd(dy, Ci, Co, d):
                # the same kernel on dy with the taps mirrored and the channel roles swapped
                a9t = a9t if a9t is not None else wq.flip(2, 3).permute(2, 3, 1, 0).reshape(9, Ci, Co).contiguous()
                dx = conv3x3_nchw(a9t, dy, d)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, wq, None, [1, 1], [d, d], [d, d], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = _wgrad_beside(ctx.wparam, lambda: conv3x3_wgrad(dy, x, d), dy, x)
        return dx, dw, None


def conv3x3_supported(x, weight, dilation):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_contiguous() and weight.dtype == torch.float32 and
            tuple(weight.shape[2:]) == (3, 3) and
            bool(load().aadg_conv3x3_wgrad_supported(weight.shape[0], weight.shape[1], x.shape[2], x.shape[3], int(dilation))))


def conv3x3(x, weight, dilation=1):
    _require_cuda(x, weight)
    if not conv3x3_supported(x, weight, dilation):
        raise AadgError("conv3x3: unsupported shape / dtype / layout")
    return _Conv3x3.apply(x, weight, int(dilation))


# ------------------------------------------------------------------------------------------------
# "f32x3": the backbone convolutions on float32 tensors at float32 precision (the reference's: search_dg.py:123-206 runs the model in
# float32).  gfx950 has no tf32 and a float32 MFMA at 1/16 of the bfloat16 rate; every operand is split into bfloat16 halves
# x = hi + lo and every product formed as hi*hi + hi*lo + lo*hi on the bfloat16 matrix cores with float32 accumulation (csrc/common.h:
# aadg_split4; the X3 instantiations of the convolution kernels).  Activations are split inside the kernels while they are staged in
# LDS; the weights come pre-split from the tracked shadows (track_bf16_weights(..., split=True)) or, untracked, from split_weight().
def conv1x1_nchw_x3(a2, x, bn_sums=None, pre=None):
    """out [N, M, H, W] float32 = a [M, K] applied to the channels of x [N, K, H, W] float32; a2 [2, M, K] bfloat16 = (hi, lo) of a.
    bn_sums (float64 [2M + 1], optional) receives the BatchNorm statistics of out from the kernel's epilogue.
    pre = (scale, shift) float32 [K] (optional; shapes of conv1x1_x3_pre_supported): x is the INPUT of a BatchNorm + ReLU and the
    kernel applies max(x * scale[k] + shift[k], 0) while it loads it."""
    _require_cuda(a2, x)
    N, K, H, W = x.shape
    M = a2.shape[1]
    if (a2.dtype != torch.bfloat16 or x.dtype != torch.float32 or a2.dim() != 3 or a2.shape[0] != 2 or a2.shape[2] != K or
            not (a2.is_contiguous() and x.is_contiguous()) or not load().aadg_conv1x1_nchw_supported(M, K, H * W)):
        raise AadgError("conv1x1_nchw_x3: unsupported shape / dtype / layout")
    if pre is not None and not all(p.dtype == torch.float32 and p.numel() == K and p.is_cuda and p.is_contiguous() for p in pre):
        raise AadgError("conv1x1_nchw_x3: pre = (scale, shift), float32 [K]")
    out = torch.empty((N, M, H, W), dtype=torch.float32, device=x.device)
    _check(load().aadg_conv1x1_nchw_f32x3_pre(a2[0].data_ptr(), a2[1].data_ptr(), x.data_ptr(), out.data_ptr(), N, M, K, H * W,
                                              _ptr(pre[0]) if pre else None, _ptr(pre[1]) if pre else None, _ptr(bn_sums), _stream()),
           "aadg_conv1x1_nchw_f32x3")
    return out


def conv1x1_wgrad_x3(dy, x, pre=None):
    """dW [Co, Ci] float32 of a 1x1 / stride-1 convolution from NCHW float32 dy [N,Co,H,W] and x [N,Ci,H,W]; pre = (scale, shift): the
    convolution consumed max(x * scale[c] + shift[c], 0) (conv1x1_nchw_x3(..., pre=...))"""
    _require_cuda(dy, x)
    if dy.dtype != torch.float32 or x.dtype != torch.float32 or not (dy.is_contiguous() and x.is_contiguous()):
        raise AadgError("conv1x1_wgrad_x3: expected contiguous NCHW float32 tensors")
    N, Co, H, W = dy.shape
    Ci = x.shape[1]
    if x.shape[0] != N or x.shape[2:] != dy.shape[2:]:
        raise AadgError("conv1x1_wgrad_x3: shape mismatch")
    dw = torch.empty((Co, Ci), dtype=torch.float32, device=x.device)
    _check(load().aadg_conv1x1_wgrad_f32x3_pre(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, Co, Ci, H * W,
                                               _ptr(pre[0]) if pre else None, _ptr(pre[1]) if pre else None, _stream()), "aadg_conv1x1_wgrad_f32x3")
    return dw


class _Conv1x1X3(torch.autograd.Function):
    """1x1 / stride-1 convolution without bias on NCHW float32 activations at float32 precision: forward, input gradient
    (csrc/conv1x1_fwd.hip, X3) and weight gradient (csrc/conv1x1_wgrad.hip, X3).  `weight` is the float32 parameter."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False, pre_scale=None, pre_shift=None):
        """want_stats: also return the float64 [2 Co + 1] BatchNorm totals of the output (sum, sum of squares per channel, count), taken
        in the kernel's epilogue -- for the BatchNorm layer behind this convolution (batch_norm_act(..., presums=...)).
        pre_scale / pre_shift: x is the output of batch_norm_lazy -- the raw input of a BatchNorm + ReLU that this convolution (and its
        weight gradient) applies on load."""
        Co, Ci = weight.shape[0], weight.shape[1]
        a2 = split_layout(weight, "plain")
        a2 = a2.view(2, Co, Ci) if a2 is not None else split_weight(weight.detach().reshape(Co, Ci))
        pre = (pre_scale, pre_shift) if pre_scale is not None else None
        ctx.save_for_backward(x, weight, pre_scale, pre_shift)
        ctx.wparam = weight
        ctx.wt = _ShadowRef(weight, "bwd", split=True)          # [2, 1, Ci, Co] of the tracked shadow (this step's weights)
        if not want_stats:
            return conv1x1_nchw_x3(a2, x, None, pre)
        sums = torch.empty(2 * Co + 1, dtype=torch.float64, device=x.device)
        y = conv1x1_nchw_x3(a2, x, sums, pre)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *unused):
        x, weight, pre_scale, pre_shift = ctx.saved_tensors
        pre = (pre_scale, pre_shift) if pre_scale is not None else None
        dy = dy.contiguous()
        Co, Ci = weight.shape[0], weight.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = ctx.wt.get()
            at = wt.view(2, Ci, Co) if wt is not None else split_weight(weight.detach().reshape(Co, Ci).t().contiguous())
            dx = conv1x1_nchw_x3(at, dy)
        if ctx.needs_input_grad[1]:
            # pre_scale / pre_shift are read by the side-stream kernel too: without record_stream the allocator hands their
            # [K] blocks to the next BatchNorm backward's dw / db while the weight-gradient kernel may still be reading them
            dw = _wgrad_beside(ctx.wparam, lambda: conv1x1_wgrad_x3(dy, x, pre).view(weight.shape), dy, x, *(pre or ()))
        return dx, dw, None, None, None


def conv1x1_x3_supported(x, weight):
    HW = x.shape[2] * x.shape[3] if x.dim() == 4 else 0
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous() and weight.dtype == torch.float32 and
            weight.dim() == 4 and weight.shape[2] == 1 and weight.shape[3] == 1 and HW % 32 == 0 and weight.shape[0] % 8 == 0 and
            bool(load().aadg_conv1x1_nchw_supported(weight.shape[0], weight.shape[1], HW)))


def conv1x1_x3_pre_supported(x, weight):
    """the shapes on which the 1x1 kernels apply a BatchNorm + ReLU to x while they load it: forward and weight gradient both need whole
    tiles of x (the transformed operand); the number of output channels is free"""
    Co, Ci = weight.shape[0], weight.shape[1]
    HW = x.shape[2] * x.shape[3] if x.dim() == 4 else 0
    lib = load()
    return (conv1x1_x3_supported(x, weight) and bool(lib.aadg_conv1x1_f32x3_pre_supported(Co, Ci, HW)) and
            bool(lib.aadg_conv1x1_wgrad_f32x3_pre_supported(x.shape[0], Co, Ci, HW)))


def conv1x1_x3(x, weight, want_stats=False, pre=None):
    """want_stats: returns the output with its BatchNorm totals attached as `y._aadg_bn_sums` (float64 [2 Co + 1]); models/deeplab.py's
    bn_act hands them to the BatchNorm kernels, which then skip their statistics pass over y.
    pre = (scale, shift) from batch_norm_lazy, x its first output: the BatchNorm + ReLU in front of this convolution runs on operand load."""
    _require_cuda(x, weight)
    if not conv1x1_x3_supported(x, weight) or (pre is not None and not conv1x1_x3_pre_supported(x, weight)):
        raise AadgError("conv1x1_x3: unsupported shape / dtype / layout")
    ps, ph = pre if pre is not None else (None, None)
    if not want_stats:
        return _Conv1x1X3.apply(x, weight, False, ps, ph)
    y, sums = _Conv1x1X3.apply(x, weight, True, ps, ph)
    y._aadg_bn_sums = sums
    return y


class _BatchNormLazy(torch.autograd.Function):
    """Training-mode relu(batch_norm(x)) whose elementwise pass the CONSUMING convolution applies on load (conv1x1_x3(..., pre=...)):
    the forward only finalises the statistics (aadg_bn_finalize_f32, from the totals the producing convolution left) and hands x on
    UNCHANGED together with scale / shift; the backward is the ordinary two-pass BatchNorm backward with the ReLU mask re-derived from
    x.  The first output stands for relu(bn(x)) in the graph but HOLDS x: only a consumer that applies (scale, shift) may read it.
    sync (data-parallel ranks, round 6): the totals are all-reduced IN PLACE before they are finalised and the backward's (sum g,
    sum g x^) between its two passes (aadg_bn_sync_backward), as _SyncBatchNormAct does -- the on-load layers keep their fusion in a
    multi-GPU job."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, presums, act=ACT_RELU, sync=False):
        lib = load()
        ctx.act, ctx.sync = act, bool(sync)
        C = x.shape[1]
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        scale = torch.empty(C, dtype=torch.float32, device=x.device)
        shift = torch.empty(C, dtype=torch.float32, device=x.device)
        if sync:
            _bn_sync_reduce(presums)               # [2C + 1]: sums, sums of squares, element count -- the global batch's from here on
        _check(lib.aadg_bn_finalize_f32(presums.data_ptr(), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var), momentum, eps,
                                        C, mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), _stream()),
               "aadg_bn_finalize_f32")
        ctx.save_for_backward(x, weight, bias, mean, invstd, presums if sync else None)
        ctx.mark_non_differentiable(scale, shift)
        return x.view_as(x), scale, shift

    @staticmethod
    def backward(ctx, dz, *unused):
        lib = load()
        x, weight, bias, mean, invstd, fsums = ctx.saved_tensors
        N, C, H, W = x.shape
        dz = dz.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bn_ws(C, x.device)
        if _EXP.get("skip_lazy_dx"):             # timing experiment only (scripts/r6/exp_skip_dx.py): the reduction alone, dx := dz
            sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
            _check(lib.aadg_bn_sync_backward(1, x.data_ptr(), None, None, dz.data_ptr(), None, 0, None, _ptr(weight), _ptr(bias), mean.data_ptr(),
                                             invstd.data_ptr(), ctx.act, dx.data_ptr(), None, dw.data_ptr(), db.data_ptr(), N, C, H * W,
                                             _BN_DTYPES[x.dtype], sums.data_ptr(), None, ws.data_ptr(), ws.numel(), 0, _stream()), "exp")
            return dz, dw, db, None, None, None, None, None, None, None
        if ctx.sync:
            sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
            args = (x.data_ptr(), None, None, dz.data_ptr(), None, 0, None, _ptr(weight), _ptr(bias), mean.data_ptr(), invstd.data_ptr(),
                    ctx.act, dx.data_ptr(), None, dw.data_ptr(), db.data_ptr(), N, C, H * W, _BN_DTYPES[x.dtype], sums.data_ptr(),
                    fsums.data_ptr() + 16 * C, ws.data_ptr(), ws.numel(), 0, _stream())
            _check(lib.aadg_bn_sync_backward(1, *args), "aadg_bn_sync_backward(1)")
            _bn_sync_reduce(sums)
            _check(lib.aadg_bn_sync_backward(2, *args), "aadg_bn_sync_backward(2)")
        else:
            rc = lib.aadg_bn_backward(x.data_ptr(), None, None, dz.data_ptr(), None, 0, None, _ptr(weight), _ptr(bias), mean.data_ptr(),
                                      invstd.data_ptr(), ctx.act, dx.data_ptr(), None, dw.data_ptr(), db.data_ptr(), N, C, H * W,
                                      _BN_DTYPES[x.dtype], ws.data_ptr(), ws.numel(), 0, _stream())
            _check(rc, "aadg_bn_backward")
        return dx, dw if weight is not None else None, db if bias is not None else None, None, None, None, None, None, None, None


class _BatchNormActResBN(torch.autograd.Function):
    """act(batch_norm(x) + batch_norm2(x2)), training mode, float32, both layers' statistics from their producers' epilogues: a
    bottleneck's bn3 with its projection shortcut's BatchNorm (no activation) folded in.  Forward: the shortcut's statistics are
    finalised (aadg_bn_finalize_f32) and its normalisation happens while the main kernel reads the residual
    (aadg_bn_forward_res_affine_f32); backward: both layers in the two passes of one (aadg_bn_backward_res_bn_f32).  The shortcut's
    normalised tensor and its own forward / backward passes do not exist.
    sync (data-parallel ranks, round 6): both layers' totals travel in ONE all-reduce per direction -- the concatenated epilogue totals
    before they are finalised, and [4C] float64 between the two phases of aadg_bn_sync_backward_res_bn_f32."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, running_mean, running_var, momentum, eps, act, handles, presums,
                weight2, bias2, running_mean2, running_var2, momentum2, eps2, presums2, sync=False):
        lib = load()
        N, C, H, W = x.shape
        dev = x.device
        ctx.sync = bool(sync)
        if sync:
            both = torch.cat([presums, presums2])
            _bn_sync_reduce(both)
            presums, presums2 = both[:2 * C + 1], both[2 * C + 1:]
        f32 = lambda: torch.empty(C, dtype=torch.float32, device=dev)      # noqa: E731
        mean, invstd, mean2, invstd2, scale2, shift2 = f32(), f32(), f32(), f32(), f32(), f32()
        _check(lib.aadg_bn_finalize_f32(presums2.data_ptr(), _ptr(weight2), _ptr(bias2), _ptr(running_mean2), _ptr(running_var2), momentum2,
                                        eps2, C, mean2.data_ptr(), invstd2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), _stream()),
               "aadg_bn_finalize_f32")
        y = torch.empty_like(x)
        mask = torch.empty(lib.aadg_bn_mask_bytes(N, C, H * W, 0), dtype=torch.uint8, device=dev)
        ws = _bn_ws(C, dev)
        _check(lib.aadg_bn_forward_res_affine_f32(x.data_ptr(), x2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), y.data_ptr(),
                                                  mask.data_ptr(), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var), momentum,
                                                  eps, act, N, C, H * W, mean.data_ptr(), invstd.data_ptr(), presums.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _stream()), "aadg_bn_forward_res_affine_f32")
        ctx.act = act
        ctx.save_for_backward(x, x2, mask, weight, bias, mean, invstd, weight2, mean2, invstd2, presums if sync else None)
        if handles > 1:
            return (y,) + tuple(y.view_as(y) for _ in range(handles - 1))
        return y

    @staticmethod
    def backward(ctx, *grads):
        lib = load()
        x, x2, mask, weight, bias, mean, invstd, weight2, mean2, invstd2, fsums = ctx.saved_tensors
        N, C, H, W = x.shape
        dy, extra, pconst, dy_stride = _bn_prepare_grads(grads, x, True)
        dx, dres, dx2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x2)
        f32 = lambda: torch.empty(C, dtype=torch.float32, device=x.device)      # noqa: E731
        dw, db, dw2, db2 = f32(), f32(), f32(), f32()
        ws = _bn_ws(C, x.device)
        ws2 = torch.empty(ws.numel(), dtype=ws.dtype, device=x.device)
        if ctx.sync:
            sums = torch.empty(4 * C, dtype=torch.float64, device=x.device)
            extra_arr = _ptr_array(extra) if extra else None
            args = (x.data_ptr(), mask.data_ptr(), dy.data_ptr(), extra_arr, len(extra), _ptr(pconst), _ptr(weight), _ptr(bias), mean.data_ptr(),
                    invstd.data_ptr(), ctx.act, dx.data_ptr(), dres.data_ptr(), dw.data_ptr(), db.data_ptr(), x2.data_ptr(), _ptr(weight2),
                    mean2.data_ptr(), invstd2.data_ptr(), dx2.data_ptr(), dw2.data_ptr(), db2.data_ptr(), N, C, H * W, sums.data_ptr(),
                    fsums.data_ptr() + 16 * C, ws.data_ptr(), ws.numel() * ws.element_size(), ws2.data_ptr(), ws2.numel() * ws2.element_size(),
                    dy_stride, _stream())
            _check(lib.aadg_bn_sync_backward_res_bn_f32(1, *args), "aadg_bn_sync_backward_res_bn_f32(1)")
            _bn_sync_reduce(sums)
            _check(lib.aadg_bn_sync_backward_res_bn_f32(2, *args), "aadg_bn_sync_backward_res_bn_f32(2)")
            return (dx, dx2, dw if weight is not None else None, db if bias is not None else None, None, None, None, None, None, None, None,
                    dw2 if weight2 is not None else None, db2 if weight2 is not None else None, None, None, None, None, None, None)
        rc = lib.aadg_bn_backward_res_bn_f32(x.data_ptr(), mask.data_ptr(), dy.data_ptr(), _ptr_array(extra) if extra else None, len(extra),
                                             _ptr(pconst), _ptr(weight), _ptr(bias), mean.data_ptr(), invstd.data_ptr(), ctx.act,
                                             dx.data_ptr(), dres.data_ptr(), dw.data_ptr(), db.data_ptr(), x2.data_ptr(), _ptr(weight2),
                                             mean2.data_ptr(), invstd2.data_ptr(), dx2.data_ptr(), dw2.data_ptr(), db2.data_ptr(), N, C,
                                             H * W, ws.data_ptr(), ws.numel() * ws.element_size(), ws2.data_ptr(),
                                             ws2.numel() * ws2.element_size(), dy_stride, _stream())
        _check(rc, "aadg_bn_backward_res_bn_f32")
        return (dx, dx2, dw if weight is not None else None, db if bias is not None else None, None, None, None, None, None, None, None,
                dw2 if weight2 is not None else None, db2 if weight2 is not None else None, None, None, None, None, None, None)


def batch_norm_act_res_bn(x, bn, x2, bn2, act, handles=1, sync=False):
    """act(bn(x) + bn2(x2)) for a bottleneck's bn3 and its projection shortcut's BatchNorm (training, per-device statistics, float32;
    x / x2 carry their producers' statistics as `_aadg_bn_sums`): see _BatchNormActResBN.  bn / bn2: (weight, bias, running_mean,
    running_var, momentum, eps)."""
    _require_cuda(x, x2)
    s1, s2 = getattr(x, '_aadg_bn_sums', None), getattr(x2, '_aadg_bn_sums', None)
    xc, x2c = x.contiguous(), x2.contiguous()
    if (s1 is None or s2 is None or xc.dtype != torch.float32 or xc.shape != x2c.shape or act == ACT_NONE or not bn_act_supported(xc, x2c) or
            not load().aadg_bn_mask_bytes(xc.shape[0], xc.shape[1], xc.shape[2] * xc.shape[3], 0) or
            (xc.data_ptr() | x2c.data_ptr()) % 16 != 0):
        raise AadgError("batch_norm_act_res_bn: expected two float32 NCHW tensors of one shape with their producers' statistics")
    w, b, rm, rv, mom, eps = bn
    w2, b2, rm2, rv2, mom2, eps2 = bn2
    return _BatchNormActResBN.apply(xc, x2c, w, b, rm, rv, float(mom), float(eps), int(act), int(handles), s1,
                                    w2, b2, rm2, rv2, float(mom2), float(eps2), s2, bool(sync))


def batch_norm_lazy(x, weight, bias, running_mean, running_var, momentum, eps, presums, act=ACT_RELU, sync=False):
    """(x', scale, shift): see _BatchNormLazy.  x float32 NCHW contiguous on the GPU, presums its float64 [2C + 1] totals (sync: this
    rank's; all-reduced in place).  act: the
    activation the consumer applies after scale / shift (ACT_RELU: the convolutions' operand load; ACT_NONE: a projection shortcut read as
    the residual of batch_norm_act(..., res_affine=(scale, shift)))."""
    _require_cuda(x)
    if (x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous() or presums is None or presums.dtype != torch.float64 or
            presums.numel() != 2 * x.shape[1] + 1 or not bn_act_supported(x, None)):
        raise AadgError("batch_norm_lazy: expected a contiguous NCHW float32 tensor and its float64 [2C + 1] totals")
    if act not in (ACT_RELU, ACT_NONE):
        raise AadgError("batch_norm_lazy: act is ReLU or none")
    return _BatchNormLazy.apply(x, weight, bias, running_mean, running_var, float(momentum), float(eps), presums, int(act), bool(sync))


def conv3x3_nchw_x3(a9, x, dilation=1, bn_sums=None, pre=None):
    """out [N, M, H, W] float32 = 3x3 convolution (stride 1, padding = dilation) of x [N, K, H, W] float32; a9 [2, 9, M, K] bfloat16 =
    (hi, lo) of the tap-major weights.  bn_sums (float64 [2M + 1], optional; shapes of conv3x3_x3_stats_supported) receives the BatchNorm
    statistics of out from the kernel's epilogue.  pre = (scale, shift) float32 [K] (with bn_sums, K <= 512): x is the INPUT of a
    BatchNorm + ReLU that the kernel applies while it stages x (zero padding as for the normalised tensor)."""
    _require_cuda(a9, x)
    if a9.dtype != torch.bfloat16 or x.dtype != torch.float32 or not (a9.is_contiguous() and x.is_contiguous()) or a9.dim() != 4:
        raise AadgError("conv3x3_nchw_x3: expected contiguous bfloat16 a9 [2,9,M,K] and NCHW float32 x")
    N, K, H, W = x.shape
    M = a9.shape[2]
    if a9.shape[0] != 2 or a9.shape[1] != 9 or a9.shape[3] != K:
        raise AadgError("conv3x3_nchw_x3: shape mismatch")
    if pre is not None and not all(p.dtype == torch.float32 and p.numel() == K and p.is_cuda and p.is_contiguous() for p in pre):
        raise AadgError("conv3x3_nchw_x3: pre = (scale, shift), float32 [K]")
    out = torch.empty((N, M, H, W), dtype=torch.float32, device=x.device)
    _check(load().aadg_conv3x3_nchw_f32x3_pre(a9[0].data_ptr(), a9[1].data_ptr(), x.data_ptr(), out.data_ptr(), N, M, K, H, W, int(dilation),
                                              _ptr(pre[0]) if pre else None, _ptr(pre[1]) if pre else None, _ptr(bn_sums), _stream()),
           "aadg_conv3x3_nchw_f32x3")
    return out


def conv3x3_wgrad_x3(dy, x, dilation=1, pre=None):
    """dW [Co, Ci, 3, 3] float32 of a 3x3 / stride-1 / padding = dilation convolution from NCHW float32 dy, x; pre = (scale, shift): the
    convolution consumed max(x * scale[c] + shift[c], 0) (conv3x3_nchw_x3(..., pre=...))"""
    _require_cuda(dy, x)
    if dy.dtype != torch.float32 or x.dtype != torch.float32 or not (dy.is_contiguous() and x.is_contiguous()):
        raise AadgError("conv3x3_wgrad_x3: expected contiguous NCHW float32 tensors")
    N, Co, H, W = dy.shape
    Ci = x.shape[1]
    if x.shape[0] != N or x.shape[2:] != dy.shape[2:]:
        raise AadgError("conv3x3_wgrad_x3: shape mismatch")
    dw9 = torch.empty((9, Co, Ci), dtype=torch.float32, device=x.device)
    _check(load().aadg_conv3x3_wgrad_f32x3_pre(dy.data_ptr(), x.data_ptr(), dw9.data_ptr(), N, Co, Ci, H, W, int(dilation),
                                               _ptr(pre[0]) if pre else None, _ptr(pre[1]) if pre else None, _stream()),
           "aadg_conv3x3_wgrad_f32x3")
    return dw9.permute(1, 2, 0).reshape(Co, Ci, 3, 3)


class _Conv3x3X3(torch.autograd.Function):
    """3x3 / stride-1 / padding = dilation convolution without bias on NCHW float32 activations at float32 precision
    (csrc/conv3x3_fwd.hip and csrc/conv3x3_wgrad.hip, X3).  `weight` is the float32 parameter."""

    @staticmethod
    def forward(ctx, x, weight, dilation, want_stats=False, pre_scale=None, pre_shift=None):
        """want_stats: also return the float64 [2 Co + 1] BatchNorm totals of the output from the kernel's epilogue (as _Conv1x1X3);
        pre_scale / pre_shift (with want_stats): x is the first output of batch_norm_lazy, normalised + rectified on operand load"""
        Co, Ci = weight.shape[0], weight.shape[1]
        a9 = split_layout(weight, "fwd")
        if a9 is None:
            a9 = split_weight(weight.detach().permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous())
        pre = (pre_scale, pre_shift) if pre_scale is not None else None
        ctx.save_for_backward(x, weight, pre_scale, pre_shift)
        ctx.wparam = weight
        ctx.a9t = _ShadowRef(weight, "bwd", split=True)
        ctx.dilation = dilation
        if not want_stats:
            return conv3x3_nchw_x3(a9, x, dilation)
        sums = torch.empty(2 * Co + 1, dtype=torch.float64, device=x.device)
        y = conv3x3_nchw_x3(a9, x, dilation, sums, pre)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *unused):
        x, weight, pre_scale, pre_shift = ctx.saved_tensors
        pre = (pre_scale, pre_shift) if pre_scale is not None else None
        d = ctx.dilation
        dy = dy.contiguous()
        Co, Ci = weight.shape[0], weight.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            a9t = ctx.a9t.get()
            if a9t is None:
                a9t = split_weight(weight.detach().flip(2, 3).permute(2, 3, 1, 0).reshape(9, Ci, Co).contiguous())
            dx = conv3x3_nchw_x3(a9t, dy, d)
        if ctx.needs_input_grad[1]:
            dw = _wgrad_beside(ctx.wparam, lambda: conv3x3_wgrad_x3(dy, x, d, pre), dy, x, *(pre or ()))
        return dx, dw, None, None, None, None


def conv3x3_x3_supported(x, weight, dilation):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous() and weight.dtype == torch.float32 and
            tuple(weight.shape[2:]) == (3, 3) and weight.shape[0] % 8 == 0 and not (x.shape[3] == 128 and int(dilation) == 2) and
            bool(load().aadg_conv3x3_nchw_supported(weight.shape[0], weight.shape[1], x.shape[2], x.shape[3], int(dilation))) and
            bool(load().aadg_conv3x3_wgrad_supported(weight.shape[0], weight.shape[1], x.shape[2], x.shape[3], int(dilation))))


def conv3x3_x3_stats_supported(x, weight, dilation):
    """the shapes whose BatchNorm statistics the 3x3 kernel takes in its epilogue (the whole-tile kernel: Co % 64 == 0, Ci % 16 == 0)"""
    return bool(load().aadg_conv3x3_f32x3_stats_supported(weight.shape[0], weight.shape[1], x.shape[2], x.shape[3], int(dilation)))


def conv3x3_x3_pre_supported(x, weight, dilation):
    """the shapes on which the 3x3 kernels apply a BatchNorm + ReLU to x while they stage it (the whole-tile forward kernel, <= 512 input
    channels; the weight gradient takes any shape it supports)"""
    return conv3x3_x3_supported(x, weight, dilation) and conv3x3_x3_stats_supported(x, weight, dilation) and weight.shape[1] <= 512


def conv3x3_x3(x, weight, dilation=1, want_stats=False, pre=None):
    """want_stats: returns the output with its BatchNorm totals attached as `y._aadg_bn_sums` (see conv1x1_x3); pre = (scale, shift) from
    batch_norm_lazy, x its first output (needs want_stats and a shape of conv3x3_x3_pre_supported)"""
    _require_cuda(x, weight)
    if not conv3x3_x3_supported(x, weight, dilation) or (pre is not None and not (want_stats and conv3x3_x3_pre_supported(x, weight, dilation))):
        raise AadgError("conv3x3_x3: unsupported shape / dtype / layout")
    if not want_stats:
        return _Conv3x3X3.apply(x, weight, int(dilation), False, None, None)
    ps, ph = pre if pre is not None else (None, None)
    y, sums = _Conv3x3X3.apply(x, weight, int(dilation), True, ps, ph)
    y._aadg_bn_sums = sums
    return y


class _Conv3x3S2X3(torch.autograd.Function):
    """3x3 / stride-2 / padding-1 convolution without bias on NCHW float32 activations at float32 precision: forward
    (csrc/conv3x3_s2_fwd.hip), input gradient (csrc/conv3x3_s2_dgrad.hip: four parity classes) and weight gradient
    (csrc/conv3x3_wgrad.hip: k_wgrad3x3_s2), all X3.  `weight` is the float32 parameter."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = load()
        Co, Ci = weight.shape[0], weight.shape[1]
        N, _, H, W = x.shape
        a9 = split_layout(weight, "fwd")
        if a9 is None:
            a9 = split_weight(weight.detach().permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous())
        ctx.save_for_backward(x, weight)
        ctx.wparam = weight
        ctx.a9t = _ShadowRef(weight, "bwd", split=True)
        out = torch.empty((N, Co, H // 2, W // 2), dtype=torch.float32, device=x.device)
        _check(lib.aadg_conv3x3s2_nchw_f32x3(a9[0].data_ptr(), a9[1].data_ptr(), x.data_ptr(), out.data_ptr(), N, Co, Ci, H // 2, W // 2,
                                             _stream()), "aadg_conv3x3s2_nchw_f32x3")
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        Co, Ci = weight.shape[0], weight.shape[1]
        N, _, Ho, Wo = dy.shape
        dx = dw = None
        if ctx.needs_input_grad[0]:
            a9t = ctx.a9t.get()                              # stride 2: taps NOT mirrored ([9, Ci, Co], a9t[t][c][m] = w[m][c][t])
            if a9t is None:
                a9t = split_weight(weight.detach().permute(2, 3, 1, 0).reshape(9, Ci, Co).contiguous())
            dx = torch.empty_like(x)
            _check(lib.aadg_conv3x3s2_dgrad_f32x3(a9t[0].data_ptr(), a9t[1].data_ptr(), dy.data_ptr(), dx.data_ptr(), N, Ci, Co, Ho, Wo,
                                                  _stream()), "aadg_conv3x3s2_dgrad_f32x3")
        if ctx.needs_input_grad[1]:
            def wgrad():
                dw9 = torch.empty((9, Co, Ci), dtype=torch.float32, device=x.device)
                _check(lib.aadg_conv3x3s2_wgrad_f32x3(dy.data_ptr(), x.data_ptr(), dw9.data_ptr(), N, Co, Ci, Ho, Wo, _stream()),
                       "aadg_conv3x3s2_wgrad_f32x3")
                return dw9.permute(1, 2, 0).reshape(Co, Ci, 3, 3)
            dw = _wgrad_beside(ctx.wparam, wgrad, dy, x)
        return dx, dw


def conv3x3s2_x3_supported(x, weight):
    lib = load()
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous() and weight.dtype == torch.float32 and
            tuple(weight.shape[2:]) == (3, 3) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
        return False
    Co, Ci, Ho, Wo = weight.shape[0], weight.shape[1], x.shape[2] // 2, x.shape[3] // 2
    return bool(lib.aadg_conv3x3s2_nchw_supported(Co, Ci, Ho, Wo) and lib.aadg_conv3x3s2_dgrad_supported(Ci, Co, Ho, Wo) and
                lib.aadg_conv3x3s2_wgrad_supported(Co, Ci, Ho, Wo))


def conv3x3s2_x3(x, weight):
    _require_cuda(x, weight)
    if not conv3x3s2_x3_supported(x, weight):
        raise AadgError("conv3x3s2_x3: unsupported shape / dtype / layout")
    return _Conv3x3S2X3.apply(x, weight)


# ------------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (_c.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def controller_dims(controller, M):
    """(M, Q, S, E, H, n_ops, n_mags) of a Controller module."""
    return (int(M), int(controller.Q), 2 * int(controller.L), int(controller.embedding_dim), int(controller.hidden_dim),
            int(controller.NUM_OPS), int(controller.NUM_MAGS))


def controller_supported(controller, M):
    ps = list(controller.parameters())
    return (len(ps) == 9 and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps) and
            bool(load().aadg_controller_supported(*controller_dims(controller, M))))


def controller_workspace(controller, M):
    need = load().aadg_controller_workspace_bytes(*controller_dims(controller, M))
    return torch.zeros(need, dtype=torch.uint8, device=next(controller.parameters()).device)


def controller_sample(controller, M, uniforms, ws):
    """Fused controller.sample(M): returns (policies int64 [M, Q*2L], mean op probs, mean mag probs, log_probs, entropies)."""
    lib = load()
    dims = controller_dims(controller, M)
    dev = uniforms.device
    policies = torch.empty((M, dims[1] * dims[2]), dtype=torch.int64, device=dev)
    op_probs = torch.empty(dims[5], dtype=torch.float32, device=dev)
    mag_probs = torch.empty(dims[6], dtype=torch.float32, device=dev)
    log_probs = torch.empty(M, dtype=torch.float32, device=dev)
    entropies = torch.empty(M, dtype=torch.float32, device=dev)
    params = _ptr_array(list(controller.parameters()))
    rc = lib.aadg_controller_sample_f32(params, *dims, float(controller.C) / float(controller.T), uniforms.data_ptr(),
                                        policies.data_ptr(), op_probs.data_ptr(), mag_probs.data_ptr(), log_probs.data_ptr(),
                                        entropies.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_controller_sample_f32")
    return policies, op_probs, mag_probs, log_probs, entropies


def controller_ppo_update(controller, M, exp_avg, exp_avg_sq, policies, old_log_probs, reward, clip, n_updates, step0, lr,
                          betas, eps, ws):
    """n_updates PPO epochs (evaluate -> clipped surrogate -> backward -> Adam) in place; returns loss terms [n_updates, M]."""
    lib = load()
    dims = controller_dims(controller, M)
    losses = torch.empty((n_updates, M), dtype=torch.float32, device=policies.device)
    rc = lib.aadg_controller_ppo_update_f32(_ptr_array(list(controller.parameters())), _ptr_array(exp_avg), _ptr_array(exp_avg_sq),
                                            *dims, float(controller.C) / float(controller.T), policies.data_ptr(),
                                            old_log_probs.data_ptr(), reward.data_ptr(), float(clip), int(n_updates), int(step0),
                                            float(lr), float(betas[0]), float(betas[1]), float(eps), losses.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "aadg_controller_ppo_update_f32")
    return losses


# ------------------------------------------------------------------------------------------------
def embed_prologue(x, w1, b1, w2=None, b2=None, slope=0.2, want_norm=False):
    """(out [N, D] or None, fe [N, E]): fe = LeakyReLU(x W1^T + b1), out = fe W2^T + b2 -- the no-grad EMA branch of the
    domain discriminator, one launch.  want_norm: returns (out, fe, |fe[n]|_2 [N]) for sinkhorn_rewards(row_norm=...)."""
    lib = load()
    _require_cuda(w1, b1, w2, b2)
    if not x.is_cuda:
        raise AadgError("aadg_amd kernels need GPU tensors (got %s); there is no CPU path" % x.device)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise AadgError("embed_prologue: expected a float32 [N, C] matrix with unit column stride")
    w1, b1 = w1.detach().contiguous(), b1.detach().contiguous()
    N, C = x.shape
    E = w1.shape[0]
    fe = torch.empty((N, E), dtype=torch.float32, device=x.device)
    out = None
    D = 0
    if w2 is not None:
        w2, b2 = w2.detach().contiguous(), b2.detach().contiguous()
        D = w2.shape[0]
        out = torch.empty((N, D), dtype=torch.float32, device=x.device)
    if want_norm:
        nrm = torch.empty(N, dtype=torch.float32, device=x.device)
        rc = lib.aadg_embed_prologue_norm_f32(x.data_ptr(), x.stride(0), N, C, w1.data_ptr(), b1.data_ptr(), E, _ptr(w2), _ptr(b2), D,
                                              float(slope), fe.data_ptr(), _ptr(out), nrm.data_ptr(), _stream())
        _check(rc, "aadg_embed_prologue_norm_f32")
        return out, fe, nrm
    rc = lib.aadg_embed_prologue_f32(x.data_ptr(), x.stride(0), N, C, w1.data_ptr(), b1.data_ptr(), E, _ptr(w2), _ptr(b2), D,
                                     float(slope), fe.data_ptr(), _ptr(out), _stream())
    _check(rc, "aadg_embed_prologue_f32")
    return out, fe


# ------------------------------------------------------------------------------------------------
class _DwGeluNHWC(torch.autograd.Function):
    """GELU(depthwise3x3(h) + bias) on tokens h [B, H*W, C] (csrc/dwconv_nhwc.hip); weight [C,1,3,3], bias [C] are the float32 masters."""

    @staticmethod
    def forward(ctx, h, weight, bias, H, W):
        lib = load()
        B, N, C = h.shape
        w9 = weight.detach().float().reshape(C, 9).t().contiguous()
        b32 = bias.detach().float().contiguous()
        out = torch.empty_like(h)
        _check(lib.aadg_dwconv3x3_gelu_nhwc_forward(h.data_ptr(), w9.data_ptr(), b32.data_ptr(), out.data_ptr(), B, H, W, C,
                                                    _BN_DTYPES[h.dtype], _stream()), "aadg_dwconv3x3_gelu_nhwc_forward")
        ctx.save_for_backward(h, w9, b32)
        ctx.hw, ctx.wdtype, ctx.bdtype = (H, W), weight.dtype, bias.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = load()
        h, w9, b32 = ctx.saved_tensors
        B, N, C = h.shape
        H, W = ctx.hw
        dout = dout.contiguous()
        g, dh = torch.empty_like(h), torch.empty_like(h)
        dw9 = torch.empty((9, C), dtype=torch.float32, device=h.device)
        db = torch.empty(C, dtype=torch.float32, device=h.device)
        _check(lib.aadg_dwconv3x3_gelu_nhwc_backward(h.data_ptr(), w9.data_ptr(), b32.data_ptr(), dout.data_ptr(), g.data_ptr(), dh.data_ptr(),
                                                     dw9.data_ptr(), db.data_ptr(), B, H, W, C, _BN_DTYPES[h.dtype], _stream()),
               "aadg_dwconv3x3_gelu_nhwc_backward")
        return dh, dw9.t().reshape(C, 1, 3, 3).to(ctx.wdtype), db.to(ctx.bdtype), None, None


def dwconv3x3_gelu_nhwc_supported(h, H, W):
    return (h.is_cuda and h.dim() == 3 and h.dtype in _BN_DTYPES and h.is_contiguous() and h.shape[1] == H * W and h.data_ptr() % 16 == 0 and
            bool(load().aadg_dwconv3x3_gelu_nhwc_supported(h.shape[0], H, W, h.shape[2], _BN_DTYPES[h.dtype])))


def dwconv3x3_gelu_nhwc(h, weight, bias, H, W):
    """GELU(depthwise 3x3 (padding 1) of the tokens h [B, H*W, C] viewed as [B, H, W, C] + bias), token layout in and out."""
    _require_cuda(h, weight, bias)
    if not dwconv3x3_gelu_nhwc_supported(h, H, W) or tuple(weight.shape) != (h.shape[2], 1, 3, 3):
        raise AadgError("dwconv3x3_gelu_nhwc: expected contiguous float32 / bfloat16 tokens [B, H*W, C], C % 8 == 0, weight [C,1,3,3]")
    return _DwGeluNHWC.apply(h, weight, bias, int(H), int(W))


# ------------------------------------------------------------------------------------------------
class _AddLayerNorm(torch.autograd.Function):
    """(s, y) = (x + rscale * r, LayerNorm(s)) in one pass (csrc/layernorm.hip); r None: y only.  x, r: [..., C] float32 / bfloat16
    (contiguous), rscale: float32 [B] per-sample factor of r (stochastic depth) or None, gamma / beta: float32 [C]."""

    @staticmethod
    def forward(ctx, x, r, rscale, gamma, beta, eps):
        lib = load()
        C = x.shape[-1]
        R = x.numel() // C
        dt = _BN_DTYPES[x.dtype]
        rps = (R // rscale.numel()) if rscale is not None else 0
        y = torch.empty_like(x)
        s = torch.empty_like(x) if r is not None else None
        mean = torch.empty(R, dtype=torch.float32, device=x.device)
        rstd = torch.empty(R, dtype=torch.float32, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _check(lib.aadg_layernorm_forward(x.data_ptr(), _ptr(r), _ptr(rscale), rps, g32.data_ptr(), b32.data_ptr(), float(eps), _ptr(s),
                                          y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, C, dt, _stream()), "aadg_layernorm_forward")
        ctx.save_for_backward(s if s is not None else x, g32, mean, rstd, rscale)
        ctx.has_r, ctx.rps, ctx.gdtype = r is not None, rps, gamma.dtype
        if s is not None:
            return s, y
        return y

    @staticmethod
    def backward(ctx, *grads):
        lib = load()
        sin, g32, mean, rstd, rscale = ctx.saved_tensors
        C = sin.shape[-1]
        R = sin.numel() // C
        dt = _BN_DTYPES[sin.dtype]
        if ctx.has_r:
            ds_extra, dy = grads
        else:
            ds_extra, dy = None, grads[0]
        if dy is None:
            dy = torch.zeros_like(sin)
        dy = dy.contiguous()
        ds_extra = ds_extra.contiguous() if ds_extra is not None else None
        dx = torch.empty_like(sin)
        dr = torch.empty_like(sin) if ctx.has_r else None
        dg = torch.empty(C, dtype=torch.float32, device=sin.device)
        db = torch.empty(C, dtype=torch.float32, device=sin.device)
        ws = workspace(lib.aadg_layernorm_workspace_bytes(R, C), sin.device, "layernorm")
        _check(lib.aadg_layernorm_backward(sin.data_ptr(), dy.data_ptr(), _ptr(ds_extra), g32.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           _ptr(rscale), ctx.rps, dx.data_ptr(), _ptr(dr), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                           ws.numel(), R, C, dt, _stream()), "aadg_layernorm_backward")
        return dx, dr, None, dg.to(ctx.gdtype), db.to(ctx.gdtype), None


def layernorm_supported(x, C):
    return (x.is_cuda and x.dtype in _BN_DTYPES and x.is_contiguous() and x.shape[-1] == C and x.data_ptr() % 16 == 0 and
            bool(load().aadg_layernorm_supported(x.numel() // C, C, _BN_DTYPES[x.dtype])))


def add_layer_norm(x, r, rscale, gamma, beta, eps):
    """r is None: LayerNorm(x).  Else (s, y) with s = x + rscale[sample] * r (rscale None: 1) and y = LayerNorm(s)."""
    if r is not None and (r.shape != x.shape or r.dtype != x.dtype):
        raise AadgError("add_layer_norm: x and r must have the same shape and dtype")
    if r is not None and not r.is_contiguous():
        r = r.contiguous()                       # the kernel reads r with row stride C (a transposed / sliced branch output is copied once)
    _require_cuda(x, r, rscale)
    if not layernorm_supported(x, x.shape[-1]):
        raise AadgError("add_layer_norm: expected contiguous float32 / bfloat16 [..., C] with C % 8 == 0, C <= 512")
    if rscale is not None:
        # one float32 factor per sample (stochastic depth): R rows split evenly over rscale.numel() samples, sample-major
        R = x.numel() // x.shape[-1]
        if rscale.dtype != torch.float32 or not rscale.is_contiguous() or rscale.numel() == 0 or R % rscale.numel() != 0 or \
                (x.dim() >= 2 and rscale.numel() != x.shape[0]):
            raise AadgError("add_layer_norm: rscale must be a contiguous float32 vector with one factor per sample (x.shape[0] = %d), got %s %s"
                            % (x.shape[0], rscale.dtype, tuple(rscale.shape)))
    return _AddLayerNorm.apply(x, r, rscale, gamma, beta, float(eps))


# ------------------------------------------------------------------------------------------------
class _UpsampleCat(torch.autograd.Function):
    """torch.cat([upsample_bilinear_ac(a, size), b], dim=1): the up-sampling writes straight into the concatenation buffer
    (no separate copy of its output) and its backward reads the corresponding channel slice of the gradient in place."""

    @staticmethod
    def forward(ctx, a, b, buf):
        lib = load()
        N, Ca, h, w = a.shape
        _, Cb, H, W = b.shape
        # buf: the concatenation buffer whose channels [Ca:] ARE b already (b = its alias part, concat_slices): nothing to copy
        out = torch.empty((N, Ca + Cb, H, W), dtype=a.dtype, device=a.device) if buf is None else buf
        rc = lib.aadg_upsample_bilinear2d_strided(a.data_ptr(), out.data_ptr(), N, Ca, h, w, H, W, (Ca + Cb) * H * W,
                                                  0 if a.dtype == torch.float32 else 1, _stream())
        _check(rc, "aadg_upsample_bilinear2d_strided")
        if buf is None:
            out[:, Ca:].copy_(b)
        else:
            out = buf.detach().view(buf.shape)
        ctx.shape_a = (N, Ca, h, w)
        ctx.Cb = Cb
        return out

    @staticmethod
    def backward(ctx, g):
        lib = load()
        N, Ca, h, w = ctx.shape_a
        H, W = g.shape[2], g.shape[3]
        g = g.contiguous()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            if lib.aadg_upsample_bilinear2d_backward_supported(h, w, H, W):
                ga = torch.empty(ctx.shape_a, dtype=g.dtype, device=g.device)
                ws = torch.empty(lib.aadg_upsample_bilinear2d_backward_workspace_bytes(h, w), dtype=torch.uint8, device=g.device)
                rc = lib.aadg_upsample_bilinear2d_backward_strided(g.data_ptr(), ga.data_ptr(), N, Ca, h, w, H, W, g.stride(0),
                                                                   0 if g.dtype == torch.float32 else 1, ws.data_ptr(), ws.numel(),
                                                                   _stream())
                _check(rc, "aadg_upsample_bilinear2d_backward_strided")
            else:
                ga = torch.ops.aten.upsample_bilinear2d_backward(g[:, :Ca].contiguous(), [H, W], list(ctx.shape_a), True, None, None)
        if ctx.needs_input_grad[1]:
            gb = g[:, Ca:]
        return ga, gb, None


def upsample_cat(a, b, buf=None):
    """cat([bilinear up-sampling of a (align_corners=True) to b's spatial size, b], dim=1) on NCHW float32 / bfloat16 tensors.
    buf: the [N, Ca + Cb, H, W] buffer of concat_slices whose second part b already is (written there by its producer)."""
    _require_cuda(a)
    if a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16) or a.dim() != 4 or b.dim() != 4 or a.shape[0] != b.shape[0]:
        raise AadgError("upsample_cat: expected two NCHW float32/bfloat16 tensors with one batch size")
    if buf is not None:
        Ca = a.shape[1]
        if (tuple(buf.shape) != (a.shape[0], Ca + b.shape[1], b.shape[2], b.shape[3]) or not buf.is_contiguous() or
                b.data_ptr() != buf.data_ptr() + Ca * b.shape[2] * b.shape[3] * buf.element_size() or b.stride() != buf.stride()):
            raise AadgError("upsample_cat: b is not the second part of buf")
        return _UpsampleCat.apply(a.contiguous(), b, buf)
    _require_cuda(b)
    return _UpsampleCat.apply(a.contiguous(), b.contiguous(), None)
