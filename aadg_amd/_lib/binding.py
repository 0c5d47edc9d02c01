"""ctypes binding of libaadg_hip.so (include/aadg_hip.h): the loader, the argtypes table of every exported symbol, error mapping, stream / pointer /
workspace helpers.  No CPU fallback: everything raises when the shared library or a GPU is missing."""
import ctypes
import os

import numpy as np
import torch


_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # aadg_amd/
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
    "aadg_bn_sync_backward_res_bn_f32", "aadg_draw_python_stream",
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
    lib.aadg_draw_python_stream.restype = _i
    lib.aadg_draw_python_stream.argtypes = [_vp, _i, _i, _i, _vp, _vp, _c.c_double, _c.c_double, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]
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
    if lib.aadg_abi_version() != 12:
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
_POOL_BWD_FUSED_F32 = True      # (scripts/ab/pool_bwd_ab.sh switches it off for the comparison)


# images per call up to which the own 1x1 kernel also takes the mid-sized GEMMs (a per-rank batch of an 8- or 4-GPU run)
CONV1X1_SMALL_BATCH = 40
_EXP = {}          # timing experiments (scripts/r6/*): never set by the product


# images per call up to which the own forward / input-gradient kernel replaces the library's (None: always); the weight gradient
# kernel is used at every batch size
CONV3X3_OWN_FWD_MAX_N = None


# ------------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (_c.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
