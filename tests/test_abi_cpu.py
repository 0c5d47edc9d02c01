"""CPU suite, part 4: the C-ABI library loads without a GPU and exports every symbol the header declares;
the product path refuses to run without a GPU (no silent CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from aadg_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "aadg_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(aadg_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), "libaadg_hip.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared
    assert lib.aadg_abi_version() == 12


def test_unit_struct_layout_matches_header():
    from aadg_amd import _lib
    from oracle import oracle as O
    assert _lib.UNIT_DTYPE == O.UNIT_DTYPE and _lib.UNIT_DTYPE.itemsize == 140
    assert _lib.UNIT_DTYPE.fields["rect"][1] == 56 and _lib.UNIT_DTYPE.fields["scaled_w"][1] == 120


def test_workspace_queries_are_pure_host_calls():
    from aadg_amd import _lib
    lib = _lib.load()
    assert lib.aadg_aug_u8_workspace_bytes(144, 512, 512, 512) >= 2 * 144 * 512 * 512 * 3
    assert lib.aadg_aug_u8_workspace_bytes(0, 512, 512, 512) == 0
    assert lib.aadg_sinkhorn_workspace_bytes(18, 8, 128) >= 18 * 4
    assert lib.aadg_sinkhorn_workspace_bytes(1, 4096, 128) >= 4 * 4096 * 4096 * 4      # cost matrices in HBM


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    from aadg_amd import _lib
    from aadg_amd.data import basic
    img = torch.zeros((8, 8, 3), dtype=torch.uint8)
    with pytest.raises(_lib.AadgError):
        _lib.op_u8(img, 1)
    with pytest.raises(_lib.AadgError):
        basic.apply_augment(img, torch.zeros((8, 8), dtype=torch.uint8), "Invert", 0.0)
    with pytest.raises(_lib.AadgError):
        _lib.sinkhorn_rewards(torch.zeros((18, 128)), 3, 2, 3)


def test_entry_points_reject_bad_arguments_without_touching_the_gpu():
    """Every entry validates its arguments before enqueuing anything: NULL pointers / bad sizes -> AADG_E_BADARG (-1)."""
    import ctypes
    from aadg_amd import _lib
    lib = _lib.load()
    z = ctypes.c_void_p(0)
    assert lib.aadg_aug_u8_forward(z, z, 1, 8, 8, z, 1, 0, 8, 0, z, z, z, 0, z) == -1
    assert lib.aadg_op_u8(z, z, 8, 8, 1, 0, 0.0, z, z, 0, z) == -1
    assert lib.aadg_sinkhorn_rewards_f32(z, 3, 8, 6, 128, 0.05, 0.5, z, z, 0, z) == -1
    assert lib.aadg_sinkhorn_divergence_f32(z, 128, 128, z, z, z, 1, 8, 0.05, 0.5, z, z, 0, z) == -1
    assert lib.aadg_normalize_rewards_f32(z, 6, z, z) == -1
    assert lib.aadg_seg_bce_dice_f32(z, z, 6, 2, 16, 6, z, z, z, z, 0, z) == -1
    assert lib.aadg_fop_f32(0, z, z, z, 0, z, z, 1, 3, 8, 8, z, 0, z) == -1
    assert lib.aadg_upsample_bilinear2d(z, z, 1, 8, 8, 16, 16, 0, z) == -1
    one = ctypes.c_void_p(16)       # non-NULL dummy: size / enum checks come before any dereference
    assert lib.aadg_fop_f32(99, one, one, z, 0, z, z, 1, 3, 8, 8, z, 0, z) == -1
    assert lib.aadg_upsample_bilinear2d(one, one, 1, 8, 8, 16, 16, 7, z) == -1
    assert lib.aadg_aug_u8_forward(one, one, 1, 8, 8, one, 1, 9, 8, 0, one, one, one, 1 << 30, z) == -1      # max_ops > 4
    assert lib.aadg_aug_u8_forward(one, one, 1, 8, 8, one, 1, 2, 8, 0, one, one, one, 16, z) == -2           # workspace too small
    assert lib.aadg_seg_bce_dice_f32(one, one, 7, 2, 16, 6, one, one, z, one, 1 << 20, z) == -1              # N % M != 0


def test_backbone_and_controller_entry_points_validate_arguments():
    """The layer / controller entries added around the hot path: NULL -> -1, too-small workspace -> -2, shapes outside a
    kernel's tiling -> -3 (AADG_E_UNSUPPORTED); nothing is enqueued in any of these cases."""
    import ctypes
    from aadg_amd import _lib
    lib = _lib.load()
    z, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    f = ctypes.c_float
    # BatchNorm
    assert lib.aadg_bn_workspace_bytes(64) > 0 and lib.aadg_bn_workspace_bytes(0) == 0
    assert lib.aadg_bn_forward(z, z, z, z, z, z, z, z, f(0.1), f(1e-5), 1, 1, 2, 4, 16, 0, z, z, z, 0, 0, z) == -1
    assert lib.aadg_bn_forward(one, z, one, z, z, z, z, z, f(0.1), f(1e-5), 7, 1, 2, 4, 16, 0, one, one, one, 1 << 20, 0, z) == -1   # act
    assert lib.aadg_bn_forward(one, z, one, z, z, z, z, z, f(0.1), f(1e-5), 1, 1, 2, 4, 16, 0, one, one, one, 8, 0, z) == -2        # workspace
    assert lib.aadg_bn_mask_bytes(2, 4, 16, 1) == 2 * 4 * 2 and lib.aadg_bn_mask_bytes(2, 4, 12, 1) == 0
    assert lib.aadg_bn_backward(one, z, z, one, z, 0, z, z, z, one, one, 1, one, one, z, z, 2, 4, 16, 0, one, 1 << 20, 0, z) == -1   # dres without y / mask
    assert lib.aadg_bn_backward(one, one, z, one, z, 2, z, z, z, one, one, 1, one, one, z, z, 2, 4, 16, 0, one, 1 << 20, 0, z) == -1  # n_extra without pointers
    assert lib.aadg_bn_backward(one, one, z, one, one, 9, z, z, z, one, one, 1, one, one, z, z, 2, 4, 16, 0, one, 1 << 20, 0, z) == -1  # more than 6 extra gradients
    assert lib.aadg_bn_backward(one, z, z, one, z, 0, one, z, z, one, one, 1, one, z, z, z, 2, 4, 16, 0, one, 1 << 20, 0, z) == -1   # plane constants without dres
    # depthwise 3x3
    assert lib.aadg_dwconv3x3_supported(32, 32, 12, 1) == 1 and lib.aadg_dwconv3x3_supported(32, 12, 1, 1) == 0
    assert lib.aadg_dwconv3x3_supported(8, 512, 1, 0) == 0
    assert lib.aadg_dwconv3x3(z, z, z, 1, 1, 8, 8, 1, 0, 0, z) == -1
    assert lib.aadg_dwconv3x3(one, one, one, 1, 1, 8, 12, 1, 0, 1, z) == -3
    assert lib.aadg_dwconv3x3_wgrad(one, one, one, 1, 4, 8, 8, 1, 1, one, 4, z) == -2
    # 1x1 weight gradient
    assert lib.aadg_conv1x1_wgrad_supported(256, 64, 1024) == 1 and lib.aadg_conv1x1_wgrad_supported(256, 64, 100) == 0
    assert lib.aadg_conv1x1_wgrad_bf16(z, z, z, 1, 8, 8, 64, z) == -1
    assert lib.aadg_conv1x1_wgrad_bf16(one, one, one, 1, 8, 8, 48, z) == -3
    # BatchNorm + ReLU + max-pool
    assert lib.aadg_bn_relu_maxpool_supported(256, 256, 1) == 1 and lib.aadg_bn_relu_maxpool_supported(16, 16, 1) == 0
    assert lib.aadg_bn_relu_maxpool_forward(z, z, z, z, z, z, z, 0.1, 1e-5, 1, 4, 64, 64, 1, z, z, z, 0, z) == -1
    assert lib.aadg_bn_relu_maxpool_backward(z, z, z, z, z, z, z, z, z, z, 1, 4, 64, 64, 1, z, 0, z) == -1
    # 1x1 convolution forward / input gradient
    assert lib.aadg_conv1x1_nchw_supported(64, 256, 1024) == 1 and lib.aadg_conv1x1_nchw_supported(64, 12, 1024) == 0
    assert lib.aadg_conv1x1_nchw_bf16(z, z, z, 1, 8, 8, 64, z) == -1
    # ABI 8: the float32-precision (f32x3) entry points validate like their bfloat16 twins (null / misaligned pointers -1, shapes -3)
    assert lib.aadg_conv1x1_nchw_f32x3(z, z, z, z, 1, 8, 8, 64, z) == -1
    assert lib.aadg_conv1x1_nchw_f32x3(one, one, one, one, 1, 8, 12, 64, z) == -3          # K % 8
    assert lib.aadg_conv1x1_wgrad_f32x3(z, z, z, 1, 8, 8, 64, z) == -1
    assert lib.aadg_conv1x1_wgrad_f32x3(one, one, one, 1, 8, 8, 48, z) == -3               # HW % 32
    assert lib.aadg_conv3x3_nchw_f32x3(z, z, z, z, 1, 8, 8, 8, 32, 1, z) == -1
    assert lib.aadg_conv3x3_nchw_f32x3(one, one, one, one, 1, 8, 8, 8, 48, 1, z) == -3     # W not in {32, 64, 128}
    assert lib.aadg_conv3x3_wgrad_f32x3(one, one, one, 1, 8, 8, 8, 128, 2, z) == -3        # W = 128 with dilation 2: LDS
    assert lib.aadg_conv3x3s2_nchw_f32x3(z, z, z, z, 1, 8, 8, 8, 32, z) == -1
    assert lib.aadg_conv3x3s2_dgrad_f32x3(one, one, one, one, 1, 8, 12, 8, 32, z) == -3    # M % 8
    assert lib.aadg_conv3x3s2_wgrad_f32x3(one, one, one, 1, 8, 8, 8, 48, z) == -3
    assert lib.aadg_stem_conv7x7_f32x3(z, z, z, 1, 64, 64, z, 0, z) == -1
    assert lib.aadg_stem_conv7x7_f32x3(one, one, one, 1, 64, 64, one, 16, z) == -2         # workspace
    assert lib.aadg_stem_conv7x7_wgrad_f32x3(one, one, one, 1, 64, 72, z) == -3
    assert lib.aadg_weight_layouts_split_bf16(z, z, 1, z) == -1 and lib.aadg_weight_layouts_split_bf16(one, one, 0, z) == 0
    # stem convolution
    assert lib.aadg_stem_conv7x7_supported(512, 512) == 1 and lib.aadg_stem_conv7x7_supported(512, 520) == 0
    assert lib.aadg_stem_conv7x7_workspace_bytes() == 2 * 11 * 2 * 64 * 16           # ABI 8: hi + lo weight fragments
    assert lib.aadg_stem_conv7x7_bf16(z, 1, z, z, 1, 32, 32, z, 0, z) == -1
    assert lib.aadg_stem_conv7x7_wgrad_bf16(z, 1, z, z, 1, 32, 32, z) == -1
    # stride-2 sub-sampling
    assert lib.aadg_subsample2x2_supported(16, 32, 1) == 1 and lib.aadg_subsample2x2_supported(16, 24, 1) == 0
    assert lib.aadg_subsample2x2_supported(16, 24, 0) == 1 and lib.aadg_subsample2x2_supported(15, 32, 0) == 0
    assert lib.aadg_subsample2x2(z, z, 1, 16, 32, 1, z) == -1
    assert lib.aadg_subsample2x2_backward(z, z, 1, 16, 32, 1, z) == -1
    # max pooling
    assert lib.aadg_maxpool3x3s2_supported(16, 16) == 1 and lib.aadg_maxpool3x3s2_supported(16, 12) == 0
    assert lib.aadg_maxpool3x3s2_forward(z, z, z, 1, 16, 16, 0, z) == -1
    assert lib.aadg_maxpool3x3s2_index_bytes(3, 16, 16) == 3 * 8 * 8 and lib.aadg_maxpool3x3s2_index_bytes(3, 16, 12) == 0
    assert lib.aadg_maxpool3x3s2_backward(one, one, one, 1, 16, 12, 0, z) == -3
    # up-sampling backward
    assert lib.aadg_upsample_bilinear2d_backward_supported(32, 32, 128, 128) == 1
    assert lib.aadg_upsample_bilinear2d_backward_supported(4, 4, 256, 256) == 0          # factor 64: rectangle exceeds the LDS tile
    assert lib.aadg_upsample_bilinear2d_backward(one, one, 1, 32, 32, 128, 128, 0, one, 8, z) == -2
    assert lib.aadg_upsample_bilinear2d_backward(z, z, 1, 32, 32, 128, 128, 0, z, 0, z) == -1
    # controller
    dims = (6, 5, 4, 32, 100, 10, 10)
    assert lib.aadg_controller_supported(*dims) == 1 and lib.aadg_controller_workspace_bytes(*dims) > 0
    assert lib.aadg_controller_supported(6, 9, 4, 32, 100, 10, 10) == 0                  # more sub-policies than a workgroup runs
    assert lib.aadg_controller_supported(6, 5, 4, 32, 200, 10, 10) == 0                  # 4H gate rows > 512 lanes
    assert lib.aadg_controller_sample_f32(z, *dims, f(1.25), z, z, z, z, z, z, z, 0, z) == -1
    assert lib.aadg_controller_sample_f32(one, 6, 9, 4, 32, 100, 10, 10, f(1.25), one, one, one, one, one, one, one, 1 << 30, z) == -3
    assert lib.aadg_controller_ppo_update_f32(one, one, one, *dims, f(1.25), one, one, one, f(0.2), 5, 0, f(3.5e-4), f(0.9), f(0.999),
                                              f(1e-8), one, one, 16, z) == -2
    # embedding prologue
    assert lib.aadg_embed_prologue_f32(z, 8, 1, 8, z, z, 128, z, z, 0, f(0.2), z, z, z) == -1
    assert lib.aadg_embed_prologue_f32(one, 5000, 1, 5000, one, one, 128, z, z, 0, f(0.2), one, z, z) == -3


def test_round2_entry_points_validate_arguments():
    """aadg_aug_u8_forward_ex2 (work lists), aadg_bn_sync_* (phase split), the norm-passing embed / Sinkhorn pair and
    aadg_upsample_sum reject bad arguments on the host, without a GPU."""
    import ctypes
    from aadg_amd import _lib
    lib = _lib.load()
    z, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    lists = _lib.AugLists()
    lists.n_stat[0] = 5                                  # a non-empty statistics list without its array
    assert lib.aadg_aug_u8_forward_ex2(one, one, 1, 8, 8, one, 4, 2, 8, 0, one, one, one, 1 << 30, z, 0, -1, z, z, ctypes.byref(lists)) == -1
    lists = _lib.AugLists()
    lists.order, lists.n_plain, lists.n_sharp = 16, 3, 3  # more listed units than N
    assert lib.aadg_aug_u8_forward_ex2(one, one, 1, 8, 8, one, 4, 2, 8, 0, one, one, one, 1 << 30, z, 0, -1, z, z, ctypes.byref(lists)) == -1
    args = (one, z, one, z, z, z, z, z, 0.1, 1e-5, 0, 2, 4, 16, 0, one, one, one, one, 1 << 20, 0, z)
    assert lib.aadg_bn_sync_forward(0, *args) == -1 and lib.aadg_bn_sync_forward(3, *args) == -1          # phase must be 1 or 2
    assert lib.aadg_bn_sync_forward(1, *(args[:17] + (z,) + args[18:])) == -1                            # no sums buffer
    assert lib.aadg_embed_prologue_norm_f32(one, 8, 2, 8, one, one, 4, z, z, 0, 0.2, one, z, z, z) == -1   # norm output missing
    assert lib.aadg_sinkhorn_rewards_norm_f32(one, z, 3, 2, 2, 8, 0.05, 0.5, one, one, 1024, z) == -1
    assert lib.aadg_upsample_sum(one, z, z, z, 4, one, 4, 8, 8, 0, z) == -1                                # n_low > 3
    assert lib.aadg_upsample_sum_backward(z, one, 4, 2, 2, 8, 8, 1, z) == -1
    assert lib.aadg_pool_histograms_u8(z, 1, 8, 8, one, z) == -1 and lib.aadg_pool_histograms_u8(one, 0, 8, 8, one, z) == -1
    lists = _lib.AugLists()
    lists.order, lists.n_generic, lists.n_generic_sharp = 16, 2, 3   # more stencil units than down-scaling units
    assert lib.aadg_aug_u8_forward_ex2(one, one, 1, 8, 8, one, 4, 2, 8, 0, one, one, one, 1 << 30, z, 0, -1, z, z, ctypes.byref(lists)) == -1
    lists = _lib.AugLists()
    lists.stat_units[1], lists.n_stat[1], lists.n_stat_stencil[1] = 16, 2, 3   # ABI 7: more stencil units than the list holds
    assert lib.aadg_aug_u8_forward_ex2(one, one, 1, 8, 8, one, 4, 2, 8, 0, one, one, one, 1 << 30, z, 0, -1, z, z, ctypes.byref(lists)) == -1
    lists = _lib.AugLists()
    lists.gen_chunk = -1
    assert lib.aadg_aug_u8_forward_ex2(one, one, 1, 8, 8, one, 4, 2, 8, 0, one, one, one, 1 << 30, z, 0, -1, z, z, ctypes.byref(lists)) == -1
    assert ctypes.sizeof(_lib.AugLists) == 8 + 16 + 8 * _lib.MAX_OPS + 4 * _lib.MAX_OPS + 8 + 8 + 8 + 4 * _lib.MAX_OPS + 8 + 8   # mirror of aadg_aug_lists (n_generic_sharp fills the padding behind n_late; ABI 7: n_stat_stencil, gen_chunk; ABI 9: n_generic_wonly in the former tail padding; ABI 12: n_plain_early, n_sharp_early)
    lists = _lib.AugLists()
    lists.order, lists.n_generic, lists.n_generic_sharp, lists.n_generic_wonly = 16, 3, 2, 2      # ABI 9: more width-only units than plain generic ones
    assert lib.aadg_aug_u8_forward_ex2(one, one, 1, 8, 8, one, 4, 2, 8, 0, one, one, one, 1 << 30, z, 0, -1, z, z, ctypes.byref(lists)) == -1
    lists = _lib.AugLists()
    lists.order, lists.n_plain, lists.n_plain_early = 16, 2, 3                                     # ABI 12: more early units than the class holds
    assert lib.aadg_aug_u8_forward_ex2(one, one, 1, 8, 8, one, 4, 2, 8, 0, one, one, one, 1 << 30, z, 0, -1, z, z, ctypes.byref(lists)) == -1


def test_round3_entry_points_validate_arguments():
    """aadg_layernorm_* and aadg_dwconv3x3_gelu_nhwc_* (SegFormer blocks) reject bad arguments on the host, without a GPU; the float-op
    entry point keeps its contract (workspace only for the statistics ops)."""
    import ctypes
    from aadg_amd import _lib
    lib = _lib.load()
    z, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    f = ctypes.c_float
    assert lib.aadg_layernorm_supported(100, 320, 1) == 1 and lib.aadg_layernorm_supported(100, 36, 1) == 0
    assert lib.aadg_layernorm_supported(100, 520, 0) == 0 and lib.aadg_layernorm_supported(100, 64, 2) == 0
    assert lib.aadg_layernorm_workspace_bytes(1000, 128) > 0
    assert lib.aadg_layernorm_forward(z, z, z, 0, one, one, f(1e-6), z, one, one, one, 8, 64, 1, z) == -1            # no input
    assert lib.aadg_layernorm_forward(one, one, z, 0, one, one, f(1e-6), z, one, one, one, 8, 64, 1, z) == -1        # residual without s_out
    assert lib.aadg_layernorm_forward(one, z, z, 0, one, one, f(1e-6), z, one, one, one, 8, 36, 1, z) == -3          # C % 8 != 0
    assert lib.aadg_layernorm_forward(one, one, one, 3, one, one, f(1e-6), one, one, one, one, 8, 64, 1, z) == -1    # rows not a multiple of rows_per_sample
    assert lib.aadg_layernorm_backward(one, one, z, one, one, one, z, 0, one, z, one, one, one, 16, 1000, 128, 1, z) == -2   # workspace too small
    assert lib.aadg_dwconv3x3_gelu_nhwc_supported(2, 16, 16, 64, 1) == 1 and lib.aadg_dwconv3x3_gelu_nhwc_supported(2, 16, 16, 60, 1) == 0
    assert lib.aadg_dwconv3x3_gelu_nhwc_forward(one, one, z, one, 2, 16, 16, 64, 1, z) == -1                          # no bias
    assert lib.aadg_dwconv3x3_gelu_nhwc_forward(one, one, one, one, 2, 16, 16, 60, 1, z) == -3
    assert lib.aadg_dwconv3x3_gelu_nhwc_backward(one, one, one, one, z, one, one, one, 2, 16, 16, 64, 1, z) == -1     # no scratch for g
    assert lib.aadg_fop_workspace_bytes(144, 512, 512) > 0
    assert lib.aadg_fop_f32(_lib.FOP["contrast"], one, one, one, 1, z, z, 2, 3, 8, 8, z, 0, z) == -1                  # statistics op without workspace
    assert lib.aadg_fop_f32(_lib.FOP["contrast"], one, one, one, 1, z, z, 2, 3, 8, 8, one, 8, z) == -2                # ... with a too small one


def test_host_planner_matches_python_statement():
    """aadg_aug_u8_plan (host code of the library: validation + work lists of one augmentation call) against the Python statement of the
    same rules (_lib.launch_plan / validate_units) on random unit records of every flow, and its refusal of bad records."""
    import ctypes
    import numpy as np
    from aadg_amd import _lib
    from helpers import random_units
    lib = _lib.load()
    K = _lib.MAX_OPS
    rs = np.random.RandomState(11)
    seen_mixed = seen_wonly = False
    for (H, crop, sr, L, N) in ((64, 64, (1.0, 1.5), 2, 97), (96, 64, (0.5, 2.0), 4, 160), (128, 96, (0.34, 0.6), 3, 50), (33, 31, (1.0, 1.5), 2, 40),
                                (64, 64, (0.5, 2.0), 4, 1)):
        P = 7
        units = random_units(rs, N, P, H, H, crop, sr, L=L)
        # more Sharpness (stencil classes) and statistics ops in late slots than the uniform op draw gives
        for i in range(0, N, 4):
            u = units[i]
            for k in range(int(u["n_ops"])):
                if rs.rand() < 0.5:
                    u["op"][k] = 8
                    u["farg"][k] = np.float32(1.0 if rs.rand() < 0.2 else 1.5)
        cont = np.ascontiguousarray(units)
        order = np.full(N, -1, np.int32)
        stat = np.full((K, N), -1, np.int32)
        late = np.full(N, -1, np.int32)
        summary = (ctypes.c_int32 * (11 + 2 * K))()
        rc = lib.aadg_aug_u8_plan(cont.ctypes.data, N, P, H, H, crop, order.ctypes.data, stat.ctypes.data, late.ctypes.data, summary)
        assert rc == 0
        classes, stats_mask, want_order, counts, stat_lists, want_late, n_sten, n_wonly, n_cls_early = _lib.launch_plan(units, H, H, crop)
        assert (summary[9 + 2 * K], summary[10 + 2 * K]) == n_cls_early    # ABI 12: the units that wait for no statistics pass lead the plain / Sharpness class
        for c0, nc, ne in ((0, counts[0], n_cls_early[0]), (counts[0], counts[1], n_cls_early[1])):
            assert set(order[c0 + ne:c0 + nc].tolist()) <= set(want_late.tolist()) and not set(order[c0:c0 + ne].tolist()) & set(want_late.tolist())
        assert tuple(summary[:4]) == tuple(counts) and summary[5] == classes and summary[6] == stats_mask
        assert summary[8 + 2 * K] == n_wonly                             # ABI 9: the width-only units lead the generic run
        seen_wonly = seen_wonly or 0 < n_wonly < counts[2] - counts[3]
        assert summary[7] == _lib.validate_units(units, P, H, H)
        assert np.array_equal(order, want_order)
        for k in range(K):
            assert summary[8 + k] == stat_lists[k].size and np.array_equal(stat[k, :stat_lists[k].size], stat_lists[k])
            assert summary[8 + K + k] == n_sten[k]                       # ABI 7: stencil units first
        seen_mixed = seen_mixed or any(0 < n_sten[k] < stat_lists[k].size for k in range(1, K))
        assert summary[4] == want_late.size and np.array_equal(late[:want_late.size], want_late)
    assert seen_mixed                 # a list with both kinds of unit was among the cases
    assert seen_wonly                 # ... and a generic run with both width-only and other plain units

    def refused(mutate):
        units = random_units(np.random.RandomState(3), 8, 4, 64, 64, 64, (1.0, 1.5))
        units["n_ops"][:] = 2
        mutate(units)
        cont = np.ascontiguousarray(units)
        bufs = [np.zeros(8 * (K if i == 1 else 1), np.int32) for i in range(3)]
        rc = lib.aadg_aug_u8_plan(cont.ctypes.data, 8, 4, 64, 64, 64, bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data,
                                  (ctypes.c_int32 * (11 + 2 * K))())
        with pytest.raises(_lib.AadgError):
            _lib.validate_units(units, 4, 64, 64)
        return rc
    bad = _lib.load().aadg_aug_u8_plan(None, 1, 1, 8, 8, 8, None, None, None, None)
    assert bad != 0

    def src(u): u["src"][3] = 4
    def nops(u): u["n_ops"][2] = K + 1
    def opid(u): u["op"][5][1] = 10
    def scale(u): u["scaled_w"][1] = 21
    def rect(u): u["op"][0][0] = 9; u["rect"][0][0] = (0, 0, 64, 10)
    def bits(u): u["op"][6][1] = 4; u["iarg"][6][1] = 9
    for m in (src, nops, opid, scale, rect, bits):
        assert refused(m) != 0, m.__name__
