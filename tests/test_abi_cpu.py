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
    assert lib.aadg_abi_version() == 1


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
