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
    assert lib.aadg_sinkhorn_workspace_bytes(18, 8) >= 18 * 4


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
