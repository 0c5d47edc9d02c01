"""ctypes binding of libaadg_hip.so (include/aadg_hip.h) for torch device tensors, and the autograd Functions over it.

torch is used only as plumbing here: device memory, the current HIP stream, and dtype/shape checks.
There is deliberately NO CPU fallback: every wrapper raises if the shared library or a GPU is
missing, so a silent eager path can never stand in for the HIP kernels.

One namespace, eight modules by concern (round 6; it was one 2.9 k-line file):

    binding    the loader, the symbol table, errors, stream / pointer / workspace helpers
    aug        the fused uint8 augmentation call and its planner statement
    hotpath    Sinkhorn rewards, BCE + Dice, the float tensor ops, the fused controller, the embedding prologue
    layers     up-sampling, depthwise, pooling, sub-sampling, LayerNorm: the streaming layers of the backbones
    batchnorm  every BatchNorm Function (fused, synchronised, lazy / on-load, shortcut pair, stem)
    wgrad      the weight-gradient side stream
    weights    bfloat16 / split shadows of the master weights
    conv       the matrix-core convolutions (bfloat16 and f32x3)

`from aadg_amd import _lib; _lib.<anything>` keeps working: every name of every module is re-exported here, and an ASSIGNMENT through this
namespace (`_lib.BN_SYNC_REDUCE = f`, `_lib.PROFILE_EVENTS = pair`, a test's `monkeypatch.setattr(_lib, "conv1x1", g)`) is forwarded to
every module that holds the name -- the modules behave as the one namespace they were split from.
"""
import ctypes
import os
import sys
import types

import numpy as np
import torch

from . import binding, aug, hotpath, layers, batchnorm, wgrad, weights, conv

_MODULES = (binding, aug, hotpath, layers, batchnorm, wgrad, weights, conv)

for _m in _MODULES:
    for _k, _v in vars(_m).items():
        if not _k.startswith("__") and not isinstance(_v, types.ModuleType):
            globals().setdefault(_k, _v)


class _OneNamespace(types.ModuleType):
    def __setattr__(self, name, value):
        for m in _MODULES:
            if name in vars(m):
                setattr(m, name, value)
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _OneNamespace
del _m, _k, _v
