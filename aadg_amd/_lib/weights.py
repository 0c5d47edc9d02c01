"""Shadows of the float32 master weights in the layouts the convolution kernels read (bfloat16, or (hi, lo) bfloat16 halves for f32x3), refreshed once per
forward in one launch (csrc/weight_layouts.hip)."""
import ctypes
import os

import numpy as np
import torch

from .binding import AadgError, _check, _stream, load


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
