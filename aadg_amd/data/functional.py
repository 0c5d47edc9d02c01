"""Float tensor ops -- API mirror of the reference's data/functional.py (__all__ :9-11, tensor_function
:49-73 and the 19 ops :110-280).  Each function takes `img` float32 [B,3,H,W] (or [3,H,W]) in [0,1] on the
GPU and a magnitude (scalar tensor or one value per sample) and returns a clamped tensor; the work is done
by the HIP kernels in csrc/tensor_ops.hip through `aadg_fop_f32`.

Forward semantics only: the reference wraps non-differentiable ops in a straight-through estimator
(data/functional.py:21-46); `ste` is provided with the same signature, and the kernels' outputs carry the
STE gradient to `mag` where the reference does (solarize, posterize), via a thin autograd wrapper.
"""
import functools
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from .. import _lib
from .kernels import get_gaussian_3x3kernel, get_sharpness_kernel

__all__ = ['shear_x', 'shear_y', 'translate_x', 'translate_y', 'hflip', 'vflip', 'rotate', 'invert', 'solarize',
           'posterize', 'gray', 'contrast', 'auto_contrast', 'saturate', 'brightness', 'hue', 'sample_pairing',
           'equalize', 'sharpness']


class _STE(Function):
    """Straight-through estimator: forward = first argument, gradient flows to the second."""

    @staticmethod
    def forward(ctx, input_forward, input_backward):
        ctx.shape = input_backward.shape
        return input_forward

    @staticmethod
    def backward(ctx, grad_in) -> Tuple[None, torch.Tensor]:
        return None, grad_in.sum_to_size(ctx.shape)


def ste(input_forward: torch.Tensor, input_backward: torch.Tensor) -> torch.Tensor:
    return _STE.apply(input_forward, input_backward).clone()


def tensor_function(func):
    """Input checks of the reference's decorator: img must be a tensor, 3-D is promoted to 4-D, `mag` must
    have 1 or B elements; the output is clamped to [0,1] (done inside the kernels)."""

    @functools.wraps(func)
    def inner(*args):
        if len(args) == 1:
            img, mag, kernel = args[0], None, None
        elif len(args) == 2:
            (img, mag), kernel = args, None
        else:
            img, mag, kernel = args
        if not torch.is_tensor(img):
            raise RuntimeError(f'img is expected to be torch.Tensor, but got {type(img)} instead')
        if img.dim() == 3:
            img = img.unsqueeze(0)
        if torch.is_tensor(mag) and mag.nelement() != 1 and mag.size(0) != img.size(0):
            raise RuntimeError('Shape of `mag` is expected to be `1` or `B`')
        if mag is not None and not torch.is_tensor(mag):
            mag = torch.tensor([float(mag)], device=img.device)
        img = img.contiguous().float()
        return func(img, mag, kernel) if len(args) == 3 else func(img, mag)

    return inner


def _k(name, img, mag=None, kernel=None, perm=None):
    return _lib.fop(name, img, None if mag is None else mag.detach().to(img.device), kernel, perm)


# geometric ------------------------------------------------------------------------------------------------
@tensor_function
def shear_x(img, mag):
    return _k("shear_x", img, mag)


@tensor_function
def shear_y(img, mag):
    return _k("shear_y", img, mag)


@tensor_function
def translate_x(img, mag):
    return _k("translate_x", img, mag)


@tensor_function
def translate_y(img, mag):
    return _k("translate_y", img, mag)


@tensor_function
def hflip(img, _=None):
    return _k("hflip", img)


@tensor_function
def vflip(img, _=None):
    return _k("vflip", img)


@tensor_function
def rotate(img, mag):
    return _k("rotate", img, mag)


# colour -----------------------------------------------------------------------------------------------------
@tensor_function
def invert(img, _=None):
    return _k("invert", img)


@tensor_function
def solarize(img, mag):
    return ste(_k("solarize", img, mag), mag.view(-1, 1, 1, 1)).clamp_(0, 1)


@tensor_function
def posterize(img, mag):
    return ste(_k("posterize", img, mag), mag.view(-1, 1, 1, 1)).clamp_(0, 1)


@tensor_function
def gray(img, _=None):
    return _k("gray", img)


@tensor_function
def contrast(img, mag):
    return _k("contrast", img, mag)


@tensor_function
def auto_contrast(img, _=None):
    return _k("auto_contrast", img)


@tensor_function
def saturate(img, mag):
    return _k("saturate", img, mag)


@tensor_function
def brightness(img, mag):
    return _k("brightness", img, mag)


@tensor_function
def hue(img, mag):
    return _k("hue", img, mag)


@tensor_function
def sample_pairing(img, mag, indices: Optional[torch.Tensor] = None):
    # the reference draws torch.randperm(B) internally (data/functional.py:236); an explicit permutation can
    # be passed as the third argument for reproducible tests
    if indices is None:
        indices = torch.randperm(img.size(0), device=img.device, dtype=torch.long)
    return _k("sample_pairing", img, mag, perm=indices)


@tensor_function
def equalize(img, _=None):
    return _k("equalize", img)


@tensor_function
def sharpness(img, mag, kernel: Optional[torch.Tensor] = None):
    if kernel is None:
        kernel = get_sharpness_kernel(img.device)
    return _k("sharpness", img, mag, kernel=kernel.to(img.device))


@tensor_function
def gaussian_blur3x3(img, mag, kernel: Optional[torch.Tensor] = None):
    if kernel is None:
        kernel = get_gaussian_3x3kernel(mag, img.device)
    return _k("gaussian_blur3x3", img, mag, kernel=kernel.to(img.device))


@tensor_function
def cutout(img, mag):
    raise NotImplementedError
