"""autograd Functions of the own matrix-core convolutions: 1x1 / 3x3 / stride-2 3x3 / the 7x7 stem, in bfloat16 and in the float32-precision "f32x3" arithmetic
(csrc/conv1x1_*.hip, conv3x3_*.hip, stem_conv.hip)."""
import ctypes
import os

import numpy as np
import torch

from .binding import AadgError, CONV1X1_SMALL_BATCH, CONV3X3_OWN_FWD_MAX_N, _check, _ptr, _require_cuda, _stream, load, workspace
from .batchnorm import _BN_DTYPES
from .wgrad import _wgrad_beside
from .weights import _ShadowRef, cast_weight, split_layout, split_weight, weight_layout


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
        ctx.set_materialize_grads(False)       # no zero-filled gradient tensor for `sums` in every backward pass (a fill launch each)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:                           # (set_materialize_grads(False): the output took no part in the loss)
            return (None,) * 5
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
        ctx.set_materialize_grads(False)       # no zero-filled gradient tensor for `sums` in every backward pass (a fill launch each)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:                           # (set_materialize_grads(False): the output took no part in the loss)
            return (None,) * 6
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
