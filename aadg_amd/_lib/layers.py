"""autograd Functions of the streaming backbone layers between the convolutions: bilinear up-sampling (+ concatenation), depthwise 3x3, max-pool,
sub-sampling, and SegFormer's depthwise + GELU / add + LayerNorm (csrc/upsample*.hip, depthwise.hip, maxpool.hip, subsample.hip, dwconv_nhwc.hip, layernorm.hip)."""
import ctypes
import os

import numpy as np
import torch

from .binding import AadgError, _check, _ptr, _require_cuda, _stream, load, workspace
from .batchnorm import _BN_DTYPES, _bn_ws


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
