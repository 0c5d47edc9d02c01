"""BatchNorm autograd Functions over csrc/batchnorm.hip: fused normalise + activation + residual, the phase-split variants with synchronised statistics
(reference: models/sync_batchnorm/batchnorm.py:102-105), the lazy variant whose elementwise pass the consuming convolution applies on load, the projection-shortcut
pair, BatchNorm + ReLU + max-pool of the stem."""
import ctypes
import os

import numpy as np
import torch

from .binding import AadgError, _EXP, _POOL_BWD_FUSED_F32, _check, _ptr, _ptr_array, _require_cuda, _stream, load, workspace


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
    from .. import distributed as adist
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
        ctx.set_materialize_grads(False)       # no zero-filled gradient tensors for scale / shift in every backward pass (two fill launches)
        return x.view_as(x), scale, shift

    @staticmethod
    def backward(ctx, dz, *unused):
        if dz is None:
            return (None,) * 10
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
