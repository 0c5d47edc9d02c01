"""DeepLabV3+ segmentation backbone with the `(mask_logits, pooled_encoder_feature)` contract the inner
loop relies on (reference: models/__init__.py:17-23 builds smp.DeepLabV3Plus(..., aux_params=dict(
pooling='avg')) and models/heads.py:14-25 reduces the aux head to pool+flatten, so `model(x)` returns
`(logits [N,K,H,W], feature [N,C_enc])`, search_dg.py:132).

segmentation_models_pytorch / torchvision / timm are not in this image, so the architecture is written
here in plain torch.nn (MIOpen convolutions; the backbone is host plumbing, not a parity target --
SURVEY.md 8c).  Same topology as smp 0.2.0's DeepLabV3Plus: encoder at output stride 16 (last stage
dilated), ASPP with separable atrous convs (12, 24, 36) + image pooling, 48-channel skip from the
stride-4 feature, separable 3x3 fuse, 1x1 classifier, x4 bilinear upsampling (align_corners=True).
Encoders: mobilenet_v2 (C_enc = 1280, the reference's only reachable choice) and resnet50
(C_enc = 2048, BASELINE config 2).  Weights are randomly initialised (no network for ImageNet weights).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _upsample_ac(x, size):
    """Bilinear, align_corners=True.  On the GPU the forward is the HIP streaming kernel (csrc/upsample.hip);
    the CPU branch only serves shape tests of this host-plumbing module."""
    if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
        from .. import _lib
        return _lib.upsample_bilinear_ac(x, size)
    return F.interpolate(x, size=size, mode='bilinear', align_corners=True)


_ACT_CODE = {None: 0, 'relu': 1, 'relu6': 2}
_BN_SYNC = False


# BatchNorm's `num_batches_tracked += 1` is one 5 us launch per layer and forward (62 of them); the layers of a model registered
# with `batch_step_bookkeeping` hand their counters to a list during the forward and one multi-tensor add bumps them afterwards.
_PENDING_BN_COUNTERS = []


def _bump(bn):
    if getattr(bn, '_aadg_deferred_counter', False):
        _PENDING_BN_COUNTERS.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


F32X3_PATHS = {}         # layer name -> "own" | "library: <reason>" (f32x3 mode; filled by the first forward passes, `f32x3_coverage`)


def _took(mod, path, x=None):
    name = getattr(mod, '_aadg_name', None)
    if name is not None and name not in F32X3_PATHS:
        F32X3_PATHS[name] = path if x is None else "%s (input %s, weight %s)" % (path, tuple(x.shape), tuple(mod.weight.shape))


def f32x3_coverage():
    """(layers on the own float32-precision kernels, {layer: reason} of those that fell back to the library's float32 convolution)"""
    own = sorted(k for k, v in F32X3_PATHS.items() if v == "own")
    return own, {k: v for k, v in F32X3_PATHS.items() if v != "own"}


def batch_step_bookkeeping(model, f32x3=False):
    """Per-forward bookkeeping of a CUDA model in two launches instead of ~130: the bfloat16 casts of the own convolutions' master
    weights (a pre-hook: _lib.track_bf16_weights) and the BatchNorm counters of its layers (a post-hook).
    f32x3 = True: the model runs on float32 activations (no autocast) and its 1x1 / 3x3 convolutions take the own kernels' float32-
    precision instantiations (three bfloat16 matrix-core products per pair of (hi, lo)-split operands, _lib.conv1x1_x3 / conv3x3_x3);
    the tracked shadows are then the (hi, lo) halves of the weights.  False: float32 inputs go to the library's float32 convolutions."""
    from .. import _lib
    for m in model.modules():
        if isinstance(m, (Conv1x1, Conv3x3, SeparableConv2d, StemConv7x7, DeepLabV3Plus)):
            m.f32x3 = bool(f32x3)
    if f32x3:
        mark_bn_producers(model)
        F32X3_PATHS.clear()
        for name, m in model.named_modules():
            if isinstance(m, (Conv1x1, Conv3x3, StemConv7x7)):
                m._aadg_name = name
        if not any(isinstance(m, (Conv1x1, Conv3x3, StemConv7x7)) for m in model.modules()):
            import sys
            print("aadg_amd: --backbone_dtype f32x3: %s has no layer the own float32-precision convolution kernels cover; it runs on the "
                  "library's float32 kernels (= --backbone_dtype fp32)" % type(model).__name__, file=sys.stderr)
    _lib.track_bf16_weights(model, (Conv1x1, Conv3x3), split=bool(f32x3))
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m._aadg_deferred_counter = True

    def before(mod, args):
        del _PENDING_BN_COUNTERS[:]

    reported = []

    def after(mod, args, out):
        if _PENDING_BN_COUNTERS:
            torch._foreach_add_(_PENDING_BN_COUNTERS, 1)
            del _PENDING_BN_COUNTERS[:]
        if f32x3 and not reported and mod.training and F32X3_PATHS:
            # once, after the first training forward: say which convolutions did NOT take the own kernels, and why (run.py's default)
            reported.append(True)
            own, lib = f32x3_coverage()
            import sys
            print("aadg_amd: f32x3: %d convolution layers on the own float32-precision matrix-core kernels, %d on the library's float32 "
                  "convolution%s" % (len(own), len(lib), "".join("\n  %s: %s" % kv for kv in sorted(lib.items()))), file=sys.stderr)

    model.register_forward_pre_hook(before)
    model.register_forward_hook(after)


def mark_bn_producers(model):
    """Sets `bn_stats` on every convolution whose output goes straight into a BatchNorm of this module tree -- the bottleneck's
    conv1 / conv2 / conv3, a (convolution, BNAct) pair inside an nn.Sequential (projection shortcuts, ASPP branches, the decoder) and the pointwise
    half of a SeparableConv2d in such a pair: in f32x3 training mode those convolutions take the BatchNorm statistics of their output in
    the kernel epilogue (csrc/conv1x1_fwd.hip) and bn_act skips the statistics pass."""
    n = 0
    for m in model.modules():
        if isinstance(m, Bottleneck):
            m.conv1.bn_stats = m.conv2.bn_stats = m.conv3.bn_stats = True      # (conv2: at stride 1, shapes of the whole-tile kernel)
            n += 3
        if isinstance(m, nn.Sequential):
            mods = list(m)
            for a_, b_ in zip(mods[:-1], mods[1:]):
                if type(b_) is BNAct:
                    c = a_[1] if isinstance(a_, SeparableConv2d) else a_
                    if isinstance(c, Conv1x1):
                        c.bn_stats = True
                        n += 1
    return n


def set_bn_sync(flag):
    """Data-parallel ranks: training-mode BatchNorm statistics are all-reduced over the process group between the HIP
    statistics and normalisation kernels (_lib._SyncBatchNormAct) -- the modules stay nn.BatchNorm2d."""
    global _BN_SYNC
    _BN_SYNC = bool(flag)


def _take_bn_sums(x):
    """The float64 totals the producing convolution attached to its output (`_aadg_bn_sums`), for the ONE BatchNorm that consumes them.
    With synchronised statistics the consumer all-reduces them in place, so the attribute is cleared on the way out: a second reader
    gets None (and takes its own statistics pass) instead of totals that were already summed over the ranks."""
    sums = getattr(x, '_aadg_bn_sums', None)
    if sums is not None and _BN_SYNC:
        x._aadg_bn_sums = None
    return sums


def bn_act(bn, x, act=None, residual=None, handles=1, out=None, res_affine=None):
    """act(bn(x) [+ residual]).  On the GPU a plain `nn.BatchNorm2d` runs as the fused HIP streaming kernels
    (csrc/batchnorm.hip: statistics, normalise + activation + residual add in one pass, two-pass backward); anything
    else (SyncBatchNorm after `--sync_bn`, CPU shape tests) takes the module's own path."""
    if type(bn) is nn.BatchNorm2d and x.is_cuda and bn.momentum is not None and bn.track_running_stats:
        from .. import _lib
        xc = x.contiguous()
        rc = residual.contiguous() if residual is not None else None
        if _lib.bn_act_supported(xc, rc):
            if bn.training:
                _bump(bn)
            # statistics the producing convolution took in its epilogue (Conv1x1 in f32x3 mode, `bn_stats`): no statistics pass over x
            presums = _take_bn_sums(x) if bn.training else None
            # out (training only): a slice of a concatenation buffer (_lib.concat_slices) that receives the result
            return _lib.batch_norm_act(xc, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum,
                                       bn.eps, _ACT_CODE[act], rc,
                                       handles=handles if (bn.training and torch.is_grad_enabled() and rc is not None) else 1,
                                       out=out if bn.training else None, sync=_BN_SYNC and bn.training, presums=presums,
                                       res_affine=res_affine)
    assert res_affine is None, "res_affine: the fused training path only (Bottleneck._shortcut_on_load checks its conditions)"
    if _BN_SYNC and bn.training and type(bn) is nn.BatchNorm2d:
        # the DDP wrapper runs with broadcast_buffers=False on the assumption that EVERY BatchNorm layer synchronises its statistics
        # through the HIP path above; a layer that falls through to plain bn(x) would silently use per-rank statistics
        raise RuntimeError("--sync_bn: this BatchNorm2d cannot take the synchronised HIP path (momentum=None, track_running_stats=False, "
                           "a CPU tensor or an unsupported dtype / shape: %s %s); use torch.nn.SyncBatchNorm for it"
                           % (x.dtype, tuple(x.shape)))
    y = bn(x)
    if residual is not None:
        y = y + residual
    if act == 'relu':
        y = F.relu(y)
    elif act == 'relu6':
        y = F.relu6(y)
    return y


class BNAct(nn.Module):
    """BatchNorm2d followed by ReLU / ReLU6 / nothing, fused on the GPU (see `bn_act`)."""

    def __init__(self, c, act='relu'):
        super().__init__()
        self.bn = nn.BatchNorm2d(c)
        self.act = act

    def forward(self, x):
        return bn_act(self.bn, x, self.act)


def _bn_relu(c):
    return [BNAct(c, 'relu')]


class DepthwiseConv3x3(nn.Conv2d):
    """groups = channels 3x3 convolution; stride 1 with 'same' padding runs as the HIP LDS-tiled kernel on the GPU
    (csrc/depthwise.hip: float32 master weights, no cast kernel), anything else takes nn.Conv2d's path."""

    def __init__(self, c, stride=1, dilation=1):
        super().__init__(c, c, 3, stride=stride, padding=dilation, dilation=dilation, groups=c, bias=False)

    def forward(self, x):
        if x.is_cuda and self.stride == (1, 1):
            from .. import _lib
            xc = x.contiguous()
            if _lib.dwconv3x3_supported(xc, self.weight, self.dilation[0]):
                return _lib.dwconv3x3(xc, self.weight, self.dilation[0])
        return super().forward(x)


class Conv1x1(nn.Conv2d):
    """Pointwise convolution without bias.  bfloat16 activations on the GPU: forward / input gradient stay the library
    GEMMs on NCHW, the weight gradient runs on the matrix cores straight from the NCHW tensors
    (csrc/conv1x1_wgrad.hip) instead of transposing both activations to NHWC first."""

    f32x3 = False           # float32 inputs: the own float32-precision kernels instead of the library's (batch_step_bookkeeping)
    bn_stats = False        # f32x3, training: a BatchNorm follows -- its statistics come out of this convolution's epilogue (mark_bn_producers)

    def __init__(self, cin, cout, stride=1):
        super().__init__(cin, cout, 1, stride=stride, bias=False)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and self.f32x3 and self.stride in ((1, 1), (2, 2)):
            from .. import _lib
            xc = x.contiguous()
            stats = self.bn_stats and self.training and torch.is_grad_enabled()
            if self.stride == (2, 2) and _lib.subsample2x2_supported(xc):
                xs = _lib.subsample2x2(xc)                   # pick the even pixels (one streaming pass), then the stride-1 kernel
                if _lib.conv1x1_x3_supported(xs, self.weight):
                    _took(self, "own")
                    return _lib.conv1x1_x3(xs, self.weight, stats)
                _took(self, "library: shape outside the 1x1 kernel's tiles", xs)
                return F.conv2d(xs, self.weight)
            if self.stride == (1, 1) and _lib.conv1x1_x3_supported(xc, self.weight):
                _took(self, "own")
                return _lib.conv1x1_x3(xc, self.weight, stats)
            _took(self, "library: shape outside the 1x1 kernel's tiles (pixels % 32, channels % 8)", x)
            return super().forward(x)
        if x.is_cuda and x.dtype == torch.bfloat16 and self.stride in ((1, 1), (2, 2)):
            from .. import _lib
            xc = x.contiguous()
            if self.stride == (2, 2):
                # stride 2 (a stage's down-sampling shortcut) = pick the even pixels, then the stride-1 path: one streaming
                # pass each way instead of the library's NCHW <-> CNHW transposes of the whole activation around its GEMM
                if not _lib.subsample2x2_supported(xc):
                    return super().forward(x)
                xc = _lib.subsample2x2(xc)
            if _lib.conv1x1_supported(xc, self.weight):
                return _lib.conv1x1(xc, self.weight)
            if self.stride == (2, 2):
                return F.conv2d(xc, self.weight.to(xc.dtype))
        return super().forward(x)


class Conv3x3(nn.Conv2d):
    """The bottleneck's dense 3x3 convolution (padding = dilation, no bias).  bfloat16 activations on the GPU at stride 1: forward,
    input gradient (csrc/conv3x3_fwd.hip: LDS transpose reads, three column-shifted copies of the staged rows) and weight gradient
    (csrc/conv3x3_wgrad.hip) run on the matrix cores straight from the NCHW tensors -- no NHWC transposes around an implicit GEMM, no
    zero-fill / cast of a float32 workspace.  Stride 2 (the first block of stages 2 and 3): the weight gradient only (even / odd column
    planes in LDS); forward and input gradient stay the library's."""

    f32x3 = False           # float32 inputs: the own float32-precision kernels instead of the library's (batch_step_bookkeeping)
    bn_stats = False        # f32x3, training, stride 1: the BatchNorm behind it takes its statistics from this convolution's epilogue

    def __init__(self, cin, cout, stride=1, dilation=1):
        super().__init__(cin, cout, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and self.f32x3:
            if self.stride == (1, 1):
                from .. import _lib
                xc = x.contiguous()
                if _lib.conv3x3_x3_supported(xc, self.weight, self.dilation[0]):
                    stats = (self.bn_stats and self.training and torch.is_grad_enabled() and
                             _lib.conv3x3_x3_stats_supported(xc, self.weight, self.dilation[0]))
                    _took(self, "own")
                    return _lib.conv3x3_x3(xc, self.weight, self.dilation[0], stats)
            elif self.stride == (2, 2) and self.dilation == (1, 1) and self.padding == (1, 1):
                from .. import _lib
                xc = x.contiguous()
                if _lib.conv3x3s2_x3_supported(xc, self.weight):
                    _took(self, "own")
                    return _lib.conv3x3s2_x3(xc, self.weight)
            _took(self, "library: shape outside the 3x3 kernels' tiles", x)
            return super().forward(x)
        if x.is_cuda and x.dtype == torch.bfloat16 and self.stride == (1, 1):
            from .. import _lib
            xc = x.contiguous()
            if _lib.conv3x3_supported(xc, self.weight, self.dilation[0]):
                return _lib.conv3x3(xc, self.weight, self.dilation[0])
        elif (x.is_cuda and x.dtype == torch.bfloat16 and self.stride == (2, 2) and self.dilation == (1, 1) and self.padding == (1, 1)):
            from .. import _lib
            xc = x.contiguous()
            if _lib.conv3x3s2_supported(xc, self.weight):
                return _lib.conv3x3s2(xc, self.weight)
        return super().forward(x)


class StemConv7x7(nn.Conv2d):
    """The ResNet stem Conv2d(3, 64, 7, stride 2, padding 3).  Under bfloat16 autocast on the GPU the forward is the MFMA
    kernel of csrc/stem_conv.hip (straight from NCHW: the library surrounds its NHWC implicit GEMM with three layout
    transposes and a zero-fill), and so is the weight gradient."""

    f32x3 = False           # float32 inputs outside autocast: the own float32-precision kernels (batch_step_bookkeeping)

    def __init__(self):
        super().__init__(3, 64, 7, stride=2, padding=3, bias=False)

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and self.f32x3 and not torch.is_autocast_enabled('cuda')):
            from .. import _lib
            xc = x.contiguous()
            if _lib.stem_conv7x7_supported(xc, self.weight):
                _took(self, "own")
                return _lib.stem_conv7x7_x3(xc, self.weight)
            _took(self, "library: shape outside the stem kernel's tiles", x)
            return super().forward(x)
        lowp = x.dtype == torch.bfloat16 or (x.is_cuda and torch.is_autocast_enabled('cuda') and
                                             torch.get_autocast_dtype('cuda') == torch.bfloat16)
        if x.is_cuda and lowp:
            from .. import _lib
            xc = x.contiguous()                              # float32 images are rounded to bfloat16 inside the kernel
            if _lib.stem_conv7x7_supported(xc, self.weight):
                return _lib.stem_conv7x7(xc, self.weight)
        return super().forward(x)


class MaxPool3x3s2(nn.MaxPool2d):
    """MaxPool2d(3, 2, 1); on the GPU the HIP kernels (csrc/maxpool.hip: one index byte per output, gather backward)."""

    def __init__(self):
        super().__init__(3, stride=2, padding=1)

    def forward(self, x):
        if x.is_cuda:
            from .. import _lib
            xc = x.contiguous()
            if _lib.maxpool3x3s2_supported(xc):
                return _lib.maxpool3x3s2(xc)
        return super().forward(x)


class SeparableConv2d(nn.Sequential):
    """Depthwise 3x3 (dilated) followed by a pointwise convolution, no normalisation in between."""

    f32x3 = False

    def __init__(self, cin, cout, k=3, dilation=1):
        assert k == 3
        super().__init__(DepthwiseConv3x3(cin, dilation=dilation), Conv1x1(cin, cout))

    def forward(self, x):
        dw, pw = self[0], self[1]
        d = dw.dilation[0]
        if d >= x.shape[2] and d >= x.shape[3] and dw.stride == (1, 1) and dw.padding == (d, d):
            # The dilation reaches past the map (ASPP rate 36 on the 32 x 32 map of a 512 x 512 input): every tap but the centre
            # reads padding only, so the depthwise pass is a per-channel scale -- folded into the pointwise weights (a [cout, cin]
            # product) instead of streaming the whole activation through a convolution, forward and backward.  The gradient of
            # the eight unused taps is exactly zero either way.
            w = pw.weight * dw.weight[:, 0, 1, 1].view(1, -1, 1, 1)
            if x.is_cuda and x.dtype == torch.float32 and self.f32x3:
                from .. import _lib
                xc = x.contiguous()
                if _lib.conv1x1_x3_supported(xc, w):
                    return _lib.conv1x1_x3(xc, w)            # (w is a product, not a tracked parameter: split per call)
            if x.is_cuda and x.dtype == torch.bfloat16:
                from .. import _lib
                xc = x.contiguous()
                if _lib.conv1x1_supported(xc, w):
                    return _lib.conv1x1(xc, w)
            return F.conv2d(x, w.to(x.dtype))
        return pw(dw(x))


# ---------------------------------------------------------------------------------------------- ResNet-50
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = Conv1x1(cin, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = Conv3x3(planes, planes, stride, dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = Conv1x1(planes, planes * 4)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x, handles=1):
        """`x` may be a tuple (main, alias, ...) produced by the previous block with handles > 1: tensors on one storage,
        one per consumer (conv1 / residual branch or downsample), so that their gradients reach the producing BatchNorm
        kernel separately and are summed there instead of by autograd adds over the whole activation."""
        x_main, x_res = (x[0], x[1]) if isinstance(x, tuple) else (x, x)
        c2 = self._bn1_on_load(self.conv1(x_main))
        c3 = self._bn2_on_load(c2)
        if c3 is None:
            c3 = self.conv3(bn_act(self.bn2, c2, 'relu'))
        if self.downsample is not None:
            pair = self._bn3_with_shortcut(x_res, c3, handles)          # (synchronised statistics included: one all-reduce per direction)
            if pair is not None:
                return pair
            if self._pairs_shortcut(x_res):
                # synchronised statistics where the fused pair does not apply (bfloat16, shapes outside its tiles): bn3 and the projection
                # shortcut's BatchNorm as two materialised layers that share one all-reduce per direction
                from .. import _lib
                short = self.__dict__.pop('_short_cached', None)
                if short is None:
                    short = self.downsample[0](x_res)
                short = short.contiguous()
                main = c3.contiguous()
                bns = self.downsample[1].bn
                if main.shape == short.shape and _lib.bn_act_supported(main, short):
                    _bump(self.bn3)
                    _bump(bns)
                    return _lib.sync_batch_norm_shortcut_pair(
                        main, (self.bn3.weight, self.bn3.bias, self.bn3.running_mean, self.bn3.running_var, self.bn3.momentum, self.bn3.eps),
                        short, (bns.weight, bns.bias, bns.running_mean, bns.running_var, bns.momentum, bns.eps), _ACT_CODE['relu'], handles)
                return bn_act(self.bn3, main, 'relu', residual=bn_act(bns, short, None), handles=handles)
        idt, raff = (x_res, None) if self.downsample is None else self._shortcut_on_load(x_res, c3)
        if raff is None:
            return bn_act(self.bn3, c3, 'relu', residual=idt, handles=handles)
        return bn_act(self.bn3, c3, 'relu', residual=idt, handles=handles, res_affine=raff)

    pair_shortcut_bn = True  # f32x3 training: bn3 and the projection shortcut's BatchNorm as ONE forward pass and ONE two-pass backward

    def _bn3_with_shortcut(self, x_res, c3, handles):
        """relu(bn3(c3) + bn_s(conv_s(x))) through _lib.batch_norm_act_res_bn (the shortcut's BatchNorm has no pass of its own in either
        direction), or None where that does not apply (then _shortcut_on_load / the materialised path)."""
        d = self.downsample
        if not (self.pair_shortcut_bn and self.lazy_shortcut and isinstance(d, nn.Sequential) and len(d) == 2 and type(d[1]) is BNAct and
                d[1].act is None and isinstance(d[0], Conv1x1) and getattr(c3, '_aadg_bn_sums', None) is not None and self.training and
                torch.is_grad_enabled() and d[0].f32x3 and d[0].bn_stats and c3.dtype == torch.float32 and
                type(self.bn3) is nn.BatchNorm2d and self.bn3.affine and self.bn3.momentum is not None and self.bn3.track_running_stats):
            return None
        bns = d[1].bn
        if not (type(bns) is nn.BatchNorm2d and bns.momentum is not None and bns.track_running_stats and bns.affine):
            return None
        from .. import _lib
        xs = x_res.contiguous()
        cs = d[0]
        # (the shortcut convolution's own conditions, so that a shape it cannot take falls back BEFORE anything is computed twice)
        probe = _lib.subsample2x2_supported(xs) if cs.stride == (2, 2) else cs.stride == (1, 1)
        if not probe:
            return None
        short = cs(x_res)
        if getattr(short, '_aadg_bn_sums', None) is None or short.shape != c3.shape or not _lib.bn_act_supported(c3.contiguous(), short.contiguous()):
            self._short_cached = short
            return None
        _bump(self.bn3)
        _bump(bns)
        b3 = self.bn3
        return _lib.batch_norm_act_res_bn(c3, (b3.weight, b3.bias, b3.running_mean, b3.running_var, b3.momentum, b3.eps),
                                          short, (bns.weight, bns.bias, bns.running_mean, bns.running_var, bns.momentum, bns.eps),
                                          _ACT_CODE['relu'], handles, sync=_BN_SYNC)

    lazy_shortcut = True    # f32x3 training: the projection shortcut's BatchNorm applied while bn3's kernel reads the residual

    def _shortcut_on_load(self, x_res, c3):
        """(residual, res_affine) of the projection shortcut: the RAW output of its convolution + (scale, shift) of its BatchNorm, which
        bn3's kernel applies while it reads the residual (no elementwise pass, no normalised shortcut tensor) -- or (downsample(x), None)."""
        d = self.downsample
        if (self.lazy_shortcut and isinstance(d, nn.Sequential) and len(d) == 2 and type(d[1]) is BNAct and d[1].act is None and
                isinstance(d[0], Conv1x1) and getattr(c3, '_aadg_bn_sums', None) is not None and self.training and torch.is_grad_enabled() and
                not _BN_SYNC and d[0].f32x3 and c3.dtype == torch.float32 and type(self.bn3) is nn.BatchNorm2d):
            bns = d[1].bn
            short = self.__dict__.pop('_short_cached', None)
            if short is None:
                short = d[0](x_res)
            if (getattr(short, '_aadg_bn_sums', None) is not None and short.shape == c3.shape and type(bns) is nn.BatchNorm2d and
                    bns.momentum is not None and bns.track_running_stats and bns.affine):
                from .. import _lib
                sc = short.contiguous()
                if _lib.bn_act_supported(c3.contiguous(), sc):
                    _bump(bns)
                    z, scale, shift = _lib.batch_norm_lazy(sc, bns.weight, bns.bias, bns.running_mean, bns.running_var, bns.momentum, bns.eps,
                                                           short._aadg_bn_sums, act=_ACT_CODE[None])
                    return z, (scale, shift)
            return d[1](short), None
        short = self.__dict__.pop('_short_cached', None)           # (_bn3_with_shortcut ran the convolution and then declined)
        return (d(x_res) if short is None else d[1](short)), None

    lazy_bn1 = True         # f32x3 training: bn1 + ReLU applied by conv2 (stride 1) while it stages its operand, as lazy_bn2 below
    lazy_bn2 = True         # f32x3 training: bn2 + ReLU applied by conv3 while it loads its operand (no elementwise pass, no normalised tensor)

    def _lazy_ok(self, bn, x, conv):
        # (_BN_SYNC: the lazy Function all-reduces the totals itself, batch_norm_lazy(sync=True))
        return (self.training and torch.is_grad_enabled() and conv.f32x3 and x.is_cuda and x.dtype == torch.float32 and
                conv.stride == (1, 1) and getattr(x, '_aadg_bn_sums', None) is not None and type(bn) is nn.BatchNorm2d and
                bn.momentum is not None and bn.track_running_stats and bn.affine)

    def _bn1_on_load(self, c1):
        """conv2(relu(bn1(c1))): bn1 + ReLU on conv2's operand load where that applies (see _bn2_on_load), else the materialised path"""
        bn, conv2 = self.bn1, self.conv2
        if self.lazy_bn1 and conv2.bn_stats and self._lazy_ok(bn, c1, conv2):
            from .. import _lib
            c1c = c1.contiguous()
            d = conv2.dilation[0]
            if _lib.conv3x3_x3_pre_supported(c1c, conv2.weight, d) and _lib.bn_act_supported(c1c, None):
                _bump(bn)
                z, scale, shift = _lib.batch_norm_lazy(c1c, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                                       _take_bn_sums(c1), sync=_BN_SYNC)
                return _lib.conv3x3_x3(z, conv2.weight, d, True, pre=(scale, shift))
        return self.conv2(bn_act(bn, c1, 'relu'))

    def _bn2_on_load(self, c2):
        """conv3(relu(bn2(c2))) with the normalisation applied on conv3's operand load (_lib.batch_norm_lazy + conv1x1_x3(pre=...)), or None
        where that path does not apply: needs the statistics conv2 left in its epilogue, plain per-device BatchNorm, whole-tile shapes."""
        bn, conv3 = self.bn2, self.conv3
        if not (self.lazy_bn2 and self._lazy_ok(bn, c2, conv3)):
            return None
        from .. import _lib
        c2c = c2.contiguous()
        if not (_lib.conv1x1_x3_pre_supported(c2c, conv3.weight) and _lib.bn_act_supported(c2c, None)):
            return None
        sums = _take_bn_sums(c2)
        _bump(bn)
        z, scale, shift = _lib.batch_norm_lazy(c2c, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, sums, sync=_BN_SYNC)
        return _lib.conv1x1_x3(z, conv3.weight, conv3.bn_stats, pre=(scale, shift))

    def _pairs_shortcut(self, x):
        d = self.downsample
        if not (_BN_SYNC and self.training and torch.is_grad_enabled() and x.is_cuda and isinstance(d, nn.Sequential) and len(d) == 2 and
                type(d[1]) is BNAct and d[1].act is None):
            return False
        return all(type(bn) is nn.BatchNorm2d and bn.momentum is not None and bn.track_running_stats for bn in (self.bn3, d[1].bn))


class _Stage(nn.Sequential):
    """Bottlenecks in sequence.  A block output consumed by the next block is handed over as a (main, alias) pair; the
    stage output gets `out_handles` handles (one per consumer outside the stage)."""

    def forward(self, x, out_handles=1):
        last = len(self) - 1
        for i, blk in enumerate(self):
            x = blk(x, handles=2 if i < last else out_handles)
        return x


def _handles(x, k):
    """k per-consumer handles of a stage output (the tensor itself k times when it came as a plain tensor)."""
    if isinstance(x, tuple):
        assert len(x) >= k
        return list(x[:k])
    return [x] * k


class ResNet50Encoder(nn.Module):
    """Stages at strides 2, 4, 8, 16, 16 (layer4 dilated by 2 for output stride 16)."""
    out_channels = 2048
    skip_channels = 256

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(StemConv7x7(), *_bn_relu(64))
        self.pool = MaxPool3x3s2()
        self.cin = 64
        self.layer1 = self._stage(64, 3, 1, 1)
        self.layer2 = self._stage(128, 4, 2, 1)
        self.layer3 = self._stage(256, 6, 2, 1)
        self.layer4 = self._stage(512, 3, 1, 2)

    def _stage(self, planes, blocks, stride, dilation):
        down = None
        if stride != 1 or self.cin != planes * 4:
            down = nn.Sequential(Conv1x1(self.cin, planes * 4, stride), BNAct(planes * 4, None))
        layers = [Bottleneck(self.cin, planes, stride, dilation, down)]
        self.cin = planes * 4
        layers += [Bottleneck(self.cin, planes, 1, dilation) for _ in range(1, blocks)]
        return _Stage(*layers)

    deep_handles = 5     # consumers of the encoder output: four ASPP convolutions + the global average (image pool & pooled feature)

    def _stem_pool(self, x):
        """pool(relu(bn(conv(x)))); in training on the GPU BatchNorm + ReLU + pooling are one pass over the convolution's output
        (the normalised 64-channel half-resolution map, the largest activation of the network, is never written)."""
        conv, bna = self.stem[0], self.stem[1]
        bn = bna.bn
        y = conv(x)
        if (y.is_cuda and bn.training and torch.is_grad_enabled() and type(bn) is nn.BatchNorm2d and bna.act == 'relu' and not _BN_SYNC and
                bn.momentum is not None and bn.track_running_stats and type(self.pool) is MaxPool3x3s2):
            from .. import _lib
            yc = y.contiguous()
            if _lib.bn_relu_maxpool_supported(yc):
                _bump(bn)
                return _lib.bn_relu_maxpool(yc, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        return self.pool(bna(y))

    def forward(self, x):
        x = self._stem_pool(x)
        s = _handles(self.layer1(x, out_handles=3), 3)            # layer2's conv1, layer2's downsample, the decoder skip
        x = self.layer2((s[0], s[1]), out_handles=2)
        x = self.layer3(x, out_handles=2)
        x = self.layer4(x, out_handles=self.deep_handles)
        return s[2], x


# ---------------------------------------------------------------------------------------------- MobileNetV2
class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, expand, dilation=1):
        super().__init__()
        hid = cin * expand
        self.use_res = stride == 1 and cin == cout
        layers = []
        if expand != 1:
            layers += [Conv1x1(cin, hid), BNAct(hid, 'relu6')]
        layers += [DepthwiseConv3x3(hid, stride, dilation), BNAct(hid, 'relu6'),
                   Conv1x1(hid, cout), BNAct(cout, None)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res else self.conv(x)


class MobileNetV2Encoder(nn.Module):
    out_channels = 1280
    skip_channels = 24

    def __init__(self):
        super().__init__()
        cfg = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
        feats = [nn.Sequential(nn.Conv2d(3, 32, 3, 2, 1, bias=False), BNAct(32, 'relu6'))]
        cin, stride_so_far, dilation = 32, 2, 1
        for t, c, n, s in cfg:
            for i in range(n):
                st = s if i == 0 else 1
                d = dilation
                if st == 2 and stride_so_far >= 16:      # keep output stride 16: dilate instead of striding
                    st, dilation = 1, dilation * 2
                if st == 2:
                    stride_so_far *= 2
                feats.append(InvertedResidual(cin, c, st, t, d if st == 1 else 1))
                cin = c
        feats.append(nn.Sequential(Conv1x1(cin, 1280), BNAct(1280, 'relu6')))
        self.features = nn.Sequential(*feats)

    def forward(self, x):
        skip = None
        for i, f in enumerate(self.features):
            x = f(x)
            if i == 3:
                skip = x                                   # stride 4, 24 channels
        return skip, x


# ---------------------------------------------------------------------------------------------- head
class _GlobalAvgPoolF32(torch.autograd.Function):
    """x.mean(dim=(2, 3), dtype=float32).  autograd's own backward builds the full-size gradient in float32 and casts it
    afterwards (1.2 GB written + read again for the encoder output at N = 144); here it is a broadcast view in x's dtype."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        return x.mean(dim=(2, 3), dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.shape
        return (g * (1.0 / (H * W))).to(ctx.dtype)[:, :, None, None].expand(N, C, H, W)


def global_avg_pool_f32(x):
    return _GlobalAvgPoolF32.apply(x)


# activations whose concatenations are produced in place (the BatchNorm kernels write their channel slice of the buffer, their backward
# reads its slice of the gradient): bfloat16 since round 2; float32 -- the f32x3 default -- since the end of round 6 (it still took torch.cat:
# 0.47 ms for the ASPP head's copy plus the slice copies of its backward)
_INPLACE_CONCAT_DTYPES = (torch.bfloat16, torch.float32)


class ASPP(nn.Module):
    def __init__(self, cin, cout=256, rates=(12, 24, 36)):
        super().__init__()
        self.branches = nn.ModuleList(
            [nn.Sequential(Conv1x1(cin, cout), *_bn_relu(cout))] +
            [nn.Sequential(SeparableConv2d(cin, cout, 3, r), *_bn_relu(cout)) for r in rates])
        self.image_pool = nn.Sequential(nn.AdaptiveAvgPool2d(1), Conv1x1(cin, cout), *_bn_relu(cout))
        self.project = nn.Sequential(Conv1x1(cout * 5, cout), *_bn_relu(cout), nn.Dropout(0.5))

    def forward(self, x, pooled=None):
        """`pooled` = the float32 global average of x when the caller has it anyway (the classification feature of the
        discriminator): the image-pool branch then starts from it instead of reducing the encoder output a second time."""
        nb = len(self.branches)
        h = _handles(x, nb if pooled is not None else nb + 1)      # one handle of the encoder output per consumer
        ref = h[0]
        fused, convs = self._forward_into_concat(h, pooled) if (pooled is not None and self.training and torch.is_grad_enabled() and
                                                                ref.is_cuda and ref.dtype in _INPLACE_CONCAT_DTYPES) else (None, None)
        if fused is not None:
            return self.project(fused)
        if convs is not None:
            # the in-place path ran the branch convolutions, found an unsupported layout and touched no BatchNorm: go on from
            # the convolution outputs (running statistics are updated exactly once per step either way)
            outs = [b[1](y) for b, y in zip(self.branches, convs)]
        else:
            outs = [b(hx) for b, hx in zip(self.branches, h)]
        if pooled is None:
            ip = self.image_pool(h[-1])
        else:
            ip = pooled.to(ref.dtype)[:, :, None, None]
            for mod in list(self.image_pool)[1:]:                  # [0] is the pooling itself
                ip = mod(ip)
        # bilinear up-sampling of a 1x1 map is a broadcast (ATen's kernel would walk all N*C planes in one workgroup)
        outs.append(ip.expand(-1, -1, ref.shape[-2], ref.shape[-1]))
        return self.project(torch.cat(outs, dim=1))


    def _forward_into_concat(self, h, pooled):
        """The five branch outputs written straight into their channel slices of the concatenation buffer by the BatchNorm
        kernels (no torch.cat copy; the gradient slices are read in place by the BatchNorm backward).  None when a branch is
        not the plain (convolution, BNAct) pair on a supported layout."""
        from .. import _lib
        ref = h[0]
        N, _, H, W = ref.shape
        cout = self.project[0].in_channels // (len(self.branches) + 1)
        for b in self.branches:
            if len(b) != 2 or type(b[1]) is not BNAct or type(b[1].bn) is not nn.BatchNorm2d or b[1].bn.num_features != cout:
                return None, None
        if (H * W) % 8 != 0:
            return None, None
        # every precondition is checked BEFORE the first BatchNorm runs (a BatchNorm call updates running statistics): first all
        # the convolutions (stateless), then the checks on their outputs, then the BatchNorm kernels
        convs = [b[0](hx) for b, hx in zip(self.branches, h)]
        if not all(y.is_cuda and y.dtype == ref.dtype and _lib.bn_act_supported(y.contiguous()) for y in convs):
            return None, convs
        buf, parts = _lib.concat_slices(N, [cout] * (len(self.branches) + 1), H, W, ref.dtype, ref.device)
        ip = pooled.to(ref.dtype)[:, :, None, None]
        pool_mods = list(self.image_pool)[1:]                      # [0] is the pooling itself
        if (_BN_SYNC and len(pool_mods) == 2 and type(pool_mods[1]) is BNAct and type(pool_mods[1].bn) is nn.BatchNorm2d and
                all(bn.momentum is not None and bn.track_running_stats for bn in [b[1].bn for b in self.branches] + [pool_mods[1].bn])):
            # synchronised statistics: the five BatchNorm layers of the head are independent of one another -- ONE all-reduce per
            # direction for all of them (sync_batch_norm_act_group) instead of five
            ipc = pool_mods[0](ip).contiguous()
            bns = [b[1] for b in self.branches] + [pool_mods[1]]
            xs = [y.contiguous() for y in convs] + [ipc]
            for m in bns:
                _bump(m.bn)
            res = _lib.sync_batch_norm_act_group([(x, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var, m.bn.momentum, m.bn.eps,
                                                   _ACT_CODE[m.act], part)
                                                  for x, m, part in zip(xs, bns, list(parts[:-1]) + [None])])
            outs = list(res[:-1])
            ip = res[-1]
        else:
            outs = []
            for b, y, part in zip(self.branches, convs, parts):
                o = bn_act(b[1].bn, y, b[1].act, out=part)
                outs.append(o if o.data_ptr() == part.data_ptr() else part.copy_(o))     # (a BatchNorm variant without the kernels)
            for mod in pool_mods:
                ip = mod(ip)
        outs.append(parts[-1].copy_(ip.expand(-1, -1, H, W)))      # bilinear up-sampling of a 1x1 map = broadcast
        return _lib.concat_from_slices(buf, outs), convs


class DeepLabV3Plus(nn.Module):
    def __init__(self, encoder_name='mobilenet_v2', classes=2, aux_pooling=True):
        super().__init__()
        if encoder_name == 'mobilenet_v2':
            self.encoder = MobileNetV2Encoder()
        elif encoder_name in ('resnet50', 'resnet-50'):
            self.encoder = ResNet50Encoder()
        else:
            raise NotImplementedError(encoder_name + ' has not been implemented!')
        self.feature_channels = self.encoder.out_channels
        self.aspp = nn.Sequential(ASPP(self.encoder.out_channels), SeparableConv2d(256, 256, 3), *_bn_relu(256))
        self.skip = nn.Sequential(Conv1x1(self.encoder.skip_channels, 48), *_bn_relu(48))
        self.fuse = nn.Sequential(SeparableConv2d(256 + 48, 256, 3), *_bn_relu(256))
        self.classifier = nn.Conv2d(256, classes, 1)
        self.aux_pooling = aux_pooling

    def forward(self, x):
        skip, deep = self.encoder(x)
        hd = _handles(deep, 5)
        # ClassificationHead = avg-pool + flatten (models/heads.py:19-25): accumulated in fp32 without materialising an fp32
        # copy of the [N, C_enc, h/16, w/16] map; the same average feeds the ASPP image-pool branch (one reduction, one
        # broadcast gradient instead of two)
        pooled = global_avg_pool_f32(hd[4])
        a = self.aspp[0](tuple(hd[:4]), pooled)          # ASPP takes one handle of the encoder output per convolution branch
        for mod in list(self.aspp)[1:]:
            a = mod(a)
        y = None
        if (a.is_cuda and self.training and torch.is_grad_enabled() and a.dtype in _INPLACE_CONCAT_DTYPES and len(self.skip) == 2 and
                type(self.skip[1]) is BNAct and type(self.skip[1].bn) is nn.BatchNorm2d):
            # both halves of the decoder concatenation are produced in place: the skip branch's BatchNorm writes its slice of
            # the buffer, the up-sampling kernel the other
            from .. import _lib
            t = self.skip[0](skip)
            if t.dtype == a.dtype and _lib.bn_act_supported(t.contiguous()) and (t.shape[2] * t.shape[3]) % 8 == 0:
                buf, parts = _lib.concat_slices(a.shape[0], [a.shape[1], t.shape[1]], t.shape[2], t.shape[3], a.dtype, a.device)
                s = bn_act(self.skip[1].bn, t, self.skip[1].act, out=parts[1])
                if s.data_ptr() != parts[1].data_ptr():
                    s = parts[1].copy_(s)
                y = _lib.upsample_cat(a, s, buf)
            else:
                s = self.skip[1](t)
        else:
            s = self.skip(skip)
        if y is not None:
            pass
        elif a.is_cuda and a.dtype == s.dtype and a.dtype in (torch.float32, torch.bfloat16):
            from .. import _lib
            y = _lib.upsample_cat(a, s)                     # up-sampling written straight into the concatenation
        else:
            y = torch.cat([_upsample_ac(a, skip.shape[-2:]), s], dim=1)
        logits = self._fuse_classify_on_load(y)
        if logits is None:
            logits = self._classify(self.fuse(y))
        mask = _upsample_ac(logits, x.shape[-2:])
        if not self.aux_pooling:
            return mask
        return mask, pooled


def _fuse_classify_on_load(self, y):
    """classifier(relu(bn(fuse_conv(y)))) with the decoder's last BatchNorm + ReLU applied while the 1x1 classifier (weight padded to 8
    rows, see _classify) loads the fuse convolution's output -- the normalised [N, 256, H/4, W/4] tensor, the largest of the decoder, is
    never written (Bottleneck._bn2_on_load has the conditions); None where that path does not apply."""
    c, fuse = self.classifier, self.fuse
    if not (self.lazy_fuse_bn and getattr(self, 'f32x3', False) and self.training and torch.is_grad_enabled() and
            y.is_cuda and y.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and c.kernel_size == (1, 1) and
            c.out_channels <= 8 and len(fuse) == 2 and isinstance(fuse[0], SeparableConv2d) and type(fuse[1]) is BNAct and
            fuse[1].act == 'relu'):
        return None
    bn = fuse[1].bn
    if not (type(bn) is nn.BatchNorm2d and bn.momentum is not None and bn.track_running_stats and bn.affine):
        return None
    from .. import _lib
    f = fuse[0](y)
    sums = getattr(f, '_aadg_bn_sums', None)
    w8 = F.pad(c.weight, (0, 0, 0, 0, 0, 0, 0, 8 - c.out_channels))
    fc = f.contiguous()
    if sums is None or not (_lib.conv1x1_x3_pre_supported(fc, w8) and _lib.bn_act_supported(fc, None)):
        return self._classify(fuse[1](f))
    _bump(bn)
    sums = _take_bn_sums(f)
    z, scale, shift = _lib.batch_norm_lazy(fc, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, sums, sync=_BN_SYNC)
    out = _lib.conv1x1_x3(z, w8, pre=(scale, shift))[:, :c.out_channels]
    return out + c.bias.view(1, -1, 1, 1) if c.bias is not None else out.contiguous()


def _classify(self, y):
    """The 1x1 classifier (256 -> classes, with bias).  f32x3: the own float32-precision 1x1 kernels with the weight padded to 8 output
    rows (their tiles want multiples of 8; the library ran this 2-channel convolution as a Winograd variant + an NHWC weight gradient
    behind a transpose of the 2.4 GB input: ~3 ms per step for 2.4 GFLOP); the padded rows are zero and sliced away again."""
    c = self.classifier
    if (getattr(self, 'f32x3', False) and y.is_cuda and y.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and
            c.kernel_size == (1, 1) and c.out_channels <= 8):
        from .. import _lib
        w8 = F.pad(c.weight, (0, 0, 0, 0, 0, 0, 0, 8 - c.out_channels))
        yc = y.contiguous()
        if _lib.conv1x1_x3_supported(yc, w8):
            out = _lib.conv1x1_x3(yc, w8)[:, :c.out_channels]
            return out + c.bias.view(1, -1, 1, 1) if c.bias is not None else out.contiguous()
    return c(y)


DeepLabV3Plus._classify = _classify
DeepLabV3Plus._fuse_classify_on_load = _fuse_classify_on_load
DeepLabV3Plus.lazy_fuse_bn = True     # f32x3 training: the decoder's last BatchNorm + ReLU on the classifier's operand load


class UNetSmall(nn.Module):
    """Small UNet returning (logits, pooled bottleneck) -- BASELINE config 0 (CPU plumbing case)."""

    def __init__(self, classes=1, width=16):
        super().__init__()
        def block(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), *_bn_relu(cout),
                                 nn.Conv2d(cout, cout, 3, padding=1, bias=False), *_bn_relu(cout))
        w = width
        self.d1, self.d2, self.d3 = block(3, w), block(w, 2 * w), block(2 * w, 4 * w)
        self.mid = block(4 * w, 8 * w)
        self.u3, self.u2, self.u1 = block(12 * w, 4 * w), block(6 * w, 2 * w), block(3 * w, w)
        self.out = nn.Conv2d(w, classes, 1)
        self.feature_channels = 8 * w

    def forward(self, x):
        a = self.d1(x)
        b = self.d2(F.max_pool2d(a, 2))
        c = self.d3(F.max_pool2d(b, 2))
        m = self.mid(F.max_pool2d(c, 2))
        up = lambda t, ref: F.interpolate(t, size=ref.shape[-2:], mode='bilinear', align_corners=False)  # noqa: E731
        y = self.u3(torch.cat([up(m, c), c], 1))
        y = self.u2(torch.cat([up(y, b), b], 1))
        y = self.u1(torch.cat([up(y, a), a], 1))
        return self.out(y), m.mean(dim=(2, 3), dtype=torch.float32)
