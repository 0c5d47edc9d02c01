"""SegFormer (MiT-B2 encoder + all-MLP head) with the `(mask_logits, pooled_encoder_feature)` contract of the inner loop
(BASELINE configs[4]: SegFormer-B2, 8 source domains, bf16).

What the reference holds for this model is architecture only: hyper-parameters of `mit_b2`
(models/mmseg/models/backbones/mix_transformer.py:392-399: embed dims 64/128/320/512, heads 1/2/5/8, depths 3/4/6/3,
spatial-reduction ratios 8/4/2/1, MLP ratio 4, qkv bias, LayerNorm eps 1e-6, drop-path 0.1), the block structure
(:19-159: pre-norm attention with a strided-conv reduction of K/V, Mix-FFN = fc1 -> depthwise 3x3 -> GELU -> fc2), overlapping
patch embeddings (:162-202: 7x7/4 then 3x3/2), and the head (models/mmseg/models/decode_heads/segformer_head.py:35-87: one Linear
per stage to embed_dim 768, bilinear resize to the stride-4 grid, concat, 1x1 conv + BN + ReLU, dropout 0.1, 1x1 classifier)
followed by x4 bilinear up-sampling (models/segformer.py:75).  Its wrapper returns only `pred` and needs mmcv/timm
(models/segformer.py:4-6,77-82), neither of which is in this image: the network is written here in plain torch.nn with parameter
names that follow the reference modules (`backbone.block2.1.attn.kv.weight`, `head.linear_c3.proj.weight`, ...) so that a
SegFormer checkpoint's state_dict maps onto it.

MI355X-first choices (same function, different schedule):
  * tokens stay [B, N, C] and bfloat16 through a stage (the Linear layers are plain hipBLASLt GEMMs on them); every residual add
    is folded into the LayerNorm that follows it (csrc/layernorm.hip: add + stochastic-depth factor + normalise in one pass, in the
    tokens' own dtype -- no float32 residual stream, no casts around the norms); attention is
    `scaled_dot_product_attention`; the Mix-FFN's depthwise convolution reads the same buffer through a channels-last view;
  * the head never builds the 4 x 768-channel concatenation at stride 4 (14.5 GB in bf16 for 144 images of 512x512):
    a 1x1 convolution commutes with bilinear interpolation, so the fuse convolution's slice for stage i is folded into that
    stage's Linear (one [768, C_i] matrix, rebuilt from the reference-layout parameters each step), applied at the stage's own
    resolution, and the four results are interpolated to the stride-4 grid and summed.  Exact in real arithmetic (interpolation
    weights sum to one, so the biases pass through), 4x less traffic and no 3072-deep GEMM at full resolution;
  * BatchNorm + ReLU of the fuse layer and the final x4 up-sampling are the HIP kernels of the DeepLab path.
Weights are randomly initialised as mix_transformer.py:33-46 prescribes (no network for ImageNet weights).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .deeplab import _upsample_ac, bn_act, global_avg_pool_f32

MIT_B2 = dict(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(4, 4, 4, 4), depths=(3, 4, 6, 3),
              sr_ratios=(8, 4, 2, 1), drop_path_rate=0.1, head_dim=768)


def _init(m):
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            nn.init.zeros_(m.bias)


class DropPath(nn.Module):
    """Stochastic depth per sample (the residual branch of a block is dropped with probability p and rescaled)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.drop_prob = float(p)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


def _dp_scale(dp, x):
    """Stochastic depth as a per-sample factor of the branch (mask / keep, float32 [B]); None when nothing is dropped."""
    if isinstance(dp, DropPath) and dp.drop_prob > 0.0 and dp.training:
        keep = 1.0 - dp.drop_prob
        return torch.empty(x.shape[0], dtype=torch.float32, device=x.device).bernoulli_(keep) / keep
    return None


def _ln(norm, x, r=None, rscale=None):
    """(s, LayerNorm(s)) with s = x + rscale[sample] * r (r None: s = x).  On the GPU one HIP pass in the tokens' own dtype
    (csrc/layernorm.hip: the pending branch joins the residual stream inside the normalisation that follows it -- under autocast
    torch would add in float32, run layer_norm in float32 and cast the result back for the next Linear)."""
    if x.is_cuda:
        from .. import _lib
        xc = x.contiguous()
        if _lib.layernorm_supported(xc, xc.shape[-1]) and (r is None or r.dtype == xc.dtype):
            out = _lib.add_layer_norm(xc, None if r is None else r.contiguous(), rscale, norm.weight, norm.bias, norm.eps)
            return (xc, out) if r is None else out
    if r is not None:
        x = x + (r if rscale is None else r * rscale.to(r.dtype).view(-1, *([1] * (r.dim() - 1))))
    return x, norm(x)


def _tokens_as_map(x, H, W):
    """[B, H*W, C] tokens viewed as a [B, C, H, W] tensor in channels-last memory format: no copy."""
    B, N, C = x.shape
    return x.view(B, H, W, C).permute(0, 3, 1, 2)


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W):
        m = _tokens_as_map(x, H, W)
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
            # the library's channels-last depthwise paths are pathological here (naive double-precision kernels forward, a
            # 30 ms grouped-convolution weight gradient per block): go through NCHW and the LDS-tiled kernels of
            # csrc/depthwise.hip (float32 master weights, own weight gradient) -- two transposing copies around them
            from .. import _lib
            mc = m.contiguous()
            if _lib.dwconv3x3_supported(mc, self.dwconv.weight, 1):
                y = _lib.dwconv3x3(mc, self.dwconv.weight, 1) + self.dwconv.bias.to(mc.dtype)[None, :, None, None]
                return y.flatten(2).transpose(1, 2)
        y = self.dwconv(m)                                            # channels-last in, channels-last out
        return y.permute(0, 2, 3, 1).reshape(x.shape)


class Mlp(nn.Module):
    """Mix-FFN: fc1 -> depthwise 3x3 (positional information leaks in through the zero padding) -> GELU -> fc2."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = DWConv(hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, H, W):
        h = self.fc1(x)
        if h.is_cuda and type(self.act) is nn.GELU and getattr(self.act, 'approximate', 'none') == 'none':
            from .. import _lib
            hc = h.contiguous()
            if _lib.dwconv3x3_gelu_nhwc_supported(hc, H, W):
                # depthwise 3x3 + bias + GELU in the tokens' own layout, one pass (csrc/dwconv_nhwc.hip): no NCHW round trip
                return self.fc2(_lib.dwconv3x3_gelu_nhwc(hc, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, H, W))
        return self.fc2(self.act(self.dwconv(h, H, W)))


class Attention(nn.Module):
    """Multi-head self-attention whose keys / values come from a sr_ratio-times down-sampled copy of the tokens
    (strided convolution + LayerNorm), so the score matrix is N x N / sr^2."""

    def __init__(self, dim, num_heads, sr_ratio):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads, self.sr_ratio = num_heads, sr_ratio
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.kv = nn.Linear(dim, dim * 2, bias=True)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim, eps=1e-5)                   # plain nn.LayerNorm in the reference (:75)

    def forward(self, x, H, W):
        B, N, C = x.shape
        h, d = self.num_heads, C // self.num_heads
        q = self.q(x).view(B, N, h, d).transpose(1, 2)                                   # [B, h, N, d]
        if self.sr_ratio > 1:
            r = self.sr(_tokens_as_map(x, H, W))                                         # [B, C, H/sr, W/sr]
            r = _ln(self.norm, r.flatten(2).transpose(1, 2))[1]
        else:
            r = x
        kv = self.kv(r).view(B, -1, 2, h, d).permute(2, 0, 3, 1, 4)                      # [2, B, h, N', d]
        o = F.scaled_dot_product_attention(q, kv[0], kv[1], scale=self.scale)            # softmax(q k^T * scale) v
        return self.proj(o.transpose(1, 2).reshape(B, N, C))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, drop_path, sr_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, H, W, pend=None, pscale=None):
        """x: the residual stream, pend (* pscale per sample): the previous block's MLP branch, not yet added.  Returns the stream
        after this block's attention branch, this block's MLP branch and its stochastic-depth factor: every residual add happens
        inside the LayerNorm kernel that consumes its result (the stage's final norm takes the last one)."""
        s, n1 = _ln(self.norm1, x, pend, pscale)
        a = self.attn(n1, H, W)
        s, n2 = _ln(self.norm2, s, a, _dp_scale(self.drop_path, a))
        m = self.mlp(n2, H, W)
        return s, m, _dp_scale(self.drop_path, m)


class OverlapPatchEmbed(nn.Module):
    def __init__(self, patch, stride, cin, dim):
        super().__init__()
        self.proj = nn.Conv2d(cin, dim, kernel_size=patch, stride=stride, padding=patch // 2)
        self.norm = nn.LayerNorm(dim, eps=1e-5)                       # plain nn.LayerNorm in the reference (:176)

    def forward(self, x):
        x = self.proj(x)
        H, W = x.shape[-2:]
        return _ln(self.norm, x.flatten(2).transpose(1, 2))[1], H, W


class MixVisionTransformer(nn.Module):
    """Four stages at strides 4 / 8 / 16 / 32; returns the four maps as [B, C_i, H_i, W_i]."""

    def __init__(self, embed_dims, num_heads, mlp_ratios, depths, sr_ratios, drop_path_rate, in_chans=3, **_):
        super().__init__()
        self.depths = depths
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        cur, cin = 0, in_chans
        for i in range(4):
            setattr(self, 'patch_embed%d' % (i + 1), OverlapPatchEmbed(7 if i == 0 else 3, 4 if i == 0 else 2, cin, embed_dims[i]))
            setattr(self, 'block%d' % (i + 1), nn.ModuleList(
                [Block(embed_dims[i], num_heads[i], mlp_ratios[i], dpr[cur + k], sr_ratios[i]) for k in range(depths[i])]))
            setattr(self, 'norm%d' % (i + 1), nn.LayerNorm(embed_dims[i], eps=1e-6))
            cur += depths[i]
            cin = embed_dims[i]
        self.apply(_init)

    def forward(self, x):
        outs = []
        for i in range(1, 5):
            t, H, W = getattr(self, 'patch_embed%d' % i)(x)
            pend = ps = None
            for blk in getattr(self, 'block%d' % i):
                t, pend, ps = blk(t, H, W, pend, ps)
            t = _ln(getattr(self, 'norm%d' % i), t, pend, ps)[1]
            x = _tokens_as_map(t, H, W).contiguous()                  # NCHW stage output (small: C_i at stride 4 ... 32)
            outs.append(x)
        return outs


class _Proj(nn.Module):
    """`MLP` of the reference head (segformer_head.py:19-31): one Linear on the flattened map."""

    def __init__(self, cin, dim):
        super().__init__()
        self.proj = nn.Linear(cin, dim)


class _Fuse(nn.Module):
    """mmcv ConvModule(4 * dim -> dim, 1x1, BN, ReLU): `conv` has no bias because a norm follows."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(4 * dim, dim, 1, bias=False)
        self.bn = nn.BatchNorm2d(dim)


class SegFormerHead(nn.Module):
    def __init__(self, in_channels, dim, num_classes, dropout=0.1):
        super().__init__()
        c1, c2, c3, c4 = in_channels
        self.dim = dim
        self.linear_c4, self.linear_c3, self.linear_c2, self.linear_c1 = _Proj(c4, dim), _Proj(c3, dim), _Proj(c2, dim), _Proj(c1, dim)
        self.linear_fuse = _Fuse(dim)
        self.dropout = nn.Dropout2d(dropout)
        self.linear_pred = nn.Conv2d(dim, num_classes, 1)
        self.apply(_init)
        # the classifier is not the backbone's business: the fan-out rule of _init (mix_transformer.py's _init_weights) gives a K-channel
        # 1x1 convolution weights of std sqrt(2 / K) = 1 -- logits of +-20 at initialisation, BCE ~ 45 and a bf16 / float32 comparison
        # that measures saturation, not rounding (round 4, tests/test_gpu_precision.py).  mmseg initialises a head's classifier with
        # normal(0, 0.01) and a zero bias (decode_head.py:135-137)
        nn.init.normal_(self.linear_pred.weight, 0.0, 0.01)
        nn.init.zeros_(self.linear_pred.bias)

    def forward(self, feats):
        c1 = feats[0]
        size = c1.shape[-2:]
        wf = self.linear_fuse.conv.weight.flatten(1)                   # [dim, 4 * dim], concat order c4, c3, c2, c1
        ys = []
        for k, (lin, f) in enumerate(zip((self.linear_c4, self.linear_c3, self.linear_c2, self.linear_c1),
                                         (feats[3], feats[2], feats[1], feats[0]))):
            wk = wf[:, k * self.dim:(k + 1) * self.dim]
            w = (wk @ lin.proj.weight).to(f.dtype)                     # fuse slice folded into the stage's Linear: [dim, C_k]
            b = (wk @ lin.proj.bias).to(f.dtype)
            ys.append(F.conv2d(f, w[:, :, None, None], b))             # at the stage's own resolution
        full, lows = ys[3], ys[:3]
        if full.is_cuda and full.dtype in (torch.float32, torch.bfloat16) and all(y.dtype == full.dtype for y in lows):
            from .. import _lib
            acc = _lib.upsample_sum(full.contiguous(), [y.contiguous() for y in lows])      # one pass: resize x3 + add (HIP)
        else:
            acc = full
            for y in lows:
                acc = acc + F.interpolate(y, size=size, mode='bilinear', align_corners=False)
        x = bn_act(self.linear_fuse.bn, acc.contiguous(), 'relu')
        return self.linear_pred(self.dropout(x))


class SegFormer(nn.Module):
    """model(x) -> (logits [N, K, H, W], pooled [N, 512] float32): the global average of the last encoder stage plays the role
    of smp's pooled encoder feature (models/heads.py:19-25)."""

    def __init__(self, classes=2, variant=None, aux_pooling=True):
        super().__init__()
        v = dict(MIT_B2 if variant is None else variant)
        self.backbone = MixVisionTransformer(**v)
        self.head = SegFormerHead(v['embed_dims'], v['head_dim'], classes)
        self.feature_channels = v['embed_dims'][-1]
        self.aux_pooling = aux_pooling

    def forward(self, x):
        feats = self.backbone(x)
        pred = self.head(feats)
        mask = _upsample_ac(pred.contiguous(), x.shape[-2:])           # nn.UpsamplingBilinear2d(scale_factor=4): align_corners=True
        if not self.aux_pooling:
            return mask
        return mask, global_avg_pool_f32(feats[3])
