"""Plain PyTorch float32 statement of the float tensor ops -- test infrastructure, the yardstick for the `aadg_fop_f32` kernels at
realistic shapes (until round 4 every check above 20 x 24 pixels compared one HIP kernel with another).

What it states: the forward semantics of the reference's data/functional.py:110-280 -- every op maps a float32 batch [B,3,H,W] in
[0,1] (+ a magnitude: one value or one per sample) to a clamped batch.  The 15 ops whose arithmetic lives in the reference itself are
PINNED: tests/test_golden_cpu.py::test_fop_torch_statement_vs_reference_golden runs this file on torch-CPU against
tests/golden/functional.npz (outputs of the reference, `make_golden.py: gen_functional`).  The four warps and hue delegate to an
unpinned kornia in the reference ("parity unpinned"): for them this file states the documented convention (centre-pivot
inverse affine map, bilinear taps on pixel centres, zeros outside; HSV with h in [0,1]) through torch.nn.functional.grid_sample and
a vectorised colorsys -- a second, independent statement, not a pin.

Run it on the CPU (`device="cpu"`): ATen's CPU `histc` is what produced the goldens, and a CPU run shares nothing with the kernels
under test -- neither the device libraries nor the reduction orders."""
import math

import torch
import torch.nn.functional as F

LUMA = (0.299, 0.587, 0.110)          # data/functional.py:83-85 (the blue weight is 0.110 there, not 0.114)


def _per_sample(mag, B):
    m = torch.as_tensor(mag, dtype=torch.float32).reshape(-1)
    return (m if m.numel() == B else m.expand(B)).reshape(B, 1, 1, 1)


def _luma(x):
    return LUMA[0] * x[:, 0:1] + LUMA[1] * x[:, 1:2] + LUMA[2] * x[:, 2:3]


def _mix(x, other, keep):
    """keep = 1 returns x, keep = 0 returns `other` (data/functional.py:74-80); clamped."""
    return (other + keep * (x - other)).clamp(0, 1)


def _box3(x, k):
    """reflect-padded depthwise 3 x 3 correlation (data/functional.py:96-106)"""
    C = x.shape[1]
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), k.to(x).reshape(1, 1, 3, 3).repeat(C, 1, 1, 1), groups=C)


def invert(x, mag=None):
    return (1.0 - x).clamp(0, 1)


def solarize(x, mag):
    t = _per_sample(mag, x.shape[0])
    return torch.where(x < t, x, 1.0 - x).clamp(0, 1)


def posterize(x, mag):
    # the shift pair of :177-179 acts on int64 and drops no bit: what is left is the truncation to 1/255 steps
    return ((x * 255).to(torch.int64).to(torch.float32) / 255).clamp(0, 1)


def gray(x, mag=None):
    return _luma(x).expand(-1, 3, -1, -1).clamp(0, 1)


def contrast(x, mag):
    B = x.shape[0]
    level = (_luma(x * 255).reshape(B, -1).mean(dim=1) + 0.5).floor().reshape(B, 1, 1, 1) / 255
    return _mix(x, level, 1 - _per_sample(mag, B))


def auto_contrast(x, mag=None):
    B, C, H, W = x.shape
    v = x.reshape(B * C, H * W).clamp(0, 1) * 255
    lo = v.min(dim=1, keepdim=True).values
    hi = v.max(dim=1, keepdim=True).values
    table = ((torch.arange(256, dtype=torch.float32) - lo) * (255 / (hi - lo + 0.1))).floor()
    return (table.gather(1, v.to(torch.int64)).reshape(B, C, H, W) / 255).clamp(0, 1)


def saturate(x, mag):
    return _mix(x, _luma(x), 1 - _per_sample(mag, x.shape[0]))


def brightness(x, mag):
    return _mix(x, torch.zeros_like(x), 1 - _per_sample(mag, x.shape[0]))


def sample_pairing(x, mag, perm):
    m = _per_sample(mag, x.shape[0])
    return ((1 - m) * x + m * x[perm.to(torch.int64)]).clamp(0, 1)


def equalize(x, mag=None):
    """:241-262 -- ONE torch.histc over all B*C planes, each shifted into its own 256-value band, with `max = 256*BC - 1` (so the bins
    are slightly narrower than 1: plane p's values move up by up to p/BC of a bin -- part of the reference's behaviour and reproduced
    by calling histc the same way); Pillow-style step / half-step offset; lookup by truncation."""
    B, C, H, W = x.shape
    P = B * C
    v = x.reshape(P, H, W).clamp(0, 1) * 255
    band = v + 256 * torch.arange(P, dtype=torch.float32).reshape(P, 1, 1)
    hist = band.histc(P * 256, 0, P * 256 - 1).reshape(P, 256)
    run = hist.cumsum(dim=1)
    step = ((run[:, 255] - hist[:, 255]) / 255).floor().reshape(P, 1)
    before = torch.cat([torch.zeros(P, 1), run[:, :255]], dim=1) + (step / 2).floor()
    table = (before / (step + 0.1)).floor().reshape(-1)
    return (table[band.to(torch.int64)].reshape(B, C, H, W) / 255).clamp(0, 1)


def sharpness(x, mag, kernel):
    return _mix(x, _box3(x, kernel), 1 - _per_sample(mag, x.shape[0]))


def gaussian_blur3x3(x, mag, kernel):
    return _box3(x, kernel).clamp(0, 1)


def hflip(x, mag=None):
    return x.flip(3)


def vflip(x, mag=None):
    return x.flip(2)


# ---- the kornia family (unpinned): the documented convention through grid_sample / colorsys --------------------------------
def _warp(x, lin, shift):
    """out(p) = bilinear(x, A^-1 (p - c - t) + c), c = ((W-1)/2, (H-1)/2), zeros outside.  lin: [B,2,2] forward matrices A (x, y
    order), shift: [B,2] translations t in pixels.  Evaluated in float64 (coordinates AND taps): grid_sample's normalised
    coordinates cost two extra roundings, which in float32 at 1024 pixels is ~1e-4 of a pixel -- the yardstick must not be the
    noisier side of the comparison."""
    B, C, H, W = x.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    cx, cy = (W - 1) / 2, (H - 1) / 2
    inv = torch.linalg.inv(lin.to(torch.float64))
    shift = shift.to(torch.float64)
    dx = xs[None] - cx - shift[:, 0].reshape(B, 1, 1)
    dy = ys[None] - cy - shift[:, 1].reshape(B, 1, 1)
    sx = inv[:, 0, 0].reshape(B, 1, 1) * dx + inv[:, 0, 1].reshape(B, 1, 1) * dy + cx
    sy = inv[:, 1, 0].reshape(B, 1, 1) * dx + inv[:, 1, 1].reshape(B, 1, 1) * dy + cy
    grid = torch.stack([2 * sx / (W - 1) - 1, 2 * sy / (H - 1) - 1], dim=-1)
    out = F.grid_sample(x.to(torch.float64), grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.to(torch.float32).clamp(0, 1)


def _eye(B):
    return torch.eye(2).repeat(B, 1, 1)


def shear_x(x, mag):
    B = x.shape[0]
    A = _eye(B)
    A[:, 0, 1] = _per_sample(mag, B).reshape(B)
    return _warp(x, A, torch.zeros(B, 2))


def shear_y(x, mag):
    B = x.shape[0]
    A = _eye(B)
    A[:, 1, 0] = _per_sample(mag, B).reshape(B)
    return _warp(x, A, torch.zeros(B, 2))


def translate_x(x, mag):
    B = x.shape[0]
    t = torch.zeros(B, 2)
    t[:, 0] = _per_sample(mag, B).reshape(B) * x.shape[3]          # float32 product, as the reference forms it (:127)
    return _warp(x, _eye(B), t)


def translate_y(x, mag):
    B = x.shape[0]
    t = torch.zeros(B, 2)
    t[:, 1] = _per_sample(mag, B).reshape(B) * x.shape[2]
    return _warp(x, _eye(B), t)


def rotate(x, mag):
    """degrees, counter-clockwise on the screen (y down)"""
    B = x.shape[0]
    a = _per_sample(mag, B).reshape(B).to(torch.float64) * (math.pi / 180)
    A = torch.stack([torch.stack([a.cos(), a.sin()], 1), torch.stack([-a.sin(), a.cos()], 1)], 1)
    return _warp(x, A, torch.zeros(B, 2))


def hue(x, mag):
    """colorsys.rgb_to_hsv / hsv_to_rgb, vectorised; h in [0,1], shifted by mag modulo 1 (data/functional.py:224-230)"""
    r, g, b = x[:, 0], x[:, 1], x[:, 2]
    v = x.max(dim=1).values
    lo = x.min(dim=1).values
    span = v - lo
    s = torch.where(v > 0, span / v.clamp_min(1e-30), torch.zeros_like(v))
    safe = torch.where(span > 0, span, torch.ones_like(span))
    rc, gc, bc = (v - r) / safe, (v - g) / safe, (v - b) / safe
    h = torch.where(r == v, bc - gc, torch.where(g == v, 2.0 + rc - bc, 4.0 + gc - rc))
    h = torch.where(span > 0, (h / 6.0) % 1.0, torch.zeros_like(h))
    h = (h + _per_sample(mag, x.shape[0]).reshape(-1, 1, 1)) % 1.0
    i = (h * 6.0).floor()
    f = h * 6.0 - i
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    i = i.to(torch.int64) % 6
    sel = lambda a0, a1, a2, a3, a4, a5: torch.where(i == 0, a0, torch.where(i == 1, a1, torch.where(i == 2, a2, torch.where(
        i == 3, a3, torch.where(i == 4, a4, a5)))))
    return torch.stack([sel(v, q, p, p, t, v), sel(t, v, v, q, p, p), sel(p, p, t, v, v, q)], dim=1).clamp(0, 1)


PINNED = ("invert", "solarize", "posterize", "gray", "contrast", "auto_contrast", "saturate", "brightness", "sample_pairing",
          "equalize", "sharpness", "gaussian_blur3x3", "hflip", "vflip")
UNPINNED = ("shear_x", "shear_y", "translate_x", "translate_y", "rotate", "hue")


def run(name, x, mag=None, kernel=None, perm=None):
    """x: CPU float32 [B,3,H,W]; returns the op's output (CPU float32)."""
    fn = globals()[name]
    if name == "sample_pairing":
        return fn(x, mag, perm)
    if name in ("sharpness", "gaussian_blur3x3"):
        return fn(x, mag, kernel)
    return fn(x, mag)
