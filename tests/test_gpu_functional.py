"""GPU parity tests for the float tensor ops (data/functional.py semantics).

Golden vectors for the ops whose arithmetic lives in the reference itself (everything except the kornia
geometric ops and hue) were produced by the reference on torch-CPU.  Tolerance: 1e-5 absolute, float32.
For the two histogram/LUT ops (equalize, auto_contrast) a pixel sitting within float rounding of a bin edge
may land in the neighbouring bin (output differs by one LUT step); at most 0.1 % of pixels may do so.
The kornia ops are "parity unpinned" and are pinned by known answers + a torch grid_sample restatement."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _golden():
    return np.load(os.path.join(GOLDEN, "functional.npz"))


def test_non_kornia_ops_vs_reference_golden(hip):
    from aadg_amd.data import functional as Fn
    z = _golden()
    img = torch.from_numpy(z["img"]).cuda()
    for key in [str(k) for k in z["keys"]]:
        name = key[4:]
        mag = z[key + "_mag"]
        fn = getattr(Fn, name)
        got = fn(img.clone()) if mag.size == 0 else fn(img.clone(), torch.from_numpy(mag).cuda())
        diff = np.abs(got.cpu().numpy() - z[key])
        if name in ("equalize", "auto_contrast"):
            assert (diff > TOL).mean() <= 1e-3, (key, (diff > TOL).mean())
            assert diff.max() <= 2.0 / 255 + TOL, key
        else:
            assert diff.max() <= TOL, (key, diff.max())
    for i in range(2):
        got = Fn.sample_pairing(img.clone(), torch.from_numpy(z["sp%d_mag" % i]).cuda(),
                                torch.from_numpy(z["sp%d_perm" % i]).cuda())
        assert np.abs(got.cpu().numpy() - z["sp%d" % i]).max() <= TOL


def test_kernels_module_matches_reference():
    from aadg_amd.data import kernels as K
    z = _golden()
    assert np.allclose(K.get_sharpness_kernel().numpy(), z["sharp_kernel"], atol=1e-7)
    for sg in (0.5, 1.0, 2.0):
        assert np.allclose(K.get_gaussian_3x3kernel(torch.tensor([sg])).numpy(), z["gauss3_%g" % sg], atol=1e-7)


def _affine_ref(img, A, t):
    """torch restatement: inverse map about the centre, bilinear on pixel centres, zero padding."""
    B, C, H, W = img.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=img.device),
                            torch.arange(W, dtype=torch.float32, device=img.device), indexing="ij")
    cx, cy = (W - 1) / 2, (H - 1) / 2
    Ai = torch.linalg.inv(torch.tensor(A, dtype=torch.float32))
    dx, dy = xs - cx - t[0], ys - cy - t[1]
    sx = Ai[0, 0] * dx + Ai[0, 1] * dy + cx
    sy = Ai[1, 0] * dx + Ai[1, 1] * dy + cy
    grid = torch.stack([2 * sx / (W - 1) - 1, 2 * sy / (H - 1) - 1], -1).unsqueeze(0).expand(B, -1, -1, -1)
    return torch.nn.functional.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def test_geometric_known_answers(hip):
    from aadg_amd.data import functional as Fn
    torch.manual_seed(0)
    img = torch.rand(2, 3, 24, 24, device="cuda")
    zero = torch.zeros(1, device="cuda")
    for fn in (Fn.shear_x, Fn.shear_y, Fn.translate_x, Fn.translate_y, Fn.rotate):
        assert torch.equal(fn(img, zero), img)                                   # identity at magnitude 0
    assert torch.equal(Fn.hflip(img), img.flip(3)) and torch.equal(Fn.vflip(img), img.flip(2))
    sh = Fn.translate_x(img, torch.tensor([3 / 24], device="cuda"))              # integer shift by 3 px
    assert torch.allclose(sh[..., 3:], img[..., :-3], atol=1e-6) and torch.all(sh[..., :3] == 0)
    sh = Fn.translate_y(img, torch.tensor([-2 / 24], device="cuda"))
    assert torch.allclose(sh[:, :, :-2], img[:, :, 2:], atol=1e-6) and torch.all(sh[:, :, -2:] == 0)
    r90 = Fn.rotate(img, torch.tensor([90.0], device="cuda"))                    # counter-clockwise
    assert torch.allclose(r90, torch.rot90(img, 1, (2, 3)), atol=1e-4)
    # against the grid_sample restatement, per-sample magnitudes
    mags = torch.tensor([0.21, -0.17], device="cuda")
    for b in range(2):
        m = mags[b].item()
        ref = _affine_ref(img[b:b + 1], [[1, m], [0, 1]], (0, 0))
        assert torch.allclose(Fn.shear_x(img, mags)[b:b + 1], ref, atol=1e-4)
        rad = math.radians(25 * m / 0.21)
        ang = torch.tensor([25 * mags[0].item() / 0.21, 25 * mags[1].item() / 0.21], device="cuda")
        ref = _affine_ref(img[b:b + 1], [[math.cos(rad), math.sin(rad)], [-math.sin(rad), math.cos(rad)]], (0, 0))
        assert torch.allclose(Fn.rotate(img, ang)[b:b + 1], ref, atol=1e-4)


def test_hue_known_answers(hip):
    import colorsys
    from aadg_amd.data import functional as Fn
    torch.manual_seed(1)
    img = torch.rand(1, 3, 8, 8, device="cuda")
    assert torch.allclose(Fn.hue(img, torch.zeros(1, device="cuda")), img, atol=1e-5)
    assert torch.allclose(Fn.hue(img, torch.ones(1, device="cuda")), img, atol=1e-5)       # (h + 1) % 1
    g = torch.full((1, 3, 4, 4), 0.37, device="cuda")
    assert torch.allclose(Fn.hue(g, torch.tensor([0.4], device="cuda")), g, atol=1e-6)      # grey has no hue
    out = Fn.hue(img, torch.tensor([0.3], device="cuda")).cpu().numpy()
    src = img.cpu().numpy()
    for (y, x) in ((0, 0), (3, 5), (7, 7)):
        h, s, v = colorsys.rgb_to_hsv(*src[0, :, y, x])
        want = colorsys.hsv_to_rgb((h + 0.3) % 1.0, s, v)
        assert np.allclose(out[0, :, y, x], want, atol=1e-5)


def test_operation_modules(hip):
    from aadg_amd.data import operations as Op
    torch.manual_seed(2)
    x = torch.rand(4, 3, 16, 16, device="cuda")
    assert len(Op.__all__) == 19
    for name in Op.__all__:
        mod = getattr(Op, name)().cuda()
        mod.train()
        y = mod(x.clone()).detach()
        assert y.shape == x.shape and float(y.min()) >= 0 and float(y.max()) <= 1, name
        mod.eval()
        y = mod(x.clone())
        assert y.shape == x.shape, name
    inv = Op.Invert(initial_probability=1.0, probability_range=None).cuda().eval()
    assert torch.allclose(inv(x.clone()), 1 - x, atol=1e-6)
    assert Op.ShearX().magnitude_scale == 0.3 and Op.TranslateY().magnitude_scale == 0.45
    assert Op.Rotate().magnitude_scale == 30 and Op.Hue().magnitude_scale == 2
    assert Op.Contrast().flip_magnitude and not Op.Solarize().flip_magnitude


def test_full_size_roundtrips(hip):
    """BASELINE size: involutions / idempotence through the kernels at [24,3,512,512]."""
    from aadg_amd.data import functional as Fn
    x = torch.rand(24, 3, 512, 512, device="cuda")
    assert torch.allclose(Fn.invert(Fn.invert(x)), x, atol=1e-6)
    assert torch.equal(Fn.hflip(Fn.hflip(x)), x) and torch.equal(Fn.vflip(Fn.vflip(x)), x)
    p = Fn.posterize(x, torch.tensor([0.5], device="cuda"))
    assert torch.equal(Fn.posterize(p, torch.tensor([0.5], device="cuda")), p)               # idempotent
    g = Fn.gray(x)
    assert torch.equal(g[:, 0], g[:, 1]) and torch.equal(g[:, 1], g[:, 2])
    e = Fn.equalize(x)
    assert float(e.min()) >= 0 and float(e.max()) <= 1 and abs(float(e.mean()) - 0.5) < 0.02  # flat histogram stays flat


def _unaligned_copy(x):
    """the same values at a 4-byte-aligned (not 16-byte-aligned) address: aadg_fop_f32 takes its one-pixel-per-lane kernels"""
    buf = torch.empty(x.numel() + 1, device=x.device, dtype=x.dtype)
    v = buf[1:].view(x.shape)
    v.copy_(x)
    assert v.is_contiguous() and v.data_ptr() % 16 == 4
    return v


def test_vector_and_scalar_kernels_agree(hip):
    """Round 3: every op has a 4-pixels-per-lane kernel (register-window stencil, tiled warps, fused statistics prologue) and a
    one-pixel-per-lane kernel for shapes / pointers the vector path cannot take.  Same arithmetic, so the results must agree: on a
    shape with a second, narrow column strip (W = 264), per-sample magnitudes, and an odd shape that only the scalar kernels take
    (against the torch formulas of data/functional.py where they are one-liners)."""
    from aadg_amd import _lib
    from aadg_amd.data import functional as Fn
    torch.manual_seed(5)
    B, H, W = 5, 70, 264
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    smooth = 0.5 + 0.35 * torch.sin(xx / 17.0) * torch.cos(yy / 11.0)
    x = (smooth[None, None] * torch.tensor([1.0, 0.8, 0.6], device="cuda")[None, :, None, None]
         + 0.15 * torch.rand(B, 3, H, W, device="cuda")).clamp(0, 1).contiguous()
    xu = _unaligned_copy(x)
    perm = torch.tensor([1, 2, 3, 4, 0], device="cuda", dtype=torch.int32)
    mags = {"solarize": 0.5, "posterize": 0.5, "contrast": 0.3, "saturate": 0.4, "brightness": 0.3, "hue": 0.2, "sample_pairing": 0.3,
            "sharpness": 0.4, "gaussian_blur3x3": 0.8, "shear_x": 0.2, "shear_y": -0.2, "translate_x": 0.11, "translate_y": -0.07, "rotate": 23.0}
    for name in _lib.FOP:
        m0 = mags.get(name)
        variants = [None] if m0 is None else [torch.tensor([m0], device="cuda")]
        if m0 is not None and name != "gaussian_blur3x3":
            variants.append(torch.linspace(0.5, 1.0, B, device="cuda") * m0)
        for mag in variants:
            kw = {"perm": perm} if name == "sample_pairing" else {}
            a = _lib.fop(name, x, mag, **kw)
            b = _lib.fop(name, xu, mag, **kw)
            d = (a - b).abs()
            if name in ("equalize", "auto_contrast", "contrast"):
                assert float(d.max()) <= 1e-6, (name, float(d.max()))
            else:
                assert torch.equal(a, b), (name, float(d.max()))
    # torch formulas (data/functional.py:158-280) on the vector path
    m = torch.tensor([0.4], device="cuda")
    assert torch.equal(Fn.invert(x.clone()), 1 - x)
    assert torch.equal(Fn.solarize(x.clone(), m).detach(), torch.where(x < m, x, 1 - x))
    gray = (0.299 * x[:, 0] + 0.587 * x[:, 1] + 0.110 * x[:, 2]).unsqueeze(1).repeat(1, 3, 1, 1)
    assert torch.allclose(Fn.gray(x.clone()), gray, atol=1e-6)
    assert torch.allclose(Fn.saturate(x.clone(), m), (gray + (1 - m) * (x - gray)).clamp(0, 1), atol=1e-6)
    k = torch.ones(3, 3, device="cuda"); k[1, 1] = 5; k /= 13
    blur = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), k.repeat(3, 1, 1, 1), groups=3)
    assert torch.allclose(Fn.sharpness(x.clone(), m, k), (blur + (1 - m) * (x - blur)).clamp(0, 1), atol=1e-5)
    # odd shape: scalar kernels only
    xo = torch.rand(3, 3, 33, 37, device="cuda")
    assert torch.equal(Fn.invert(xo.clone()), 1 - xo) and torch.equal(Fn.hflip(xo.clone()), xo.flip(3))
    blur = torch.nn.functional.conv2d(torch.nn.functional.pad(xo, (1, 1, 1, 1), mode="reflect"), k.repeat(3, 1, 1, 1), groups=3)
    assert torch.allclose(Fn.sharpness(xo.clone(), m, k), (blur + (1 - m) * (xo - blur)).clamp(0, 1), atol=1e-5)
    e = Fn.equalize(xo.clone())
    assert e.shape == xo.shape and float(e.min()) >= 0 and float(e.max()) <= 1


@pytest.mark.parametrize("shape", [(2, 3, 40, 512), (1, 3, 36, 1032), (2, 3, 70, 264)])
def test_stencil_strips_vs_torch_conv(hip, shape):
    """The register-window stencil: one 512-wide strip with 8 pixels per lane (no edge loads), several strips with interior edges
    (1032 = 512 + 512 + 8), 4 pixels per lane (264 = 256 + 8) -- against reflect-pad + depthwise conv2d (data/functional.py:98-106)
    and against the one-pixel-per-lane kernel (bit-exact: shared arithmetic)."""
    from aadg_amd import _lib
    from aadg_amd.data import functional as Fn
    from aadg_amd.data.kernels import get_gaussian_3x3kernel
    torch.manual_seed(11)
    x = torch.rand(*shape, device="cuda")
    m = torch.tensor([0.35], device="cuda")
    k = torch.ones(3, 3, device="cuda"); k[1, 1] = 5; k /= 13
    pad = torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect")
    blur = torch.nn.functional.conv2d(pad, k.repeat(3, 1, 1, 1), groups=3)
    got = Fn.sharpness(x.clone(), m, k)
    assert torch.allclose(got, (blur + (1 - m) * (x - blur)).clamp(0, 1), atol=1e-5)
    assert torch.equal(got, _lib.fop("sharpness", _unaligned_copy(x), m, kernel=k))
    g = get_gaussian_3x3kernel(torch.tensor([0.9])).cuda()
    gb = torch.nn.functional.conv2d(pad, g.repeat(3, 1, 1, 1), groups=3).clamp(0, 1)
    got = _lib.fop("gaussian_blur3x3", x, torch.tensor([0.9], device="cuda"), kernel=g)
    assert torch.allclose(got, gb, atol=1e-5)
    assert torch.equal(got, _lib.fop("gaussian_blur3x3", _unaligned_copy(x), torch.tensor([0.9], device="cuda"), kernel=g))


def test_sample_pairing_walks_the_cycles_of_perm(hip):
    """Batches beyond the streaming threshold (128 MB) take sample_pairing along the cycles of perm (k_fop_pair_order: the partner image of
    one step is the self image of the next, found in the Infinity Cache).  The result must be what the index-order path gives: every
    output sample equals the op on the two-image batch [x[b], x[perm[b]]] (small batch: plain order) -- for a permutation with several
    cycles and fixed points, and for a `perm` that is not a permutation."""
    torch.manual_seed(9)
    B, H = 48, 512                                              # 48 x 3 x 512 x 512 x 4 B = 151 MB
    x = torch.rand(B, 3, H, H, device="cuda")
    mag = torch.rand(B, device="cuda") * 0.4
    swap = torch.tensor([1, 0], device="cuda", dtype=torch.int32)
    perms = [torch.randperm(B, device="cuda").to(torch.int32)]
    p = torch.arange(B, device="cuda", dtype=torch.int32)
    p[:10] = torch.roll(p[:10], 1)                              # one 10-cycle, a 3-cycle, fixed points elsewhere
    p[20:23] = torch.roll(p[20:23], -1)
    perms.append(p)
    perms.append(torch.randint(0, B, (B,), device="cuda", dtype=torch.int32))      # repeated partners
    for perm in perms:
        got = hip.fop("sample_pairing", x, mag, perm=perm)
        for b in (0, 1, 5, 9, 21, 30, B - 1):
            pair = torch.stack([x[b], x[int(perm[b])]])
            want = hip.fop("sample_pairing", pair, mag[b].repeat(2), perm=swap)[0]
            assert torch.equal(got[b], want), b


def test_tile_kernel_image_major_numbering(hip):
    """rotate / shear_y number their tile workgroups image-major per XCD for the first 8 * (B // 8) images and in the plain order for the
    rest: B = 11 exercises both, B = 16 only the first -- against the one-pixel-per-lane kernels (same arithmetic, plain grid), with
    per-sample magnitudes so that a tile mapped to the wrong image cannot go unnoticed."""
    from aadg_amd import _lib
    torch.manual_seed(6)
    for B in (11, 16):
        x = torch.rand(B, 3, 96, 136, device="cuda")
        xu = _unaligned_copy(x)
        for name, top in (("rotate", 30.0), ("shear_y", 0.3)):
            mag = torch.linspace(-1.0, 1.0, B, device="cuda") * top
            assert torch.equal(_lib.fop(name, x, mag), _lib.fop(name, xu, mag)), (name, B)
