"""Round 4: every `aadg_fop_f32` op at realistic shapes against an INDEPENDENT statement -- tests/fop_torch.py,
plain PyTorch float32 run on the CPU, itself pinned to the reference's goldens in the CPU suite
(tests/test_golden_cpu.py::test_fop_torch_statement_vs_reference_golden).  No op is left whose only check above 20 x 24 pixels is
another HIP kernel.

Shapes [16,3,512,512] and [3,3,1024,1024]; scalar and per-sample magnitudes; 16-byte-aligned inputs (vector kernels) and the same
values at a 4-byte offset (one-pixel-per-lane kernels); sample_pairing at B = 48 (> 128 MB: the cycle walk of k_fop_pair_order).

Tolerances (float32, written here as north_star asks):
  * 1e-5 absolute for every op of the reference's own arithmetic (data/functional.py:158-280);
  * equalize / auto_contrast: a pixel within float rounding of a histogram-bin edge may land in the neighbouring bin, at most 0.1 % of
    the pixels, each by at most two LUT steps;
  * contrast: a sample whose mean luma lies within 1e-3 of a half-integer is ill-conditioned in the reference itself (the
    `floor(mean + 0.5)` of :191 flips on the summation order) and is left out of the comparison -- none occurs with these seeds;
  * the kornia family (warps, hue; parity unpinned): 1e-4 against the grid_sample / colorsys statement of the convention."""
import numpy as np
import pytest
import torch

import fop_torch as FT

pytestmark = pytest.mark.gpu
TOL = 1e-5
TOL_UNPINNED = 1e-4
SHAPES = {"512": (16, 512, 512), "1024": (3, 1024, 1024)}
MAGS = {"solarize": 0.45, "posterize": 0.5, "contrast": 0.35, "saturate": 0.4, "brightness": 0.3, "hue": 0.27, "sample_pairing": 0.3,
        "sharpness": 0.4, "gaussian_blur3x3": 0.8, "shear_x": 0.21, "shear_y": -0.17, "translate_x": 0.113, "translate_y": -0.071,
        "rotate": 23.0}


@pytest.fixture(scope="module", autouse=True)
def _cpu_threads():
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 16))            # the GPU boxes' containers have a 16-core quota
    yield
    torch.set_num_threads(n)


def _images(B, H, W, seed):
    """Image-like content: a smooth field with a per-sample colour cast + noise; every other sample quantised to k/255 (what a
    ToTensor'd uint8 image holds: 255 * (k/255) sits exactly ON the bin edges of the LUT ops); one constant plane and one with a
    reduced range (degenerate / narrow histograms)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    x = torch.empty(B, 3, H, W)
    for b in range(B):
        field = 0.45 + 0.3 * torch.sin(xx / (9.0 + b)) * torch.cos(yy / (13.0 + 2 * b))
        cast = torch.tensor([1.0, 0.85 - 0.02 * b, 0.55 + 0.03 * b]).reshape(3, 1, 1)
        x[b] = (field[None] * cast + 0.25 * torch.rand(3, H, W, generator=g)).clamp(0, 1)
        if b % 2 == 1:
            x[b] = (x[b] * 255).round() / 255
    x[0, 1] = 0.5 * x[0, 1]
    if B > 2:
        x[2, 2] = 0.25
    return x.contiguous()


def _unaligned(x):
    buf = torch.empty(x.numel() + 1, device=x.device, dtype=x.dtype)
    v = buf[1:].view(x.shape)
    v.copy_(x)
    assert v.is_contiguous() and v.data_ptr() % 16 == 4
    return v


def _kernel_for(name, mag):
    from aadg_amd.data.kernels import get_gaussian_3x3kernel, get_sharpness_kernel
    if name == "sharpness":
        return get_sharpness_kernel()
    if name == "gaussian_blur3x3":
        return get_gaussian_3x3kernel(mag[:1])
    return None


def _compare(name, got, want, x_cpu, mag):
    d = (got - want).abs()
    if name in ("equalize", "auto_contrast"):
        frac = float((d > TOL).float().mean())
        assert frac <= 1e-3, (name, frac)
        assert float(d.max()) <= 2.0 / 255 + TOL, (name, float(d.max()))
        return
    if name == "contrast":
        luma = (0.299 * x_cpu[:, 0] + 0.587 * x_cpu[:, 1] + 0.110 * x_cpu[:, 2]).double().flatten(1).mean(1) * 255
        t = (luma + 0.5) % 1.0
        keep = torch.minimum(t, 1.0 - t) > 1e-3
        assert int(keep.sum()) >= x_cpu.shape[0] - 1, "too many ill-conditioned samples: pick other seeds"
        d = d[keep]
    tol = TOL_UNPINNED if name in FT.UNPINNED else TOL
    assert float(d.max()) <= tol, (name, float(d.max()), float((d > tol).float().mean()))


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_every_float_op_vs_torch_statement(hip, shape):
    B, H, W = SHAPES[shape]
    x_cpu = _images(B, H, W, seed=100 + H)
    x = x_cpu.cuda()
    xu = _unaligned(x)
    perm = torch.roll(torch.arange(B), 1)
    perm[0], perm[1] = perm[1].clone(), perm[0].clone()
    names = [n for n in hip.FOP]
    assert set(names) == set(FT.PINNED) | set(FT.UNPINNED), set(names) ^ (set(FT.PINNED) | set(FT.UNPINNED))
    for name in names:
        m0 = MAGS.get(name)
        variants = [None] if m0 is None else [torch.tensor([m0])]
        if m0 is not None and name != "gaussian_blur3x3":                         # the blur's magnitude only picks the kernel
            variants.append(torch.linspace(-0.6 if name in ("shear_x", "shear_y", "translate_x", "translate_y", "rotate") else 0.15,
                                           1.0, B) * m0)
        for mag in variants:
            kernel = _kernel_for(name, mag)
            kw = {}
            if kernel is not None:
                kw["kernel"] = kernel.cuda()
            if name == "sample_pairing":
                kw["perm"] = perm.cuda()
            want = FT.run(name, x_cpu, mag, kernel=kernel, perm=perm)
            mg = None if mag is None else mag.cuda()
            got = hip.fop(name, x, mg, **kw).cpu()
            _compare(name, got, want, x_cpu, mag)
            got_u = hip.fop(name, xu, mg, **kw).cpu()
            _compare(name, got_u, want, x_cpu, mag)
            del got, got_u, want


def test_functional_api_at_512_vs_torch_statement(hip):
    """The same through the reference-named wrappers (aadg_amd.data.functional: tensor_function checks, default kernels, the STE
    wrappers of solarize / posterize) for the ops the search uses most, [16,3,512,512]."""
    from aadg_amd.data import functional as Fn
    B, H, W = 16, 512, 512
    x_cpu = _images(B, H, W, seed=7)
    x = x_cpu.cuda()
    mag = torch.linspace(0.1, 0.9, B)
    from aadg_amd.data.kernels import get_sharpness_kernel
    for name in ("solarize", "posterize", "contrast", "saturate", "brightness", "sharpness"):
        want = FT.run(name, x_cpu, mag, kernel=get_sharpness_kernel() if name == "sharpness" else None)
        got = getattr(Fn, name)(x.clone(), mag.cuda()).detach().cpu()
        _compare(name, got, want, x_cpu, mag)
    for name in ("invert", "gray", "auto_contrast", "equalize", "hflip", "vflip"):
        _compare(name, getattr(Fn, name)(x.clone()).cpu(), FT.run(name, x_cpu), x_cpu, None)


def test_sample_pairing_cycle_walk_vs_torch_statement(hip):
    """B = 48 at 512 x 512 (151 MB > the 128 MB streaming threshold): aadg_fop_f32 walks the cycles of `perm`; the torch statement
    indexes x[perm] -- permutations with long cycles, fixed points, and a non-permutation."""
    B, H = 48, 512
    g = torch.Generator().manual_seed(9)
    x_cpu = torch.rand(B, 3, H, H, generator=g)
    x = x_cpu.cuda()
    mag = torch.rand(B, generator=g) * 0.4
    p1 = torch.randperm(B, generator=g)
    p2 = torch.arange(B)
    p2[:10] = torch.roll(p2[:10], 1)
    p2[20:23] = torch.roll(p2[20:23], -1)
    p3 = torch.randint(0, B, (B,), generator=g)
    for perm in (p1, p2, p3):
        want = FT.sample_pairing(x_cpu, mag, perm)
        got = hip.fop("sample_pairing", x, mag.cuda(), perm=perm.cuda()).cpu()
        assert float((got - want).abs().max()) <= TOL
    want = FT.sample_pairing(x_cpu, torch.tensor([0.3]), p1)
    got = hip.fop("sample_pairing", x, torch.tensor([0.3]).cuda(), perm=p1.cuda()).cpu()
    assert float((got - want).abs().max()) <= TOL
