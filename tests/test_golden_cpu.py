"""CPU suite, part 1: the oracle (and aadg_amd's host-side draw logic) against the golden vectors the
reference itself produced (tests/golden/make_golden.py), and against live Pillow when importable."""
import json
import math
import os

import numpy as np
import pytest

from helpers import GOLDEN, Cfg, draw_batch, load_pipeline_golden


def test_oracle_ops_match_reference_golden(oracle):
    """All 10 registry ops x 10 magnitude levels, bit-exact vs reference apply_augment outputs."""
    from aadg_amd.data.basic import cutout_rect
    z = np.load(os.path.join(GOLDEN, "u8_ops.npz"))
    los = [0, 0, 0, 0, 4, .1, .1, .1, .1, 0]
    his = [1, 1, 1, 256, 8, 1.9, 1.9, 1.9, 1.9, .2]
    for tag in ("a", "b"):
        img = z["img_" + tag]
        H, W, _ = img.shape
        for op in range(10):
            for li in range(10):
                v = li / 9 * (his[op] - los[op]) + los[op]
                iarg, farg, rect = 0, 0.0, None
                if op == 3:
                    iarg = int(math.ceil(v))
                elif op == 4:
                    iarg = int(v)
                elif op in (5, 6, 7, 8):
                    farg = np.float32(v)
                elif op == 9:
                    if v <= 0:
                        rect = (0, 0, -1, -1)
                    else:
                        np.random.seed(1000 * op + li)
                        x0, y0 = np.random.uniform(W), np.random.uniform(H)
                        rect = cutout_rect(W, H, v * W, x0, y0)
                got = oracle.op_u8(img, op, iarg, farg, rect)
                assert np.array_equal(got, z["out_" + tag][op, li]), (tag, op, li)


def test_parse_policies_bit_exact():
    """Policy indexing must be bit-exact (north_star): names AND float64 levels."""
    from aadg_amd.data.policy import parse_policies
    with open(os.path.join(GOLDEN, "parse_policies.json")) as f:
        cases = json.load(f)
    for c in cases:
        got = parse_policies(np.array(c["policies"], np.int64), Cfg(L=c["L"], EXCLUDE_OPS=c["exclude"]), None)
        want = c["parsed"]
        assert len(got) == len(want)
        for gp, wp in zip(got, want):
            assert len(gp) == len(wp)
            for gs, ws in zip(gp, wp):
                assert [(n, float(v)) for n, v in gs] == [(n, v) for n, v in ws]


@pytest.mark.parametrize("case", range(4))
def test_pipeline_seed_for_seed_vs_reference(oracle, case):
    """Same seeds as the reference run -> aadg_amd's host code records the same draws -> the oracle
    reproduces the reference's collated tensors exactly (images, labels, soft domain codes)."""
    z, meta = load_pipeline_golden()
    m = meta[case]
    name = m["name"]
    pool, flat, refs, M = draw_batch(z[name + "_pool_img"], z[name + "_pool_msk"], z[name + "_policies"], m)
    from aadg_amd.data.transform import refs_to_units
    import torch
    units = refs_to_units(refs)
    S = len(flat)
    assert [int(u) for u in units["src"][:S]] == [int(p) for p in z[name + "_picks"]]
    kind = 0 if m["dataset"] == "optic" else 1
    img, lbl = oracle.aug_units(pool.images.numpy(), pool.masks.numpy(), units, m["crop"], kind)
    lut = z["lut256"]
    assert np.array_equal(img[:S], lut[z[name + "_image"]])
    assert np.array_equal(lbl[:S], z[name + "_label"].astype(np.float32))
    assert np.array_equal(img[S:], lut[z[name + "_aug_images"]])
    assert np.array_equal(lbl[S:], z[name + "_aug_labels"].astype(np.float32))
    dc = torch.cat([b['dc'] for b in flat], dim=0).numpy()
    assert np.array_equal(dc, z[name + "_dc"])


def test_oracle_vs_live_pillow(oracle):
    """Stronger pin when Pillow is importable: random sizes/scales, every op, resize both filters."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageOps, ImageEnhance
    rs = np.random.RandomState(5)
    for (H, W) in [(32, 32), (17, 45), (64, 48)]:
        a = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        a[:, : W // 2, 1] //= 3
        pil = Image.fromarray(a)
        assert np.array_equal(np.asarray(ImageOps.autocontrast(pil)), oracle.op_u8(a, 0))
        assert np.array_equal(np.asarray(ImageOps.equalize(pil)), oracle.op_u8(a, 2))
        for f in (0.1, 0.5, 0.9, 1.0, 1.3, 1.9):
            for op, E in ((5, ImageEnhance.Contrast), (6, ImageEnhance.Color), (7, ImageEnhance.Brightness),
                          (8, ImageEnhance.Sharpness)):
                assert np.array_equal(np.asarray(E(pil).enhance(f)), oracle.op_u8(a, op, farg=np.float32(f))), (op, f)
        m = (rs.randint(0, 3, (H, W)) * 127).astype(np.uint8)
        pm = Image.fromarray(m)
        for _ in range(25):
            w, h = int(rs.uniform(0.5, 2) * W), int(rs.uniform(0.5, 2) * H)
            assert np.array_equal(np.asarray(pil.resize((w, h), Image.BILINEAR)), oracle.resize_bilinear(a, w, h))
            assert np.array_equal(np.asarray(pm.resize((w, h), Image.NEAREST)), oracle.resize_nearest(m, w, h))


def test_ste_and_operation_parameters_vs_reference_golden():
    """`ste` (data/functional.py:21-46) is pure torch: forward value, gradient routed to the second argument and summed to its
    shape -- against the reference's own outputs; `_Operation` magnitude = clamp(_magnitude, range) * scale
    (data/operations.py:110-119) and the registered parameters / buffers."""
    import torch
    from aadg_amd.data import functional as Fn
    from aadg_amd.data import operations as Ops
    z = np.load(os.path.join(GOLDEN, "operations.npz"))
    a = torch.from_numpy(z["ste_a"])
    b = torch.from_numpy(z["ste_b"]).requires_grad_(True)
    y = Fn.ste(a, b)
    assert np.array_equal(y.detach().numpy(), z["ste_y"])
    y.backward(torch.from_numpy(z["ste_g"]))
    assert np.abs(b.grad.numpy() - z["ste_grad_b"]).max() <= 1e-5
    op = Ops.Contrast(initial_magnitude=1.3, initial_probability=0.1)
    assert float(op.magnitude) == 1.0 and float(op.probability) == pytest.approx(0.1)
    assert set(dict(op.named_parameters())) == {"_magnitude", "_probability"} and "temperature" in dict(op.named_buffers())
    assert Ops.Sharpness().kernel.shape == (3, 3) and Ops.Invert()._magnitude is None


def test_fop_torch_statement_vs_reference_golden():
    """Round 4: tests/fop_torch.py -- the plain PyTorch float32 statement the GPU suite compares the `aadg_fop_f32` kernels with at
    512 x 512 / 1024 x 1024 -- is itself pinned to the reference's own outputs (tests/golden/functional.npz: every non-kornia op of
    data/functional.py on torch-CPU, scalar and per-sample magnitudes).  Exact for the integer-valued ops, 1e-6 otherwise."""
    import torch
    import fop_torch as FT
    z = np.load(os.path.join(GOLDEN, "functional.npz"))
    img = torch.from_numpy(z["img"])
    sharp_k = torch.from_numpy(z["sharp_kernel"])
    seen = set()
    for key in [str(k) for k in z["keys"]]:
        name = key[4:]
        mag = z[key + "_mag"]
        mag_t = None if mag.size == 0 else torch.from_numpy(mag)
        kernel = None
        if name == "sharpness":
            kernel = sharp_k
        elif name == "gaussian_blur3x3":
            from aadg_amd.data.kernels import get_gaussian_3x3kernel
            kernel = get_gaussian_3x3kernel(mag_t)
        got = FT.run(name, img.clone(), mag_t, kernel=kernel).numpy()
        d = np.abs(got - z[key]).max()
        assert d <= (0.0 if name in ("equalize", "auto_contrast", "posterize", "invert", "hflip", "vflip", "solarize") else 1e-6), (key, d)
        seen.add(name)
    for i in range(2):
        got = FT.sample_pairing(img.clone(), torch.from_numpy(z["sp%d_mag" % i]), torch.from_numpy(z["sp%d_perm" % i])).numpy()
        assert np.abs(got - z["sp%d" % i]).max() <= 1e-6
        seen.add("sample_pairing")
    assert seen == set(FT.PINNED), set(FT.PINNED) ^ seen
    # the unpinned family: known answers of the convention (identity, integer shifts, quarter turn, grey has no hue)
    x = torch.rand(2, 3, 24, 24, generator=torch.Generator().manual_seed(0))
    for name in ("shear_x", "shear_y", "translate_x", "translate_y", "rotate", "hue"):
        assert torch.allclose(FT.run(name, x, torch.zeros(1)), x, atol=1e-6), name
    sh = FT.translate_x(x, torch.tensor([3 / 24]))
    assert torch.allclose(sh[..., 3:], x[..., :-3], atol=1e-6) and float(sh[..., :3].abs().max()) < 1e-6   # grid_sample's normalised coordinates
    assert torch.allclose(FT.rotate(x, torch.tensor([90.0])), torch.rot90(x, 1, (2, 3)), atol=1e-5)
    import colorsys
    out = FT.hue(x, torch.tensor([0.3, 0.8])).numpy()
    for (b, yy, xx) in ((0, 0, 0), (1, 3, 5), (0, 23, 7)):
        h, s, v = colorsys.rgb_to_hsv(*x[b, :, yy, xx].numpy().astype(np.float64))
        want = colorsys.hsv_to_rgb((h + (0.3, 0.8)[b]) % 1.0, s, v)
        assert np.allclose(out[b, :, yy, xx], want, atol=2e-6)
