"""GPU parity of the nn.Module wrappers of the float ops against fixtures the REFERENCE's own
`_Operation.forward` produced (data/operations.py:73-100; tests/golden/make_golden.py: gen_operations): training mode
(RelaxedBernoulli mask blend `mask*op(x) + (1-mask)*x`), eval mode (Bernoulli mask, op applied in place to the selected
samples), magnitude clamp * scale, per-sample sign flip.  The random draws are injected: the fixtures record the mask, the
0/1 sign draws and the permutation rule, because CPU and GPU generators give different streams.
Also: the straight-through estimator `ste` (data/functional.py:21-46), forward value and gradient routing.
Tolerance 1e-5 (float32); the two histogram ops may put <= 0.1 % of the pixels into the neighbouring bin (as in
tests/test_gpu_functional.py)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _golden():
    return np.load(os.path.join(GOLDEN, "operations.npz"))


def test_operation_modules_vs_reference_golden(hip, monkeypatch):
    from aadg_amd.data import operations as Ops
    z = _golden()
    img = torch.from_numpy(z["img"]).cuda()
    signs01 = torch.from_numpy(z["signs01"]).cuda()
    assert str(z["perm_rule"]) == "randperm(n) := (arange(n) + 1) % n"
    monkeypatch.setattr(torch, "randint", lambda *a, **k: signs01.clone())
    monkeypatch.setattr(torch, "randperm", lambda n, **k: ((torch.arange(n) + 1) % n).to(k.get("device", "cpu")))
    seen = set()
    for key in [str(k) for k in z["keys"]]:
        name, ci, mode = key.split("_")
        mag0, prob0 = z[key + "_cfg"]
        cls = getattr(Ops, name)
        op = cls(initial_probability=float(prob0)) if np.isnan(mag0) else cls(initial_magnitude=float(mag0),
                                                                             initial_probability=float(prob0))
        op = op.cuda()
        op.train() if mode == "train" else op.eval()
        mask = torch.from_numpy(z[key + "_mask"]).cuda()
        op.get_mask = lambda batch_size=None, m=mask: m.clone()
        with torch.no_grad():
            got = op(img.clone())
        diff = np.abs(got.cpu().numpy() - z[key])
        if name in ("Equalize", "AutoContrast"):
            assert (diff > TOL).mean() <= 1e-3 and diff.max() <= 2.0 / 255 + TOL, (key, diff.max())
        else:
            assert diff.max() <= TOL, (key, diff.max())
        seen.add(name)
    assert seen == {"Invert", "Solarize", "Posterize", "Gray", "Contrast", "AutoContrast", "Saturate", "Brightness", "SamplePairing",
                    "Equalize", "Sharpness", "HorizontalFlip", "VerticalFlip"}


def test_operation_draws_follow_the_reference_distributions(hip):
    """get_mask: RelaxedBernoulli(T, p).rsample((B,1,1)) in training, Bernoulli(p) in eval (data/operations.py:102-108)."""
    from aadg_amd.data import operations as Ops
    op = Ops.Brightness(initial_magnitude=0.5, initial_probability=0.7).cuda()
    op.train()
    m = op.get_mask(4096)
    assert m.shape == (4096, 1, 1, 1) and float(m.detach().min()) >= 0 and float(m.detach().max()) <= 1
    assert abs(float((m.detach() > 0.5).float().mean()) - 0.7) < 0.05 and m.requires_grad
    op.eval()
    m = op.get_mask(4096)
    assert set(np.unique(m.cpu().numpy())) <= {0.0, 1.0} and abs(float(m.mean()) - 0.7) < 0.05
    # magnitude = clamp(_magnitude, range) * scale (data/operations.py:110-119)
    assert abs(float(Ops.Rotate(initial_magnitude=1.7).magnitude.detach()) - 30.0) < 1e-6
    assert abs(float(Ops.ShearX(initial_magnitude=0.5).magnitude.detach()) - 0.15) < 1e-6
    assert abs(float(Ops.TranslateY(initial_magnitude=0.2).magnitude.detach()) - 0.09) < 1e-6
    assert abs(float(Ops.Hue(initial_magnitude=0.25).magnitude.detach()) - 0.5) < 1e-6


def test_ste_forward_and_backward_vs_reference_golden(hip):
    from aadg_amd.data import functional as Fn
    z = _golden()
    a = torch.from_numpy(z["ste_a"]).cuda()
    b = torch.from_numpy(z["ste_b"]).cuda().requires_grad_(True)
    y = Fn.ste(a, b)
    assert torch.equal(y.detach().cpu(), torch.from_numpy(z["ste_y"])) and y.data_ptr() != a.data_ptr()     # forward = first argument, cloned
    y.backward(torch.from_numpy(z["ste_g"]).cuda())
    assert np.abs(b.grad.cpu().numpy() - z["ste_grad_b"]).max() <= 1e-5                                 # gradient summed to b's shape
    assert a.grad is None
    # through an op: solarize sends the output gradient to the magnitude (data/functional.py:163)
    img = torch.from_numpy(z["img"]).cuda()
    m = torch.from_numpy(z["sol_mag"]).cuda().requires_grad_(True)
    r = Fn.solarize(img.clone(), m)
    assert np.abs(r.detach().cpu().numpy() - z["sol_out"]).max() <= TOL
    r.backward(torch.from_numpy(z["sol_g"]).cuda())
    assert np.abs(m.grad.cpu().numpy() - z["sol_grad_mag"]).max() <= 1e-3 * np.abs(z["sol_grad_mag"]).max()
