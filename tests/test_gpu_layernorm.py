"""GPU: residual add + LayerNorm kernel (csrc/layernorm.hip, the pre-norm blocks of the SegFormer backbone: mix_transformer.py:96-117)
against the plain PyTorch float32 reference of the same op: s = x + scale[sample] * r, y = F.layer_norm(s), and every gradient.
bfloat16 tensors: the reference is evaluated in float32 on the same rounded inputs (and on the rounded s); tolerance one bfloat16
rounding of the output."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,C", [(3, 50, 64), (2, 37, 128), (2, 19, 320), (3, 11, 512), (2, 9, 40)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["plain", "residual", "residual_scaled"])
def test_add_layer_norm_matches_float32_reference(hip, B, T, C, dtype, mode):
    torch.manual_seed(B * 1000 + C)
    x = (torch.randn(B, T, C, device="cuda") * 1.3 + 0.2).to(dtype)
    r = (torch.randn(B, T, C, device="cuda") * 0.7).to(dtype) if mode != "plain" else None
    sc = torch.tensor([0.0, 1.25, 1.25][:B], device="cuda") if mode == "residual_scaled" else None
    gamma = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device="cuda") * 0.1).requires_grad_(True)
    gy = torch.randn(B, T, C, device="cuda").to(dtype)
    gs = torch.randn(B, T, C, device="cuda").to(dtype)
    xa = x.clone().requires_grad_(True)
    ra = r.clone().requires_grad_(True) if r is not None else None
    out = hip.add_layer_norm(xa, ra, sc, gamma, beta, 1e-6)
    if r is None:
        y = out
        y.backward(gy)
    else:
        s, y = out
        torch.autograd.backward([s, y], [gs, gy])
    # reference in float32
    xr = x.float().clone().requires_grad_(True)
    rr = r.float().clone().requires_grad_(True) if r is not None else None
    g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    if r is None:
        sref = xr
    else:
        sref = xr + (sc[:, None, None] if sc is not None else 1.0) * rr
        sref = sref + (sref.to(dtype).float() - sref).detach()                       # the kernel normalises the ROUNDED sum
    yref = F.layer_norm(sref, (C,), g2, b2, 1e-6)
    if r is None:
        yref.backward(gy.float())
    else:
        torch.autograd.backward([sref, yref], [gs.float(), gy.float()])
    tol = 1e-5 if dtype == torch.float32 else 2 ** -7
    scale = lambda t: max(1.0, float(t.detach().abs().max()))                                  # noqa: E731
    assert (y.float() - yref).abs().max().item() <= tol * scale(yref)
    if r is not None:
        assert (s.float() - sref).abs().max().item() <= tol * scale(sref)
        assert (ra.grad.float() - rr.grad).abs().max().item() <= 2 * tol * scale(rr.grad)
    assert (xa.grad.float() - xr.grad).abs().max().item() <= 2 * tol * scale(xr.grad)
    assert torch.allclose(gamma.grad, g2.grad, rtol=2e-3, atol=2e-3 * scale(g2.grad))
    assert torch.allclose(beta.grad, b2.grad, rtol=2e-3, atol=2e-3 * scale(b2.grad))


def test_layer_norm_rejects_what_it_does_not_cover(hip):
    x = torch.randn(4, 10, 520, device="cuda")
    assert not hip.layernorm_supported(x, 520) and not hip.layernorm_supported(torch.randn(4, 10, 36, device="cuda"), 36)
    with pytest.raises(hip.AadgError):
        hip.add_layer_norm(x, None, None, torch.ones(520, device="cuda"), torch.zeros(520, device="cuda"), 1e-6)


def test_add_layer_norm_validates_branch_and_scale(hip):
    """the public wrapper checks what the kernel relies on -- a branch `r` that is not contiguous is copied (the kernel
    reads it with row stride C: a transposed view used to give silently wrong sums), and the per-sample stochastic-depth factors
    must be a contiguous float32 vector with one entry per sample."""
    torch.manual_seed(0)
    B, T, C = 2, 16, 64
    x = torch.randn(B, T, C, device="cuda")
    rt = torch.randn(B, C, T, device="cuda").transpose(1, 2)            # [B, T, C] view with strides (C*T, 1, T)
    assert not rt.is_contiguous()
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    s1, y1 = hip.add_layer_norm(x, rt, None, g, b, 1e-6)
    s2, y2 = hip.add_layer_norm(x, rt.contiguous(), None, g, b, 1e-6)
    assert torch.equal(s1, s2) and torch.equal(y1, y2) and torch.allclose(s1, x + rt, atol=1e-6)
    sc = torch.tensor([0.5, 2.0], device="cuda")
    s3, _ = hip.add_layer_norm(x, rt, sc, g, b, 1e-6)
    assert torch.allclose(s3, x + rt * sc.view(B, 1, 1), atol=1e-6)
    for bad in (sc.double(), sc.half(), torch.ones(3, device="cuda"), torch.ones(B, 2, device="cuda")[:, 0], torch.ones(0, device="cuda")):
        with pytest.raises(hip.AadgError):
            hip.add_layer_norm(x, rt, bad, g, b, 1e-6)
