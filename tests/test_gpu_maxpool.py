"""GPU: 3x3/2 max-pooling kernels vs torch.nn.functional.max_pool2d (forward bit-exact, backward incl. ties)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (1, 2, 33, 24), (3, 4, 64, 256), (2, 2, 35, 136), (1, 1, 2, 8), (2, 5, 17, 264)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ties", [False, True])
def test_maxpool_matches_torch(hip, shape, dtype, ties):
    torch.manual_seed(shape[2] * 7 + shape[3])
    x = torch.randn(shape, device="cuda")
    if ties:
        x = (x * 1.5).round().clamp_(-1, 1)          # many equal values inside every window
    x = x.to(dtype).requires_grad_(True)
    assert hip.maxpool3x3s2_supported(x)
    y = hip.maxpool3x3s2(x)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    assert y.shape == yr.shape and torch.equal(y, yr)
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-6       # up to 4 gradients are summed per input (bf16 rounding of the sum)
    assert (x.grad.float() - xr.grad.float()).abs().max().item() <= tol


def test_maxpool_without_gradient_skips_the_index(hip):
    x = torch.randn(2, 3, 20, 32, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        y = hip.maxpool3x3s2(x)
    assert torch.equal(y, F.max_pool2d(x, 3, 2, 1))


def test_maxpool_unsupported_width(hip):
    x = torch.randn(1, 1, 8, 12, device="cuda")
    assert not hip.maxpool3x3s2_supported(x)
    with pytest.raises(hip.AadgError):
        hip.maxpool3x3s2(x)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4, 64, 64), (3, 2, 128, 32), (1, 3, 66, 64)])
def test_bn_relu_maxpool_matches_the_three_layers(hip, dtype, shape):
    """One pass (BatchNorm statistics -> normalise + ReLU + 3x3/2 pooling) vs batch_norm -> relu -> max_pool2d: same pooled map,
    running statistics and gradients."""
    torch.manual_seed(sum(shape))
    N, C, H, W = shape
    x = (torch.randn(shape, device="cuda") * 1.5 + 0.3).to(dtype)
    if not hip.bn_relu_maxpool_supported(x):
        pytest.skip("shape outside the fused kernel's domain")
    w = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    b = (torch.randn(C, device="cuda") * 0.3).requires_grad_(True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    x1 = x.clone().requires_grad_(True)
    y = hip.bn_relu_maxpool(x1, w, b, rm, rv, 0.1, 1e-5)
    g = torch.randn_like(y)
    y.backward(g)
    x2 = x.clone().requires_grad_(True)
    w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    t = hip.batch_norm_act(x2, w2, b2, rm2, rv2, True, 0.1, 1e-5, 1)           # the unfused kernels: same arithmetic
    y2 = hip.maxpool3x3s2(t)
    y2.backward(g)
    assert torch.equal(y, y2)
    assert torch.allclose(rm, rm2) and torch.allclose(rv, rv2)
    lo = dtype == torch.bfloat16
    assert (x1.grad.float() - x2.grad.float()).abs().max().item() <= (2e-2 if lo else 1e-5)
    assert (w.grad - w2.grad).abs().max().item() <= (2e-2 if lo else 1e-4) * max(1.0, w2.grad.abs().max().item())
    assert (b.grad - b2.grad).abs().max().item() <= (2e-2 if lo else 1e-4) * max(1.0, b2.grad.abs().max().item())
