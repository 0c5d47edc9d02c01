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
