"""GPU: stride-2 sub-sampling kernels (the front half of a stride-2 1x1 convolution) vs slicing, and Conv1x1(stride=2) built on
them vs torch.nn.Conv2d."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 3, 8, 16), (1, 5, 34, 48), (3, 2, 128, 128), (2, 4, 2, 32)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_subsample_matches_slicing(hip, shape, dtype):
    torch.manual_seed(shape[2] + shape[3])
    x = torch.randn(shape, device="cuda").to(dtype).requires_grad_(True)
    assert hip.subsample2x2_supported(x)
    y = hip.subsample2x2(x)
    xr = x.detach().clone().requires_grad_(True)
    yr = xr[:, :, ::2, ::2]
    assert y.is_contiguous() and torch.equal(y, yr)
    g = torch.randn_like(y)
    y.backward(g)
    yr.backward(g)
    assert torch.equal(x.grad, xr.grad)                   # every element written: the gradient, or an exact zero


def test_subsample_unsupported_shapes(hip):
    assert not hip.subsample2x2_supported(torch.zeros(1, 1, 7, 16, device="cuda"))
    assert not hip.subsample2x2_supported(torch.zeros(1, 1, 8, 8, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(hip.AadgError):
        hip.subsample2x2(torch.zeros(1, 1, 7, 16, device="cuda"))


def test_conv1x1_stride2_matches_conv2d(hip):
    from aadg_amd.models.deeplab import Conv1x1
    torch.manual_seed(3)
    m = Conv1x1(64, 128, stride=2).cuda()
    x = torch.randn(4, 64, 32, 64, device="cuda", dtype=torch.bfloat16).requires_grad_(True)
    calls = {"n": 0}
    real = hip.subsample2x2

    def counted(t):
        calls["n"] += 1
        return real(t)
    hip.subsample2x2 = counted
    try:
        y = m(x)
    finally:
        hip.subsample2x2 = real
    assert calls["n"] == 1 and y.shape == (4, 128, 16, 32)
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = m.weight.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=2)
    yr.backward(g.float())
    assert (y.float() - yr).abs().max().item() <= 0.05 * max(1.0, yr.abs().max().item())
    assert (x.grad.float() - xr.grad).abs().max().item() <= 0.05 * max(1.0, xr.grad.abs().max().item())
    assert (m.weight.grad - wr.grad).abs().max().item() <= 0.02 * max(1.0, wr.grad.abs().max().item())
    odd = x.grad[:, :, 1::2, :]
    assert odd.abs().max().item() == 0.0 and x.grad[:, :, :, 1::2].abs().max().item() == 0.0
