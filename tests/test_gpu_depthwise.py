"""GPU: depthwise 3x3 (dilated) convolution kernels vs torch.nn.functional.conv2d -- forward, input and weight gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,d", [((2, 8, 32, 32), 12), ((2, 8, 32, 32), 36), ((3, 5, 32, 32), 24), ((2, 16, 128, 128), 1),
                                     ((2, 4, 16, 8), 2), ((1, 3, 40, 24), 1), ((2, 6, 9, 256), 3), ((5, 7, 8, 8), 1),
                                     # widths that are no power of two: more rows per thread than the weight gradient preloads (9 > 8),
                                     # strided and walking row order
                                     ((2, 3, 96, 96), 1), ((2, 3, 90, 88), 1), ((2, 2, 96, 96), 2)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_matches_conv2d(hip, shape, d, dtype):
    torch.manual_seed(shape[1] + d)
    N, C, H, W = shape
    x = torch.randn(shape, device="cuda").to(dtype).requires_grad_(True)
    w = (torch.randn(C, 1, 3, 3, device="cuda") * 0.5).requires_grad_(True)
    assert hip.dwconv3x3_supported(x, w, d)
    y = hip.dwconv3x3(x, w, d)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, d, d, C)
    lo = dtype == torch.bfloat16
    assert y.dtype == dtype and y.shape == yr.shape
    assert (y.float() - yr).abs().max().item() <= (5e-2 if lo else 2e-5)
    g = torch.randn(shape, device="cuda").to(dtype)
    y.backward(g)
    yr.backward(g.float())
    assert (x.grad.float() - xr.grad).abs().max().item() <= (5e-2 if lo else 2e-5)
    assert w.grad.dtype == torch.float32
    scale = max(1.0, (N * H * W) ** 0.5)
    assert (w.grad - wr.grad).abs().max().item() <= (2e-2 if lo else 2e-4) * scale


def test_dwconv_unsupported_shapes_are_reported(hip):
    x = torch.randn(1, 4, 8, 12, device="cuda", dtype=torch.bfloat16)      # W not a multiple of 8 elements
    w = torch.randn(4, 1, 3, 3, device="cuda")
    assert not hip.dwconv3x3_supported(x, w, 1)
    with pytest.raises(hip.AadgError):
        hip.dwconv3x3(x, w, 1)
    big = torch.empty(1, 1, 4, 512, device="cuda")
    assert not hip.dwconv3x3_supported(big, torch.randn(1, 1, 3, 3, device="cuda"), 1)


def test_module_falls_back_for_stride_2(hip):
    from aadg_amd.models.deeplab import DepthwiseConv3x3
    m = DepthwiseConv3x3(8, stride=2).cuda()
    x = torch.randn(2, 8, 16, 16, device="cuda")
    assert m(x).shape == (2, 8, 8, 8)
    m1 = DepthwiseConv3x3(8, dilation=2).cuda()
    y = m1(x)
    assert torch.allclose(y, F.conv2d(x, m1.weight, None, 1, 2, 2, 8), atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 16, 128, 128), (3, 5, 40, 64), (2, 6, 70, 32), (3, 4, 33, 128), (19, 3, 64, 64), (1, 2, 1, 32)])
def test_dwconv_weight_gradient_register_window_kernel(hip, shape):
    """k_dw3x3_wgrad_rows (dilation 1, bfloat16, W in {32, 64, 128}): strips that end inside / at the image, the padding columns at both
    row ends (lane groups narrower than a DPP row), several tasks per lane group -- against the float32 gradient of the same tensors"""
    torch.manual_seed(shape[1] + shape[2])
    N, C, H, W = shape
    x = torch.randn(shape, device="cuda").to(torch.bfloat16)
    g = torch.randn(shape, device="cuda").to(torch.bfloat16)
    w = torch.randn(C, 1, 3, 3, device="cuda", requires_grad=True)
    xq = x.clone().requires_grad_(True)
    hip.dwconv3x3(xq, w, 1).backward(g)
    wr = w.detach().clone().requires_grad_(True)
    F.conv2d(x.float(), wr, None, 1, 1, 1, C).backward(g.float())
    assert (w.grad - wr.grad).abs().max().item() <= 2e-4 * wr.grad.abs().max().item() + 1e-4

