"""GPU: HIP bilinear (align_corners=True) up-sampling vs torch.nn.functional.interpolate, forward and backward."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,size,dtype", [((3, 5, 8, 8), (32, 32), torch.float32), ((2, 7, 9, 13), (33, 50), torch.float32),
                                              ((4, 16, 32, 32), (128, 128), torch.bfloat16), ((2, 2, 128, 128), (512, 512), torch.float32),
                                              ((1, 3, 1, 1), (16, 16), torch.float32)])
def test_upsample_matches_torch(hip, shape, size, dtype):
    torch.manual_seed(0)
    x = torch.randn(shape, device="cuda").to(dtype).requires_grad_(True)
    y = hip.upsample_bilinear_ac(x, size)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.interpolate(xr, size=size, mode="bilinear", align_corners=True)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert y.dtype == dtype and y.shape == yr.shape
    assert (y.float() - yr.float()).abs().max().item() <= tol
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    assert (x.grad.float() - xr.grad.float()).abs().max().item() <= tol * 50


def test_deeplab_uses_the_kernel_and_matches_interpolate(hip):
    from aadg_amd.models import deeplab
    torch.manual_seed(1)
    m = deeplab.DeepLabV3Plus("mobilenet_v2", 2).cuda().eval()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    with torch.no_grad():
        y1, f1 = m(x)
        saved = deeplab._upsample_ac
        deeplab._upsample_ac = lambda t, size: F.interpolate(t, size=size, mode="bilinear", align_corners=True)
        try:
            y2, f2 = m(x)
        finally:
            deeplab._upsample_ac = saved
    assert torch.allclose(y1, y2, atol=1e-5) and torch.equal(f1, f2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_backward_reads_a_channel_slice_in_place(hip, dtype):
    """The decoder concatenates the up-sampled map with the skip branch: the gradient reaching the up-sampling is a channel
    slice of the concatenation's gradient and is consumed without a contiguous copy."""
    torch.manual_seed(2)
    x = torch.randn(3, 6, 8, 8, device="cuda").to(dtype).requires_grad_(True)
    other = torch.randn(3, 5, 32, 32, device="cuda").to(dtype).requires_grad_(True)
    g = torch.randn(3, 11, 32, 32, device="cuda").to(dtype)
    torch.cat([hip.upsample_bilinear_ac(x, (32, 32)), other], dim=1).backward(g)
    xr = x.detach().clone().requires_grad_(True)
    torch.cat([F.interpolate(xr, size=(32, 32), mode="bilinear", align_corners=True), other.detach()], dim=1).backward(g)
    tol = 1e-4 if dtype == torch.float32 else 0.15
    assert (x.grad.float() - xr.grad.float()).abs().max().item() <= tol
    assert torch.equal(other.grad, g[:, 6:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape_a,shape_b", [((3, 6, 8, 8), (3, 5, 32, 32)), ((2, 16, 32, 32), (2, 48, 128, 128)), ((1, 3, 5, 7), (1, 2, 9, 20))])
def test_upsample_cat_matches_cat_of_interpolate(hip, dtype, shape_a, shape_b):
    torch.manual_seed(6)
    a = torch.randn(shape_a, device="cuda").to(dtype).requires_grad_(True)
    b = torch.randn(shape_b, device="cuda").to(dtype).requires_grad_(True)
    y = hip.upsample_cat(a, b)
    ar, br = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = torch.cat([F.interpolate(ar, size=shape_b[2:], mode="bilinear", align_corners=True), br], dim=1)
    lo = dtype == torch.bfloat16
    assert y.shape == yr.shape and (y.float() - yr.float()).abs().max().item() <= (2e-2 if lo else 1e-5)
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    assert (a.grad.float() - ar.grad.float()).abs().max().item() <= (0.2 if lo else 5e-4)
    assert torch.equal(b.grad, br.grad)


def test_upsample_cat_with_second_half_already_in_place(hip):
    """The decoder concatenation with both halves produced in place: b is written into its slice of the buffer by its producer
    (here a BatchNorm with out=), the up-sampling kernel fills the other slice -- same result and gradients as the copying path."""
    torch.manual_seed(8)
    dtype = torch.bfloat16
    a0 = torch.randn(2, 16, 8, 8, device="cuda").to(dtype)
    t0 = torch.randn(2, 8, 32, 32, device="cuda").to(dtype)
    w0, b0 = torch.rand(8, device="cuda") + 0.5, torch.randn(8, device="cuda") * 0.2
    g = torch.randn(2, 24, 32, 32, device="cuda").to(dtype)
    res = []
    for inplace in (True, False):
        a, t = a0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
        rm, rv = torch.zeros(8, device="cuda"), torch.ones(8, device="cuda")
        if inplace:
            buf, parts = hip.concat_slices(2, [16, 8], 32, 32, dtype, a.device)
            s = hip.batch_norm_act(t, w0, b0, rm, rv, True, 0.1, 1e-5, 1, out=parts[1])
            y = hip.upsample_cat(a, s, buf)
        else:
            s = hip.batch_norm_act(t, w0, b0, rm, rv, True, 0.1, 1e-5, 1)
            y = hip.upsample_cat(a, s)
        y.backward(g)
        res.append((y.detach().clone(), a.grad.clone(), t.grad.clone()))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    with pytest.raises(hip.AadgError):
        hip.upsample_cat(a0, t0, torch.empty(2, 24, 32, 32, device="cuda", dtype=dtype))       # t0 is not part of that buffer
