"""GPU: depthwise 3x3 + bias + GELU on token-layout activations (csrc/dwconv_nhwc.hip; SegFormer's Mix-FFN, mix_transformer.py:19-46,
149-159) against the plain PyTorch float32 reference of the same op (tokens -> NCHW -> conv2d(groups = C) -> GELU -> tokens) and its
autograd gradients.  bfloat16 tensors: the reference runs in float32 on the same rounded inputs; tolerance one bfloat16 rounding."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (3, 9, 7, 40), (1, 33, 20, 256), (2, 5, 18, 1280)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_gelu_nhwc_matches_float32_reference(hip, B, H, W, C, dtype):
    torch.manual_seed(C + H)
    h = (torch.randn(B, H * W, C, device="cuda") * 1.2).to(dtype)
    weight = (torch.randn(C, 1, 3, 3, device="cuda") * 0.4).requires_grad_(True)
    bias = (torch.randn(C, device="cuda") * 0.2).requires_grad_(True)
    dout = torch.randn(B, H * W, C, device="cuda").to(dtype)
    ha = h.clone().requires_grad_(True)
    out = hip.dwconv3x3_gelu_nhwc(ha, weight, bias, H, W)
    out.backward(dout)
    hr = h.float().clone().requires_grad_(True)
    w2, b2 = weight.detach().clone().requires_grad_(True), bias.detach().clone().requires_grad_(True)
    m = hr.view(B, H, W, C).permute(0, 3, 1, 2)
    ref = F.gelu(F.conv2d(m, w2, b2, padding=1, groups=C)).permute(0, 2, 3, 1).reshape(B, H * W, C)
    ref.backward(dout.float())
    tol = 2e-5 if dtype == torch.float32 else 2 ** -7
    sc = lambda t: max(1.0, float(t.detach().abs().max()))                            # noqa: E731
    assert (out.float() - ref).abs().max().item() <= tol * sc(ref)
    # bfloat16: the kernel rounds the intermediate g = dout * GELU'(z) once more before the transposed convolution
    assert (ha.grad.float() - hr.grad).abs().max().item() <= 3 * tol * sc(hr.grad)
    assert torch.allclose(weight.grad, w2.grad, rtol=1e-2 if dtype == torch.bfloat16 else 1e-4, atol=(2e-2 if dtype == torch.bfloat16 else 1e-3) * sc(w2.grad) / 10)
    assert torch.allclose(bias.grad, b2.grad, rtol=1e-2 if dtype == torch.bfloat16 else 1e-4, atol=(2e-2 if dtype == torch.bfloat16 else 1e-3) * sc(b2.grad) / 10)
