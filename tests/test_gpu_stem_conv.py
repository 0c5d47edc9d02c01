"""GPU: the MFMA stem convolution (7x7 / stride 2, 3 -> 64 channels, bfloat16) vs torch.nn.functional.conv2d on the same
bfloat16-rounded operands (float32 accumulation on both sides)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W", [(2, 32, 32), (1, 64, 144), (3, 18, 16), (2, 128, 256), (1, 2, 16)])
def test_stem_forward_matches_conv2d(hip, N, H, W):
    torch.manual_seed(H + W)
    x = torch.randn(N, 3, H, W, device="cuda").to(torch.bfloat16)
    w = torch.randn(64, 3, 7, 7, device="cuda") * 0.1
    assert hip.stem_conv7x7_supported(x, w)
    y = hip.stem_conv7x7(x, w)
    ref = F.conv2d(x.float(), w.to(torch.bfloat16).float(), stride=2, padding=3)      # exact products, float32 sums
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    err = (y.float() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err                       # one bfloat16 rounding of the output
    # transpose / tap-order detecting: a single impulse reproduces the (flipped) kernel of every channel
    x0 = torch.zeros(1, 3, 32, 32, device="cuda", dtype=torch.bfloat16)
    x0[0, 1, 16, 16] = 1.0
    y0 = hip.stem_conv7x7(x0, w)
    r0 = F.conv2d(x0.float(), w.to(torch.bfloat16).float(), stride=2, padding=3)
    assert torch.equal(y0.float(), r0.to(torch.bfloat16).float())


def test_stem_module_trains_like_conv2d(hip):
    from aadg_amd.models.deeplab import StemConv7x7
    torch.manual_seed(5)
    m = StemConv7x7().cuda()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    calls = {"n": 0}
    real = hip.stem_conv7x7

    def counted(a, b):
        calls["n"] += 1
        return real(a, b)
    hip.stem_conv7x7 = counted
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
    finally:
        hip.stem_conv7x7 = real
    assert calls["n"] == 1 and y.dtype == torch.bfloat16
    g = torch.randn_like(y)
    y.backward(g)
    wr = m.weight.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yr = F.conv2d(x, wr, stride=2, padding=3)
    yr.backward(g)
    assert (y.float() - yr.float()).abs().max().item() <= 3e-2 * max(1.0, yr.float().abs().max().item())
    assert (m.weight.grad - wr.grad).abs().max().item() <= 2e-2 * max(1.0, wr.grad.abs().max().item())
    # float32 without autocast: the module's own path
    y32 = m(x)
    assert y32.dtype == torch.float32 and calls["n"] == 1


def test_stem_rejects_unsupported(hip):
    w = torch.randn(64, 3, 7, 7, device="cuda")
    assert not hip.stem_conv7x7_supported(torch.zeros(1, 3, 32, 24, device="cuda", dtype=torch.bfloat16), w)
    assert hip.stem_conv7x7_supported(torch.zeros(1, 3, 32, 32, device="cuda"), w)              # float32 images are rounded on load
    assert not hip.stem_conv7x7_supported(torch.zeros(1, 3, 32, 32, device="cuda", dtype=torch.float16), w)
    with pytest.raises(hip.AadgError):
        hip.stem_conv7x7(torch.zeros(1, 3, 32, 24, device="cuda", dtype=torch.bfloat16), w)


@pytest.mark.parametrize("N,H,W", [(2, 32, 32), (3, 18, 48), (2, 128, 256), (1, 2, 16)])
def test_stem_weight_gradient_matches_conv2d(hip, N, H, W):
    torch.manual_seed(H * 3 + W)
    x = torch.randn(N, 3, H, W, device="cuda").to(torch.bfloat16)
    w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.1).requires_grad_(True)
    y = hip.stem_conv7x7(x, w)
    g = torch.randn_like(y)
    y.backward(g)
    wr = w.detach().clone().requires_grad_(True)
    F.conv2d(x.float(), wr, stride=2, padding=3).backward(g.float())          # exact products of the same bf16 operands, float32 sums
    scale = max(1.0, wr.grad.abs().max().item())
    assert (w.grad - wr.grad).abs().max().item() <= 2e-3 * scale


def test_stem_float32_image_is_rounded_on_load(hip):
    torch.manual_seed(2)
    x = torch.randn(2, 3, 64, 96, device="cuda")
    w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.1).requires_grad_(True)
    y32 = hip.stem_conv7x7(x, w)
    g = torch.randn_like(y32)
    y32.backward(g)
    g32 = w.grad.clone()
    w.grad = None
    y16 = hip.stem_conv7x7(x.to(torch.bfloat16), w)
    y16.backward(g)
    assert y32.dtype == torch.bfloat16 and torch.equal(y32, y16)
    assert (g32 - w.grad).abs().max().item() <= 1e-4 * max(1.0, w.grad.abs().max().item())     # same products, atomics reorder the sums
