"""GPU: the MFMA weight gradient of the bottleneck 3x3 convolutions (csrc/conv3x3_wgrad.hip) against the float32 convolution
backward of the same bfloat16 tensors, and the Conv3x3 module against nn.Conv2d."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_wgrad(dy, x, w_shape, d):
    w = torch.zeros(w_shape, device=x.device, dtype=torch.float32)
    return torch.ops.aten.convolution_backward(dy.float(), x.float(), w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1,
                                               [False, True, False])[1]


@pytest.mark.parametrize("N,Co,Ci,H,W,d", [
    (2, 64, 64, 32, 32, 1), (3, 64, 64, 32, 32, 2),        # layer3 / layer4 row width, both dilations
    (2, 128, 64, 64, 64, 1), (1, 64, 128, 64, 64, 2),      # several tiles along either channel axis
    (2, 64, 64, 128, 128, 1),                              # layer1 row width
    (5, 96, 40, 20, 32, 1), (4, 40, 96, 7, 64, 2),         # channel counts that are not multiples of the tile, odd heights, H < 2 D + 2
    (37, 64, 64, 32, 32, 1),                               # long reduction: several images per workgroup, slices that start mid-image
])
def test_wgrad_matches_float32_backward(hip, N, Co, Ci, H, W, d):
    torch.manual_seed(N * 1000 + Co + H)
    x = torch.randn(N, Ci, H, W, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, Co, H, W, device="cuda").to(torch.bfloat16)
    assert hip.load().aadg_conv3x3_wgrad_supported(Co, Ci, H, W, d) == 1
    got = hip.conv3x3_wgrad(dy, x, d)
    want = _ref_wgrad(dy, x, (Co, Ci, 3, 3), d)
    assert got.shape == want.shape and got.dtype == torch.float32
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-4 * scale, ((got - want).abs().max().item(), scale)


def test_wgrad_rejects_what_it_does_not_cover(hip):
    lib = hip.load()
    assert lib.aadg_conv3x3_wgrad_supported(64, 64, 32, 48, 1) == 0 and lib.aadg_conv3x3_wgrad_supported(64, 64, 32, 32, 3) == 0
    x = torch.randn(1, 8, 16, 48, device="cuda").to(torch.bfloat16)
    with pytest.raises(hip.AadgError):
        hip.conv3x3_wgrad(x, x, 1)


@pytest.mark.parametrize("d", [1, 2])
def test_module_equals_conv2d(hip, d):
    """Conv3x3 (own forward, input gradient and weight gradient kernels) against nn.Conv2d (the library) under bfloat16 autocast."""
    from aadg_amd.models.deeplab import Conv3x3
    torch.manual_seed(3)
    ours = Conv3x3(64, 64, 1, d).cuda()
    ref = torch.nn.Conv2d(64, 64, 3, padding=d, dilation=d, bias=False).cuda()
    ref.load_state_dict(ours.state_dict())
    x1 = torch.randn(4, 64, 32, 32, device="cuda").to(torch.bfloat16).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    g = torch.randn(4, 64, 32, 32, device="cuda").to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1, y2 = ours(x1), ref(x2)
    # yardstick: float32 convolution of the same bfloat16 tensors (the library may run a Winograd variant here: looser)
    wb = ours.weight.detach().to(torch.bfloat16).float()
    xr = x1.detach().float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wb, padding=d, dilation=d)
    yr.backward(g.float())
    tol, loose = 2.0 ** -8, 2.0 ** -5
    assert (y1.float() - yr).abs().max().item() <= tol * yr.abs().max().item()
    assert (y2.float() - yr).abs().max().item() <= loose * yr.abs().max().item()
    y1.backward(g); y2.backward(g)
    assert (x1.grad.float() - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
    assert (x2.grad.float() - xr.grad).abs().max().item() <= loose * xr.grad.abs().max().item()
    gw, rw = ours.weight.grad, ref.weight.grad
    assert gw.dtype == torch.float32 and gw.shape == rw.shape
    # the library rounds its weight gradient to bfloat16 before the float32 master copy sees it; ours stays float32
    assert (gw - rw).abs().max().item() <= 1e-2 * rw.abs().max().item()
    want = _ref_wgrad(g, x1.detach(), tuple(gw.shape), d)
    assert (gw - want).abs().max().item() <= 2e-4 * want.abs().max().item()


@pytest.mark.parametrize("N,M,K,H,W,d", [
    (2, 64, 64, 32, 32, 1), (2, 64, 64, 32, 32, 2), (1, 128, 64, 64, 64, 1), (1, 64, 128, 64, 64, 2),
    (2, 64, 64, 128, 128, 1), (1, 32, 64, 10, 128, 2),                             # layer1 row width
    (3, 96, 40, 20, 32, 1), (2, 40, 24, 7, 64, 2), (9, 64, 16, 32, 32, 1),       # ragged channels / heights, more pixel tiles than a multiple of 8
])
def test_forward_kernel_matches_float32_convolution(hip, N, M, K, H, W, d):
    torch.manual_seed(M + K + H)
    x = torch.randn(N, K, H, W, device="cuda").to(torch.bfloat16)
    w = (torch.randn(M, K, 3, 3, device="cuda") / (3.0 * K ** 0.5)).to(torch.bfloat16)
    assert hip.load().aadg_conv3x3_nchw_supported(M, K, H, W, d) == 1
    got = hip.conv3x3_nchw(w.permute(2, 3, 0, 1).reshape(9, M, K).contiguous(), x, d)
    want = torch.nn.functional.conv2d(x.float(), w.float(), padding=d, dilation=d)
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    # float32 accumulation, one rounding to bfloat16 at the end
    assert (got.float() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() + 1e-6


@pytest.mark.parametrize("d,W", [(1, 32), (2, 32), (1, 64)])
def test_autograd_function_all_three_kernels(hip, d, W):
    """_Conv3x3 with the own forward, input gradient (the same kernel on dy with mirrored taps) and weight gradient against float32
    autograd of the same bfloat16 tensors."""
    torch.manual_seed(7)
    Co, Ci, N, H = 64, 128, 3, 32
    x = torch.randn(N, Ci, H, W, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, device="cuda") / (3.0 * Ci ** 0.5)).requires_grad_(True)
    g = torch.randn(N, Co, H, W, device="cuda").to(torch.bfloat16)
    y = hip.conv3x3(x, w, d)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, padding=d, dilation=d)
    yr.backward(g.float())
    assert (y.float() - yr).abs().max().item() <= 2.0 ** -8 * yr.abs().max().item()
    assert (x.grad.float() - xr.grad).abs().max().item() <= 2.0 ** -8 * xr.grad.abs().max().item()
    assert (w.grad - wr.grad).abs().max().item() <= 2e-4 * wr.grad.abs().max().item()


@pytest.mark.parametrize("N,Co,Ci,Ho,Wo", [(2, 64, 64, 32, 32), (2, 128, 64, 64, 64), (3, 96, 40, 9, 32), (19, 64, 64, 16, 32)])
def test_stride2_wgrad_matches_float32_backward(hip, N, Co, Ci, Ho, Wo):
    torch.manual_seed(Co + Ho)
    x = torch.randn(N, Ci, 2 * Ho, 2 * Wo, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, Co, Ho, Wo, device="cuda").to(torch.bfloat16)
    got = hip.conv3x3s2_wgrad(dy, x)
    w = torch.zeros(Co, Ci, 3, 3, device="cuda")
    want = torch.ops.aten.convolution_backward(dy.float(), x.float(), w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-4 * want.abs().max().item()


def test_stride2_module(hip):
    from aadg_amd.models.deeplab import Conv3x3
    torch.manual_seed(5)
    ours = Conv3x3(64, 64, 2, 1).cuda()
    x = torch.randn(3, 64, 64, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(3, 64, 32, 32, device="cuda").to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ours(x)
    assert y.shape == (3, 64, 32, 32) and type(y.grad_fn).__name__ == "_Conv3x3S2Backward"
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = ours.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, stride=2, padding=1)
    yr.backward(g.float())
    assert (y.float() - yr).abs().max().item() <= 2.0 ** -5 * yr.abs().max().item()
    assert (ours.weight.grad - wr.grad).abs().max().item() <= 2e-4 * wr.grad.abs().max().item()
    assert (x.grad.float() - xr.grad).abs().max().item() <= 2.0 ** -8 * xr.grad.abs().max().item()


@pytest.mark.parametrize("N,C,M,Ho,Wo", [(2, 64, 64, 32, 32), (2, 128, 128, 64, 64), (3, 40, 96, 9, 32), (19, 100, 24, 7, 64),
                                         (1, 256, 256, 32, 32)])
def test_stride2_dgrad_matches_float32_backward(hip, N, C, M, Ho, Wo):
    """k_dgrad3x3_s2: the four parity classes (1 + 2 + 2 + 4 taps), ragged channel / row counts, the zero column at j + 1 = Wo and
    the zero row at i + 1 = Ho"""
    torch.manual_seed(C + Ho)
    dy = torch.randn(N, M, Ho, Wo, device="cuda").to(torch.bfloat16)
    w = (torch.randn(M, C, 3, 3, device="cuda") / 8).to(torch.bfloat16)
    assert hip.load().aadg_conv3x3s2_dgrad_supported(C, M, Ho, Wo) == 1
    got = hip.conv3x3s2_dgrad(w.permute(2, 3, 1, 0).reshape(9, C, M).contiguous(), dy)
    x = torch.zeros(N, C, 2 * Ho, 2 * Wo, device="cuda")
    want = torch.ops.aten.convolution_backward(dy.float(), x, w.float(), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert (got.float() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item()
    lib = hip.load()
    assert lib.aadg_conv3x3s2_dgrad_supported(64, 60, 32, 32) == 0 and lib.aadg_conv3x3s2_dgrad_supported(64, 64, 32, 48) == 0


@pytest.mark.parametrize("N,M,K,Ho,Wo", [(2, 64, 64, 32, 32), (2, 128, 128, 64, 64), (3, 96, 40, 9, 32), (19, 24, 104, 7, 64),
                                         (1, 256, 256, 32, 32)])
def test_stride2_forward_matches_float32_convolution(hip, N, M, K, Ho, Wo):
    """k_conv3x3_s2: even / odd column planes, the padding column at 2x - 1 = -1 and the padding row at 2y - 1 = -1, ragged channel
    and row counts"""
    torch.manual_seed(M + Ho)
    x = torch.randn(N, K, 2 * Ho, 2 * Wo, device="cuda").to(torch.bfloat16)
    w = (torch.randn(M, K, 3, 3, device="cuda") / 8).to(torch.bfloat16)
    assert hip.load().aadg_conv3x3s2_nchw_supported(M, K, Ho, Wo) == 1
    got = hip.conv3x3s2_nchw(w.permute(2, 3, 0, 1).reshape(9, M, K).contiguous(), x)
    want = torch.nn.functional.conv2d(x.float(), w.float(), stride=2, padding=1)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert (got.float() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item()
    lib = hip.load()
    assert lib.aadg_conv3x3s2_nchw_supported(64, 60, 32, 32) == 0 and lib.aadg_conv3x3s2_nchw_supported(64, 64, 32, 48) == 0
