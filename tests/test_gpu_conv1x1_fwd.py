"""GPU: the matrix-core 1x1 convolution on NCHW (LDS transpose reads, csrc/conv1x1_fwd.hip) vs a float32 reference of the same
bfloat16 operands, forward and as the input gradient (transposed weight)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,M,K,H,W", [(2, 64, 64, 16, 16), (3, 48, 256, 8, 24), (2, 256, 64, 16, 32), (1, 128, 512, 8, 8),
                                       (2, 304, 256, 16, 16), (2, 20, 40, 4, 8), (1, 130, 72, 8, 40), (2, 256, 1024, 8, 16)])
def test_conv1x1_nchw_matches_reference(hip, N, M, K, H, W):
    torch.manual_seed(M + K)
    x = torch.randn(N, K, H, W, device="cuda").to(torch.bfloat16)
    a = (torch.randn(M, K, device="cuda") * 0.1).to(torch.bfloat16)
    # transpose-detecting: make both operands asymmetric along every axis
    x[:, :, 0, :] += 1.0
    a[0] += 0.5
    y = hip.conv1x1_nchw(a, x)
    ref = F.conv2d(x.float(), a.float().view(M, K, 1, 1))
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    err = (y.float() - ref).abs().max().item()
    assert err <= 1e-2 * max(1.0, ref.abs().max().item()), err         # one bfloat16 rounding of the float32 sum


def test_conv1x1_module_routes_and_trains(hip):
    """Conv1x1 on a shape the own kernel takes (64 -> 256) and one the library keeps (512 -> 512): same numbers as F.conv2d."""
    from aadg_amd.models.deeplab import Conv1x1
    torch.manual_seed(3)
    for cin, cout in ((64, 256), (512, 512)):
        m = Conv1x1(cin, cout).cuda()
        x = torch.randn(3, cin, 16, 16, device="cuda").to(torch.bfloat16).requires_grad_(True)
        y = m(x)
        g = torch.randn_like(y)
        y.backward(g)
        xr = x.detach().float().requires_grad_(True)
        wr = m.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
        yr = F.conv2d(xr, wr)
        yr.backward(g.float())
        assert (y.float() - yr).abs().max().item() <= 2e-2 * max(1.0, yr.abs().max().item())
        assert (x.grad.float() - xr.grad).abs().max().item() <= 2e-2 * max(1.0, xr.grad.abs().max().item())
        assert (m.weight.grad - wr.grad).abs().max().item() <= 2e-2 * max(1.0, wr.grad.abs().max().item())


def test_conv1x1_nchw_rejects_unsupported(hip):
    a = torch.zeros(8, 12, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(hip.AadgError):
        hip.conv1x1_nchw(a, torch.zeros(1, 12, 4, 8, device="cuda", dtype=torch.bfloat16))       # K % 8 != 0
    with pytest.raises(hip.AadgError):
        hip.conv1x1_nchw(a.float(), torch.zeros(1, 12, 4, 8, device="cuda"))
