"""GPU: MFMA weight-gradient kernel of the 1x1 convolutions vs a float32 reference, and the module wired to it."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Co,Ci,H,W", [(3, 64, 64, 16, 16), (2, 256, 64, 16, 8), (2, 64, 256, 8, 8), (5, 128, 192, 16, 16),
                                          (2, 48, 256, 32, 32), (3, 256, 304, 8, 16), (1, 2048, 512, 8, 8), (7, 20, 36, 8, 8),
                                          (4, 130, 70, 16, 4), (3, 512, 1024, 16, 8), (2, 768, 512, 8, 8)])
def test_wgrad_matches_float32_reference(hip, N, Co, Ci, H, W):
    torch.manual_seed(Co + Ci)
    dy = torch.randn(N, Co, H, W, device="cuda").bfloat16()
    x = (torch.randn(N, Ci, H, W, device="cuda") + torch.arange(Ci, device="cuda").view(1, Ci, 1, 1) * 0.01).bfloat16()
    dw = hip.conv1x1_wgrad(dy, x)
    ref = torch.einsum("nok,nck->oc", dy.float().view(N, Co, -1), x.float().view(N, Ci, -1))
    assert dw.shape == (Co, Ci) and dw.dtype == torch.float32
    assert (dw - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())      # float32 accumulation of exact products


def test_wgrad_rejects_unsupported(hip):
    dy = torch.randn(1, 8, 4, 4, device="cuda").bfloat16()          # HW = 16 < 64
    with pytest.raises(hip.AadgError):
        hip.conv1x1_wgrad(dy, dy)
    with pytest.raises(hip.AadgError):
        hip.conv1x1_wgrad(dy.float(), dy.float())


def test_module_matches_conv2d_under_autocast(hip):
    from aadg_amd.models.deeplab import Conv1x1
    torch.manual_seed(2)
    m = Conv1x1(96, 160).cuda()
    x = torch.randn(4, 96, 16, 16, device="cuda").bfloat16().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    assert y.dtype == torch.bfloat16
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().clone().requires_grad_(True)
    wr = m.weight.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yr = F.conv2d(xr, wr)
    yr.backward(g)
    # forward / input gradient may run on the matrix-core kernel (csrc/conv1x1_fwd.hip) instead of the library GEMM: same
    # bfloat16 operands, float32 accumulation in a different order, one bfloat16 rounding of the result
    assert (y.float() - yr.float()).abs().max().item() <= 2e-2 * max(1.0, yr.float().abs().max().item())
    assert (x.grad.float() - xr.grad.float()).abs().max().item() <= 2e-2 * max(1.0, xr.grad.float().abs().max().item())
    assert m.weight.grad.dtype == torch.float32
    assert (m.weight.grad - wr.grad).abs().max().item() <= 2e-2 * wr.grad.abs().max().item()   # the reference rounds dW to bf16
    # float32 activations (no autocast) keep the library path
    xf = torch.randn(2, 96, 16, 16, device="cuda")
    assert torch.allclose(m(xf), F.conv2d(xf, m.weight), atol=1e-4)


def test_tracked_shadows_follow_the_weights_and_guard_late_backwards(hip):
    """The bfloat16 shadows of tracked weights are rebuilt IN PLACE by every forward of their model (round 4: unconditionally -- no
    version / epoch bookkeeping, no optimizer hooks).  So (1) whatever changed a master weight -- a `.data` write, torch's fused Adam,
    neither of which bumps the version counter -- the next forward sees it; (2) an optimizer step BETWEEN a forward and its backward is
    harmless: the backward reads the buffers as the forward left them; (3) a backward that runs after a LATER forward of the model
    must not silently use that forward's weights: it fails loudly."""
    from aadg_amd import _lib
    from aadg_amd.models.deeplab import Conv1x1
    torch.manual_seed(3)
    net = torch.nn.Sequential(Conv1x1(64, 64)).cuda()
    assert _lib.track_bf16_weights(net, (Conv1x1,)) == 1
    x = torch.randn(2, 64, 64, 64, device="cuda").bfloat16().requires_grad_(True)

    def fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return net(x)

    def ref_out(w):
        return F.conv2d(x.detach().float(), w.bfloat16().float())

    fwd().float().sum().backward()                                  # the normal order works
    # (1) changes the version counter does not see
    w0 = net[0].weight.detach().clone()
    net[0].weight.data.mul_(0.5)
    y = fwd()
    assert (y.float() - ref_out(w0 * 0.5)).abs().max().item() <= 2e-2 * max(1.0, ref_out(w0 * 0.5).abs().max().item())
    # (2) a fused optimizer step between forward and backward: the input gradient is that of the forward's weights
    w1 = net[0].weight.detach().clone()
    opt = torch.optim.Adam(net.parameters(), lr=0.1, fused=True)
    x.grad = None
    net.zero_grad()
    y = fwd()
    net[0].weight.grad = torch.ones_like(net[0].weight)
    opt.step()
    assert not torch.equal(net[0].weight.detach(), w1)
    net.zero_grad()
    y.float().sum().backward()
    want_dx = torch.nn.functional.conv_transpose2d(torch.ones_like(y).float(), w1.bfloat16().float())
    assert (x.grad.float() - want_dx).abs().max().item() <= 2e-2 * max(1.0, want_dx.abs().max().item())
    y3 = fwd()                                                      # ... and the next forward runs on the stepped weights
    w2 = net[0].weight.detach()
    assert (y3.float() - ref_out(w2)).abs().max().item() <= 2e-2 * max(1.0, ref_out(w2).abs().max().item())
    # (3) two forwards, then the backward of the first
    ya = fwd()
    yb = fwd()
    with pytest.raises(_lib.AadgError):
        ya.float().sum().backward()
    yb.float().sum().backward()                                     # the latest forward's backward is served
    # a foreign model's optimizer steps and forwards are nobody's business
    other = torch.nn.Sequential(Conv1x1(64, 64)).cuda()
    assert _lib.track_bf16_weights(other, (Conv1x1,)) == 1
    yc = fwd()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        other(x.detach()).float().sum().backward()
    torch.optim.SGD(other.parameters(), lr=0.1).step()
    yc.float().sum().backward()

