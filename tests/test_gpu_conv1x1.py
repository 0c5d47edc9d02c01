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


def test_tracked_shadow_backward_after_weight_change_fails_loudly(hip):
    """the bfloat16 shadows of tracked weights are rewritten in place by the next refresh.  A backward that runs after the
    weights changed (optimizer step, or invalidate_weight_shadows() after a `.data` write) must not silently use the new weights."""
    from aadg_amd import _lib
    from aadg_amd.models.deeplab import Conv1x1
    torch.manual_seed(3)
    net = torch.nn.Sequential(Conv1x1(64, 64)).cuda()
    assert _lib.track_bf16_weights(net, (Conv1x1,)) == 1
    x = torch.randn(2, 64, 64, 64, device="cuda").bfloat16().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    y.float().sum().backward()                                  # the normal order works
    w0 = net[0].weight.detach().clone()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    net[0].weight.data.mul_(0.5)                                # a write the version counter does not see ...
    _lib.invalidate_weight_shadows()                            # ... announced by the caller
    with pytest.raises(_lib.AadgError):
        y.float().sum().backward()
    with torch.autocast("cuda", dtype=torch.bfloat16):          # the next forward rebuilds the shadow from the new master
        y2 = net(x)
    ref = F.conv2d(x.float(), (w0 * 0.5).bfloat16().float())
    assert (y2.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


def test_foreign_optimizer_steps_do_not_invalidate_shadows(hip):
    """The discriminator's and the controller's optimizers step between the model's forward and backward.  Ownership of a tracked
    weight is decided on the objects: neither a recycled optimizer id nor a parameter whose id once belonged to a tracked weight
    (a stale table entry whose weak reference is dead) may turn a foreign optimizer's step into an invalidation."""
    import gc
    from aadg_amd import _lib
    from aadg_amd.models.deeplab import Conv1x1
    torch.manual_seed(4)
    net = torch.nn.Sequential(Conv1x1(64, 64)).cuda()
    assert _lib.track_bf16_weights(net, (Conv1x1,)) == 1
    x = torch.randn(2, 64, 32, 32, device="cuda").bfloat16().requires_grad_(True)
    other = torch.nn.Linear(8, 8).cuda()
    # a stale entry under the id of a foreign parameter, as left behind by a dead tracked weight whose id was reused
    dead = torch.nn.Parameter(torch.zeros(1, device="cuda"))
    stale = _lib._Shadow()
    stale.ref = __import__("weakref").ref(dead)
    del dead
    gc.collect()
    _lib._SHADOWS[id(other.weight)] = stale
    try:
        for _ in range(3):                                      # fresh optimizer objects: ids get recycled
            opt = torch.optim.Adam(other.parameters(), lr=1e-3)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x)
            other(torch.randn(4, 8, device="cuda")).sum().backward()
            opt.step()                                          # between the tracked model's forward and backward
            y.float().sum().backward()                          # must not raise
            del opt
            gc.collect()
        own = torch.optim.SGD(net.parameters(), lr=0.1)         # the owner's step does invalidate
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        net[0].weight.grad = torch.zeros_like(net[0].weight)
        own.step()
        with pytest.raises(_lib.AadgError):
            y.float().sum().backward()
    finally:
        _lib._SHADOWS.pop(id(other.weight), None)
