"""GPU parity: fused per-policy BCE + Dice (+ gradient) kernel vs the oracle and vs plain torch fp32
(tolerance 1e-4 on the losses, north_star: Dice within 1e-4 fp32; Dice itself is count-exact)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,K,H,W,M", [(12, 2, 32, 32, 6), (6, 1, 40, 36, 3), (18, 2, 17, 19, 6), (8, 2, 256, 256, 2)])
def test_bce_dice_and_grad(hip, oracle, N, K, H, W, M):
    rs = np.random.RandomState(N * H + W)
    z = (rs.randn(N, K, H, W) * 3).astype(np.float32)
    z[0, 0, 0, :4] = [60.0, -60.0, 120.0, -120.0]          # saturated logits: clamped logs / eps-guarded gradient
    y = (rs.rand(N, K, H, W) > 0.6).astype(np.float32)
    zt = torch.from_numpy(z).cuda().requires_grad_(True)
    yt = torch.from_numpy(y).cuda()
    loss, bce, dice = hip.policy_bce_loss(zt, yt, M)
    loss.backward()
    want_bce = oracle.policy_bce(z, y, M)
    assert np.abs(bce.detach().cpu().numpy() - want_bce).max() < 1e-4 * max(1.0, want_bce.max())
    assert np.abs(dice.detach().cpu().numpy() - oracle.dice(z, y)).max() < 1e-6
    # plain torch fp32 reference of the same op (search_dg.py:140-142)
    zr = torch.from_numpy(z).cuda().requires_grad_(True)
    p = torch.sigmoid(zr)
    ref = torch.stack([torch.nn.functional.binary_cross_entropy(p[j::M], yt[j::M]) for j in range(M)]).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))
    assert (zt.grad - zr.grad).abs().max().item() < 1e-6 + 1e-3 * zr.grad.abs().max().item()


def test_full_size_property(hip):
    """BASELINE size: logits == +/-20 exactly on the labels -> Dice 1, BCE ~ 2e-9; flipped -> Dice 0."""
    N, K, S, M = 24, 2, 512, 6
    y = (torch.rand(N, K, S, S, device="cuda") > 0.5).float()
    z = (y * 2 - 1) * 20
    bce, dice, _ = hip.seg_bce_dice(z, y, M)
    assert torch.all(dice == 1.0) and bce.max().item() < 1e-7
    bce, dice, _ = hip.seg_bce_dice(-z, y, M)
    # fp32 sigmoid saturates: sigmoid(20) == 1.0f -> log(1-p) clamps at -100 (nn.BCELoss semantics), so the
    # wrong-side loss is 20 where y=1 and 100 where y=0; compare with torch's own fp32 result
    ref = torch.nn.functional.binary_cross_entropy(torch.sigmoid(-z), y)
    assert torch.all(dice == 0.0) and abs(bce.mean().item() - ref.item()) < 1e-3


@pytest.mark.parametrize("scale,dtype", [(1.0, torch.float32), (0.75, torch.float32), (1.25, torch.bfloat16)])
def test_loss_and_backward_in_one_call(hip, scale, dtype):
    """Round 4: _lib.policy_bce_backward = the loss, the Dice monitor and the backward pass started from the kernel's own gradient
    (grad_scale folded into the kernel: aadg_seg_bce_dice_scaled_f32) -- against `(scale * policy_bce_loss(...)).backward()`, the
    autograd route it replaces, through a small differentiable producer (what the backbone is in the search step), float32 and a
    bfloat16 producer (autocast: the cast node is part of the graph)."""
    torch.manual_seed(3)
    N, K, H, W, M = 12, 2, 24, 40, 6
    x = torch.randn(N, 3, H, W, device="cuda")
    y = (torch.rand(N, K, H, W, device="cuda") > 0.6).float()
    conv = torch.nn.Conv2d(3, K, 3, padding=1).cuda()

    def produce():
        out = conv(x)
        return out.to(dtype) if dtype != torch.float32 else out

    conv.zero_grad()
    loss_a, bce_a, dice_a = hip.policy_bce_loss(produce().float(), y, M)
    (loss_a * scale).backward()
    want = [p.grad.clone() for p in conv.parameters()]
    conv.zero_grad()
    loss_b, bce_b, dice_b = hip.policy_bce_backward(produce(), y, M, scale)
    got = [p.grad for p in conv.parameters()]
    assert abs(loss_b.item() - scale * loss_a.item()) <= 1e-6 * max(1.0, abs(loss_a.item()))
    assert torch.equal(bce_a.detach(), bce_b) and torch.equal(dice_a, dice_b)
    for g, w in zip(got, want):
        assert (g - w).abs().max().item() <= 2e-6 * max(1.0, w.abs().max().item()), (g - w).abs().max().item()
    # no graph behind the logits: the losses still come back, nothing to propagate
    loss_c, _, _ = hip.policy_bce_backward(produce().detach(), y, M, scale)
    assert abs(loss_c.item() - loss_b.item()) < 1e-7


def test_one_launch_accumulators_are_left_zeroed_and_results_deterministic(hip):
    """Round 5: the loss is ONE launch -- integer device-scope atomics into accumulators in the workspace, the last-arriving workgroup
    finalises and zeroes them.  Repeated calls (same and other shapes, sharing the cached workspace of their size) must return
    bit-identical results, and the workspace must be all zero after every call."""
    torch.manual_seed(3)
    for N, K, S, M in ((24, 2, 256, 6), (6, 1, 64, 3), (24, 2, 256, 6), (144, 2, 128, 6)):
        z = torch.randn(N, K, S, S, device="cuda") * 3
        y = (torch.rand(N, K, S, S, device="cuda") > 0.6).float()
        first = None
        for _ in range(3):
            bce, dice, grad = hip.seg_bce_dice(z, y, M, want_grad=True)
            got = (bce.clone(), dice.clone())
            if first is None:
                first = got
            assert torch.equal(got[0], first[0]) and torch.equal(got[1], first[1])
        torch.cuda.synchronize()
        for key, buf in hip._zws_cache.items():
            if key[2] == "segloss":
                assert int(buf.count_nonzero()) == 0, key
        p = torch.sigmoid(z)
        ref = torch.stack([torch.nn.functional.binary_cross_entropy(p[j::M], y[j::M]) for j in range(M)])
        assert (first[0] - ref).abs().max().item() < 1e-4
