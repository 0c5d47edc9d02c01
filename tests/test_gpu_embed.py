"""GPU: fused embedding prologue (EMA discriminator branch) vs the torch modules."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,C", [(144, 2048), (24, 1280), (5, 128), (3, 77)])
def test_embed_prologue_matches_modules(hip, N, C):
    from aadg_amd.models.discriminator import MomentumFeatureDiscriminator
    torch.manual_seed(N + C)
    d = MomentumFeatureDiscriminator(3, C).cuda()
    with torch.no_grad():
        for p in d.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    x = torch.randn(N, C, device="cuda")
    out, fe = d(x, momentum=True, return_feature=True)
    with torch.no_grad():
        fe_r = d.mom_dis(x.clone())
        out_r = d.mom_fc(fe_r)
    assert fe.shape == (N, 128) and out.shape == (N, 3)
    assert torch.allclose(fe, fe_r, atol=2e-5, rtol=1e-5) and torch.allclose(out, out_r, atol=2e-5, rtol=1e-5)
    assert not fe.requires_grad and not out.requires_grad
    # a strided view (row stride > C) and the fe-only form
    big = torch.randn(N, C + 8, device="cuda")
    _, fe2 = hip.embed_prologue(big[:, :C], d.mom_dis[0].weight, d.mom_dis[0].bias)
    assert torch.allclose(fe2, d.mom_dis(big[:, :C].contiguous()), atol=2e-5, rtol=1e-5)


def test_embed_prologue_rejects_bad_input(hip):
    w = torch.randn(128, 16, device="cuda")
    b = torch.randn(128, device="cuda")
    with pytest.raises(hip.AadgError):
        hip.embed_prologue(torch.randn(4, 16), w, b)
    with pytest.raises(hip.AadgError):
        hip.embed_prologue(torch.randn(4, 16, device="cuda").half(), w, b)


def test_embed_norms_feed_the_sinkhorn_kernel(hip, oracle):
    """SURVEY (f)1: the prologue hands |fe[n]| to the reward kernel (aadg_embed_prologue_norm_f32 ->
    aadg_sinkhorn_rewards_norm_f32).  Same rewards as the two-step path (kernel computes its own norms) and as the oracle,
    <= 1e-5; odd row counts exercise the last, partially filled workgroup of the prologue."""
    import numpy as np
    from aadg_amd.models.discriminator import MomentumFeatureDiscriminator
    for D, B, M, C in ((3, 8, 6, 2048), (8, 2, 6, 512), (3, 3, 5, 1280)):
        N = D * B * M
        torch.manual_seed(N)
        d = MomentumFeatureDiscriminator(D, C).cuda()
        x = torch.randn(N, C, device="cuda") * 0.5 + torch.arange(N, device="cuda")[:, None] % D * 0.3
        out, fe, nrm = d(x, momentum=True, return_feature=True, return_norm=True)
        assert nrm.shape == (N,) and torch.allclose(nrm, fe.norm(dim=1), rtol=1e-6, atol=1e-6)
        out2, fe2 = d(x, momentum=True, return_feature=True)
        assert torch.equal(fe, fe2) and torch.equal(out, out2)
        a = hip.sinkhorn_rewards(fe, D, B, M, row_norm=nrm).cpu().numpy()
        b = hip.sinkhorn_rewards(fe, D, B, M).cpu().numpy()
        want = oracle.sinkhorn_rewards(fe.cpu().numpy(), D, B, M)
        assert np.abs(a - b).max() <= 1e-5 and np.abs(a - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
