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
