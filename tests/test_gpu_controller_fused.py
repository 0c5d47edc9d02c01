"""GPU: the fused controller kernels (csrc/controller.hip) against the eager Controller / PPO criterion, which are
pinned to the reference by tests/test_host_cpu.py."""
import copy

import pytest
import torch

from helpers import Cfg

pytestmark = pytest.mark.gpu


def _cfg(L=2, num_mags=10, exclude=0):
    cfg = Cfg()
    cfg.CONTROLLER.T, cfg.CONTROLLER.C, cfg.CONTROLLER.PENALTY, cfg.CONTROLLER.LOSS = 2, 2.5, 1e-5, "ppo"
    cfg.CONTROLLER.L, cfg.CONTROLLER.NUM_MAGS, cfg.CONTROLLER.EXCLUDE_OPS_NUM = L, num_mags, exclude
    return cfg


# widths (32, 100) are the module's defaults = the register-resident-weights instantiation of k_ctrl_rollout; any other
# width runs the run-time-width instantiation
# the limits of the one-workgroup-per-sequence kernels (k_ctrl_sample_seq / k_ppo_rollout + k_ppo_grad_adam: S = 2 L <= 8 steps, M Q <= 128
# sequences) and the shapes just beyond them, which run k_ctrl_rollout; the widest heads the library takes (16 actions); one policy
EDGE_SHAPES = [(4, 10, 0, 6, (32, 100)), (2, 10, 0, 25, (32, 100)), (2, 10, 0, 26, (32, 100)), (4, 10, 0, 26, (32, 100)),
               (2, 16, 0, 4, (32, 100)), (1, 3, 5, 1, (32, 100))]


@pytest.mark.parametrize("L,num_mags,exclude,M,widths", [(2, 10, 0, 6, (32, 100)), (1, 5, 2, 3, (32, 100)), (3, 12, 0, 8, (32, 100)),
                                                       (2, 10, 0, 6, (16, 60))] + EDGE_SHAPES)
def test_fused_sample_is_consistent_with_the_module(hip, L, num_mags, exclude, M, widths):
    from aadg_amd.models.controller import Controller
    cfg = _cfg(L, num_mags, exclude)
    torch.manual_seed(11)
    c = Controller(cfg, embedding_dim=widths[0], hidden_dim=widths[1]).cuda()
    with torch.no_grad():
        for p in c.parameters():
            p.mul_(3.0)                       # move away from the near-uniform initial policy
    assert hip.controller_supported(c, M)
    ws = hip.controller_workspace(c, M)
    u = torch.rand(M, c.Q * c.L * 2, device="cuda")
    policies, op_probs, mag_probs, log_probs, entropies = hip.controller_sample(c, M, u, ws)
    assert policies.dtype == torch.int64 and tuple(policies.shape) == (M, c.Q * c.L * 2)
    assert int(policies[:, 0::2].max()) < c.NUM_OPS and int(policies[:, 1::2].max()) < c.NUM_MAGS and int(policies.min()) >= 0
    # teacher-forcing the module on the drawn actions reproduces log-probs, entropies and the mean head probabilities
    with torch.no_grad():
        _, lps, ents, p_op, p_mag = c._rollout(M, forced=policies, want_entropy=True)
    assert torch.allclose(torch.stack(lps, -1).sum(-1), log_probs, atol=2e-5)
    assert torch.allclose(torch.stack(ents, -1).sum(-1), entropies, atol=2e-5)
    assert torch.allclose(torch.stack(p_op, -1).permute(0, 2, 1).reshape(-1, c.NUM_OPS).mean(0), op_probs, atol=1e-6)
    assert torch.allclose(torch.stack(p_mag, -1).permute(0, 2, 1).reshape(-1, c.NUM_MAGS).mean(0), mag_probs, atol=1e-6)
    # same uniforms -> same draw; different uniforms -> (almost surely) a different draw
    again = hip.controller_sample(c, M, u, ws)[0]
    assert torch.equal(again, policies)
    other = hip.controller_sample(c, M, torch.rand_like(u), ws)[0]
    assert not torch.equal(other, policies)


def test_fused_sampling_follows_the_head_distribution(hip):
    from aadg_amd.models.controller import Controller
    cfg = _cfg()
    torch.manual_seed(3)
    c = Controller(cfg).cuda()
    with torch.no_grad():
        c.outop.bias.copy_(torch.linspace(-2, 2, c.NUM_OPS))
    M = 6
    ws = hip.controller_workspace(c, M)
    counts = torch.zeros(c.NUM_OPS, device="cuda")
    n = 400
    for _ in range(n):
        pol = hip.controller_sample(c, M, torch.rand(M, 20, device="cuda"), ws)[0]
        counts += torch.bincount(pol[:, 0], minlength=c.NUM_OPS).float()      # first decision: state-independent
    with torch.no_grad():
        inp, hx, cx = c._fresh_state(1)
        hx, cx = c.lstm(inp, (hx, cx))
        p = torch.softmax(c.C * torch.tanh(c.outop(hx)) / c.T, -1)[0]
    freq = counts / counts.sum()
    assert (freq - p).abs().max().item() < 0.04


@pytest.mark.parametrize("L,num_mags,exclude,M,widths", [(2, 10, 0, 6, (32, 100)), (1, 5, 2, 3, (32, 100)), (3, 12, 0, 8, (32, 100)),
                                                       (2, 10, 0, 6, (16, 60))] + EDGE_SHAPES)
def test_fused_ppo_update_equals_eager(hip, L, num_mags, exclude, M, widths):
    from aadg_amd.models.controller import Controller
    from aadg_amd.models.graphed import FusedControllerStep, make_controller_step
    from aadg_amd import losses
    cfg = _cfg(L, num_mags, exclude)
    torch.manual_seed(5)
    eager = Controller(cfg, embedding_dim=widths[0], hidden_dim=widths[1]).cuda()
    fused_c = copy.deepcopy(eager)
    reward = torch.randn(M, device="cuda")
    opt_f = torch.optim.Adam(fused_c.parameters(), lr=0.00035)
    crit_f = losses.search_loss(cfg)
    f = make_controller_step(fused_c, crit_f, opt_f, M)
    assert isinstance(f, FusedControllerStep)
    opt_e = torch.optim.Adam(eager.parameters(), lr=0.00035)
    crit_e = losses.search_loss(cfg)
    crit_e.register_optimizer(opt_e)
    for rnd in range(3):
        policies, op_probs, mag_probs, log_probs, entropies = f.sample()
        pol = policies.clone()
        lp = eager.evaluate(pol, M)
        assert torch.allclose(lp.detach(), log_probs, atol=2e-5)
        le, se, pe = crit_e(eager, pol, lp, entropies.clone(), reward)
        lf, sf, pf = f.update(reward, entropies)
        assert abs(le.item() - lf.item()) < 2e-5 and abs(pe.item() - pf.item()) < 1e-6
        for (name, a), b in zip(eager.named_parameters(), fused_c.parameters()):
            assert torch.allclose(a, b, atol=3e-6), (rnd, name, (a - b).abs().max().item())
        reward = torch.randn(M, device="cuda") * (rnd + 1)
    # the optimizer state is torch's: same step count and moments as the eager optimizer
    for pe_, pf_ in zip(eager.parameters(), fused_c.parameters()):
        se_, sf_ = opt_e.state[pe_], opt_f.state[pf_]
        assert float(se_['step']) == float(sf_['step']) == 15.0
        assert torch.allclose(se_['exp_avg'], sf_['exp_avg'], rtol=1e-3, atol=1e-6)
        assert torch.allclose(se_['exp_avg_sq'], sf_['exp_avg_sq'], rtol=1e-3, atol=1e-9)
    sd = opt_f.state_dict()
    assert len(sd['state']) == 9


def test_fused_falls_back_for_reinforce(hip):
    from aadg_amd.models.controller import Controller
    from aadg_amd.models.graphed import GraphedControllerStep, make_controller_step
    from aadg_amd import losses
    cfg = _cfg()
    cfg.CONTROLLER.LOSS = "reinforce"
    c = Controller(cfg).cuda()
    step = make_controller_step(c, losses.search_loss(cfg), torch.optim.Adam(c.parameters(), lr=1e-3), 6)
    assert isinstance(step, GraphedControllerStep)
