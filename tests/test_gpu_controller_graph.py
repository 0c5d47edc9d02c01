"""GPU: the hipGraph-captured controller step performs the same parameter update as the eager criterion
objects (which are pinned to the reference by tests/test_host_cpu.py)."""
import copy

import numpy as np
import pytest
import torch

from helpers import Cfg

pytestmark = pytest.mark.gpu


def _cfg(loss):
    cfg = Cfg()
    cfg.CONTROLLER.T, cfg.CONTROLLER.C, cfg.CONTROLLER.PENALTY, cfg.CONTROLLER.LOSS = 2, 2.5, 1e-5, loss
    return cfg


@pytest.mark.parametrize("loss", ["ppo", "reinforce"])
def test_graphed_update_equals_eager(hip, loss):
    from aadg_amd.models.controller import Controller
    from aadg_amd.models.graphed import GraphedControllerStep
    from aadg_amd import losses
    cfg = _cfg(loss)
    torch.manual_seed(5)
    eager = Controller(cfg).cuda()
    graphed_c = copy.deepcopy(eager)
    M = 6
    reward = torch.tensor([0.3, -1.2, 0.8, 1.1, -0.4, -0.6], device="cuda")
    # graphed: sample (graph 1) then update (graph 2), two rounds to exercise replay
    opt_g = torch.optim.Adam(graphed_c.parameters(), lr=0.00035)
    crit_g = losses.search_loss(cfg)
    g = GraphedControllerStep(graphed_c, crit_g, opt_g, M)
    opt_e = torch.optim.Adam(eager.parameters(), lr=0.00035)
    crit_e = losses.search_loss(cfg)
    crit_e.register_optimizer(opt_e)
    for rnd in range(2):
        policies, op_probs, mag_probs, log_probs, entropies = g.sample()
        assert policies.dtype == torch.int64 and tuple(policies.shape) == (M, 20)
        assert int(policies[:, 0::2].max()) < 10 and int(policies[:, 1::2].max()) < 10
        pol = policies.clone()
        # eager twin: same actions (teacher-forced), same reward
        if loss == "ppo":
            lp = eager.evaluate(pol, M)
            ent = entropies.clone()
        else:
            _, lps, ents, _, _ = eager._rollout(M, forced=pol, want_entropy=True)
            lp, ent = torch.stack(lps, -1).sum(-1), torch.stack(ents, -1).sum(-1)
        assert torch.allclose(lp.detach(), log_probs, atol=1e-5)          # evaluate == sample log-prob
        le, se, pe = crit_e(eager, pol, lp, ent, reward)
        lg, sg, pg = g.update(reward, entropies)
        assert abs(le.item() - lg.item()) < 1e-5 and abs(se.item() - sg.item()) < 1e-5
        for a, b in zip(eager.parameters(), graphed_c.parameters()):
            assert torch.allclose(a, b, atol=2e-6), (rnd, (a - b).abs().max().item())
    # replays draw fresh actions
    p1 = g.sample()[0].clone()
    p2 = g.sample()[0].clone()
    assert not torch.equal(p1, p2)
