"""Event times of the fused controller calls (sample, 5-epoch PPO update) for the library at AADG_LIB_PATH (default: the built one).
    python scripts/ubench/ctrl_time.py [repeats]          (A/B: AADG_LIB_PATH=exp_libs/<tag>.so, scripts/ab/build_variant.py)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import torch
from aadg_amd.config.defaults import get_default_config
from aadg_amd.losses import search_loss
from aadg_amd.models.controller import Controller
from aadg_amd.models.graphed import FusedControllerStep
from aadg_amd.scheduler import CONTROLLER_LR

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M = 6


def times(fn, n):
    for _ in range(5):
        fn()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in pairs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in pairs]) * 1e3


cfg = get_default_config()
c = Controller(cfg).cuda()
crit = search_loss(cfg)
opt = torch.optim.Adam(c.parameters(), lr=CONTROLLER_LR)
crit.register_optimizer(opt)
step = FusedControllerStep(c, crit, opt, M)
rw = torch.randn(M, device="cuda")
ts = times(lambda: step.sample(), repeats)
ent = step.sample()[4]
tu = times(lambda: step.update(rw, ent), repeats)
print("%-28s sample %6.1f us (p10 %6.1f)   ppo update %6.1f us (p10 %6.1f, p90 %6.1f)" % (
    os.path.basename(os.environ.get("AADG_LIB_PATH", "built")), np.median(ts), np.percentile(ts, 10), np.median(tu), np.percentile(tu, 10),
    np.percentile(tu, 90)))
