"""Times the fused controller sample / PPO update (HIP events around the library calls).  AADG_CTRL_GENERIC=1 selects the
run-time-width kernels (weights re-read per step) for an A/B on the same box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import Cfg
from aadg_amd import _lib as hip, losses
from aadg_amd.models.controller import Controller
from aadg_amd.models.graphed import make_controller_step

cfg = Cfg()
cfg.CONTROLLER.T, cfg.CONTROLLER.C, cfg.CONTROLLER.PENALTY, cfg.CONTROLLER.LOSS = 2, 2.5, 1e-5, "ppo"
cfg.CONTROLLER.L, cfg.CONTROLLER.NUM_MAGS, cfg.CONTROLLER.EXCLUDE_OPS_NUM = 2, 10, 0
M = 6
torch.manual_seed(0)
c = Controller(cfg).cuda()
opt = torch.optim.Adam(c.parameters(), lr=0.00035)
f = make_controller_step(c, losses.search_loss(cfg), opt, M)
reward = torch.randn(M, device="cuda")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


state = {}


def sample():
    state["s"] = f.sample()


def update():
    f.update(reward, state["s"][4])


sample()
print("sample  %.1f us" % timed(sample))
print("update  %.1f us (5 epochs)" % timed(update))
