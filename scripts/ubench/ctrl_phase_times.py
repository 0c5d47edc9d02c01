"""Phase timestamps of k_ctrl_rollout (instrumented copy of controller.hip: a wall_clock64 stamp after every barrier of
workgroup 0).  Run with AADG_LIB_PATH=scripts/ubench/libaadg_timed.so."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from helpers import Cfg
from aadg_amd import _lib as hip, losses
from aadg_amd.models.controller import Controller
from aadg_amd.models.graphed import make_controller_step

cfg = Cfg()
cfg.CONTROLLER.T, cfg.CONTROLLER.C, cfg.CONTROLLER.PENALTY, cfg.CONTROLLER.LOSS = 2, 2.5, 1e-5, "ppo"
cfg.CONTROLLER.L, cfg.CONTROLLER.NUM_MAGS, cfg.CONTROLLER.EXCLUDE_OPS_NUM = 2, 10, 0
M = 6
c = Controller(cfg).cuda()
f = make_controller_step(c, losses.search_loss(cfg), torch.optim.Adam(c.parameters(), lr=0.00035), M)
lib = hip.load()
buf = (ctypes.c_ulonglong * 1000)()


def dump(tag):
    n = lib.aadg_debug_ctrl_times(buf, 500)
    t0 = buf[0]
    print(tag, "stamps", n)
    prev = t0
    for i in range(n):
        t, line = buf[2 * i], buf[2 * i + 1]
        print("  stamp %3d  +%6.2f us  (total %6.2f)" % (line, (t - prev) / 100.0, (t - t0) / 100.0))
        prev = t


for _ in range(3):
    s = f.sample()
torch.cuda.synchronize()
dump("sample")
f.update(torch.randn(M, device="cuda"), s[4])
torch.cuda.synchronize()
dump("update (last epoch)")
