"""Event times of the fused BCE / Dice pass at the headline shape [144, 2, 512, 512]: python scripts/r6/segloss_time.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib
_lib.load()
N, K, S, M = 144, 2, 512, 6
z = torch.randn(N, K, S, S, device="cuda"); y = (torch.rand(N, K, S, S, device="cuda") > 0.5).float()
junk = torch.empty(1 << 28, device="cuda")
ts = []
for i in range(30):
    junk.fill_(float(i))                      # evict the operands from the caches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.seg_bce_dice(z, y, M, want_grad=True); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
b = N * K * S * S * 12
print("k_seg_loss %.1f us median (min %.1f)  %.2f TB/s  frac %.3f" % (np.median(ts) * 1e3, min(ts) * 1e3, b / np.median(ts) / 1e9, b / np.median(ts) / 1e9 / 8))
g = torch.empty_like(z)
for name, fn in (("torch.add(z, y, out=g)  [2 reads + 1 write, same bytes]", lambda: torch.add(z, y, out=g)), ("g.copy_(z)  [1 read + 1 write]", lambda: g.copy_(z))):
    ts = []
    for i in range(20):
        junk.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    bb = N * K * S * S * (12 if "add" in name else 8)
    print("%s: %.1f us  %.2f TB/s  frac %.3f" % (name, np.median(ts) * 1e3, bb / np.median(ts) / 1e9, bb / np.median(ts) / 1e9 / 8))
