"""Where does k_fused's time go?  Times the final kernel of aadg_aug_u8_forward_ex (HIP events around it) at 512x512, 168 units,
for controlled op mixes / scale classes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aadg_amd import _lib
from helpers import random_units, synth_pool

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rs = np.random.RandomState(1023)
P, N = 24, 168
imgs, msks = synth_pool(rs, P, H, H)
d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
oi = torch.empty((N, 3, H, H), device="cuda"); ol = torch.empty((N, 2, H, H), device="cuda")


def units_for(ops, scale, p_scale=1.0):
    u = random_units(np.random.RandomState(7), N, P, H, H, H, scale, L=2, p_scale=p_scale)
    if ops is not None:
        u['n_ops'] = len(ops)
        for k, (op, ia, fa) in enumerate(ops):
            u['op'][:, k] = op; u['iarg'][:, k] = ia; u['farg'][:, k] = fa
            u['rect'][:, k] = (0, 0, -1, -1)
    return u


def time_units(u, reps=10):
    pe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    pe[0].record(); pe[1].record(); torch.cuda.synchronize()
    ts = []
    for i in range(reps + 2):
        _lib.PROFILE_EVENTS = pe
        _lib.aug_u8_forward(d_img, d_msk, u, H, 0, oi, ol)
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(pe[0].elapsed_time(pe[1]))
    _lib.PROFILE_EVENTS = None
    return float(np.median(ts)) * 1e3


cases = [
    ("no ops, identity scale", units_for([], (1.0, 1.0), 0.0)),
    ("no ops, scale 1..1.5", units_for([], (1.0, 1.5))),
    ("no ops, scale 1.5", units_for([], (1.5, 1.5))),
    ("2 x Invert, scaled", units_for([(1, 0, 0), (1, 0, 0)], (1.0, 1.5))),
    ("2 x Brightness (LUT), scaled", units_for([(7, 0, 1.3), (7, 0, 0.7)], (1.0, 1.5))),
    ("2 x Color, scaled", units_for([(6, 0, 1.3), (6, 0, 0.7)], (1.0, 1.5))),
    ("1 x Sharpness, scaled", units_for([(8, 0, 1.3)], (1.0, 1.5))),
    ("2 x Sharpness, scaled", units_for([(8, 0, 1.3), (8, 0, 0.7)], (1.0, 1.5))),
    ("random mix (bench-like, p_scale .8)", units_for(None, (1.0, 1.5), 0.8)),
]
byts = N * (4 * H * H + 5 * H * H * 4)
for name, u in cases:
    us = time_units(u)
    print("%-40s %7.1f us   %5.2f TB/s" % (name, us, byts / us / 1e6))
