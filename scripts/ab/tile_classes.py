"""Tile kernels by unit class: homogeneous batches (up / down scaling, without ops, with one Sharpness, with one LUT op) at 512 and 1024,
whole augmentation call timed with events; GB/s over the algorithmic bytes (source + mask once, 4 float planes out: vessel, K = 1)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aadg_amd import _lib
from helpers import random_units, synth_pool

def run(H, N, scale, ops):
    rs = np.random.RandomState(7)
    P = 24
    imgs, msks = synth_pool(rs, P, H, H, vessel=True)
    units = random_units(rs, N, P, H, H, H, scale, L=0, p_scale=1.0)
    for u in units:
        u['n_ops'] = len(ops)
        for k, (op, f) in enumerate(ops):
            u['op'][k] = op; u['farg'][k] = np.float32(f)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    oi = torch.empty((N, 3, H, H), device="cuda"); ol = torch.empty((N, 1, H, H), device="cuda")
    for _ in range(3): _lib.aug_u8_forward(d_img, d_msk, units, H, 1, oi, ol)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.aug_u8_forward(d_img, d_msk, units, H, 1, oi, ol); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    alg = N * (4 * H * H + 16 * H * H)
    return ms, alg / ms / 1e6

for H, N in ((512, 960), (1024, 320)):
    for name, scale in (("up", (1.0, 2.0)), ("down", (0.5, 0.99))):
        for oname, ops in (("no ops", []), ("1 sharpness", [(8, 1.5)]), ("2 sharpness", [(8, 1.5), (8, 0.4)]), ("color", [(6, 1.3)]), ("brightness", [(7, 1.3)])):
            ms, gbs = run(H, N, scale, ops)
            print("%4d %-5s %-12s %7.3f ms  %6.0f GB/s  %.3f of 8 TB/s" % (H, name, oname, ms, gbs, gbs / 8000), flush=True)
