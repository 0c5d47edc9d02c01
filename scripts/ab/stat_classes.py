"""The statistics pass (k_hist_fused) by late-unit class: homogeneous batches of N_LATE late units (+ none else), each class launched
R times in a row; run under `rocprofv3 --kernel-trace` and read the per-launch durations back with scripts/ab/stat_classes_read.py.
Classes: (op0, op1) -- op1 in {Contrast 5, AutoContrast 0, Equalize 2}, op0 in {Brightness 7 (byte map), Color 6, Cutout 9, Sharpness 8}."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aadg_amd import _lib
from aadg_amd.data.basic import cutout_rect
from helpers import random_units, synth_pool

R = 5
order = []
for H, N in ((512, 27), (1024, 27)):
    rs = np.random.RandomState(7)
    P = 24
    imgs, msks = synth_pool(rs, P, H, H, vessel=True)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    ph = _lib.pool_histograms(d_img)
    oi = torch.empty((N, 3, H, H), device="cuda"); ol = torch.empty((N, 1, H, H), device="cuda")
    for op0, n0 in ((7, "brightness"), (6, "color"), (9, "cutout"), (8, "sharpness")):
        for op1, n1 in ((5, "contrast"), (0, "autocontrast"), (2, "equalize")):
            if op0 == 7 and op1 != 5:
                continue                                   # pushed forward: no pixel pass
            units = random_units(rs, N, P, H, H, H, (1.0, 1.5), L=0, p_scale=1.0)
            for u in units:
                u['n_ops'] = 2
                u['op'][0] = op0; u['farg'][0] = np.float32(1.3)
                if op0 == 9:
                    u['rect'][0] = cutout_rect(H, H, 0.15 * H, rs.uniform(H), rs.uniform(H))
                u['op'][1] = op1; u['farg'][1] = np.float32(1.4)
            for _ in range(R):
                _lib.aug_u8_forward(d_img, d_msk, units, H, 1, oi, ol, pool_hist=ph)
            torch.cuda.synchronize()
            order.append("%d %s->%s" % (H, n0, n1))
json.dump({"R": R, "order": order}, open(os.environ.get("STAT_CLASSES_OUT", "/tmp/stat_classes.json"), "w"))
