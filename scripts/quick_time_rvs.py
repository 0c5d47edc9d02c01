"""Ad-hoc timing of the augmentation path at an RVS-like setting (scale range [0.5, 2], K=1)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from aadg_amd import _lib
from helpers import random_units, synth_pool

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
crop = int(sys.argv[2]) if len(sys.argv) > 2 else H
rs = np.random.RandomState(1023)
P, N = 24, 144
imgs, msks = synth_pool(rs, P, H, H, vessel=True)
units = random_units(rs, N, P, H, H, crop, (0.5, 2.0))
cls, sm = _lib.launch_hints(units, H, H, crop)[:2]
n_f = int(((units["scaled_w"] >= H) & (units["scaled_h"] >= H)).sum())
d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
oi = torch.empty((N, 3, crop, crop), device="cuda"); ol = torch.empty((N, 1, crop, crop), device="cuda")
for _ in range(3): _lib.aug_u8_forward(d_img, d_msk, units, crop, 1, oi, ol)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): _lib.aug_u8_forward(d_img, d_msk, units, crop, 1, oi, ol)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
alg = N * (3 * H * H + H * H + 4 * crop * crop * 4)
print("rvs-like src %d crop %d N=%d (%d fused-eligible): %.3f ms/batch, algorithmic %.0f MB -> %.0f GB/s" % (H, crop, N, n_f, ms, alg / 1e6, alg / ms / 1e6))
