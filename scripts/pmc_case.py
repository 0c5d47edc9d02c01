"""One ablation case of scripts/ablate_fused.py, repeated a few times: the target of a `rocprofv3 --pmc ...` run."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aadg_amd import _lib
from helpers import random_units, synth_pool

case = sys.argv[1] if len(sys.argv) > 1 else "identity"
H, P, N = 512, 24, 168
rs = np.random.RandomState(1023)
imgs, msks = synth_pool(rs, P, H, H)
d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
oi = torch.empty((N, 3, H, H), device="cuda"); ol = torch.empty((N, 2, H, H), device="cuda")
if case == "identity":
    u = random_units(np.random.RandomState(7), N, P, H, H, H, (1.0, 1.0), L=2, p_scale=0.0); u['n_ops'] = 0
elif case == "scaled":
    u = random_units(np.random.RandomState(7), N, P, H, H, H, (1.0, 1.5), L=2, p_scale=1.0); u['n_ops'] = 0
else:
    u = random_units(np.random.RandomState(7), N, P, H, H, H, (1.0, 1.5), L=2, p_scale=0.8)
for _ in range(4):
    _lib.aug_u8_forward(d_img, d_msk, u, H, 0, oi, ol)
torch.cuda.synchronize()
