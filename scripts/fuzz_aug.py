"""Differential fuzz of the augmentation call against the CPU oracle (test infrastructure): random source sizes (square and not, down
to 64, up to 520 -- patches wider than 256 pixels, partial tiles), crop sizes, scale ranges that cover up- and down-scaling by up to 2,
1 - 4 op slots (chains of Sharpness stencils, statistics ops in late slots), both datasets.  Every output byte / float must be equal.
    python scripts/fuzz_aug.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aadg_amd import _lib
from oracle import oracle as O
from helpers import random_units, synth_pool

def run(cases, seed, verbose=True):
    rs = np.random.RandomState(seed)
    sizes = [64, 96, 128, 198, 200, 250, 256, 260, 320, 512, 520]
    bad = 0
    for c in range(cases):
        H, W = sizes[rs.randint(len(sizes))], sizes[rs.randint(len(sizes))]
        if rs.rand() < 0.5: W = H
        crop = [32, 64, 90, 100, 128, 130, 256, 260, 512][rs.randint(9)]
        crop = min(crop, 2 * min(H, W))
        L = [1, 2, 2, 2, 3, 4][rs.randint(6)]
        rng = [(1.0, 1.5), (0.5, 2.0), (0.5, 1.0), (0.7, 1.3)][rs.randint(4)]
        dataset = rs.randint(2)
        P, N = 3, 12 + rs.randint(20)
        imgs, msks = synth_pool(rs, P, H, W)
        units = random_units(rs, N, P, H, W, crop, rng, L=L)
        try:
            pool, mk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
            # both flows: statistics passes of the call itself / the per-pool statistics cache + late lists (the hot path's)
            ph = _lib.pool_histograms(pool) if c % 2 == 0 else None
            gi, gl = _lib.aug_u8_forward(pool, mk, units, crop, dataset, pool_hist=ph)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print("case %d H=%d W=%d crop=%d L=%d range=%s: call failed: %s" % (c, H, W, crop, L, rng, e)); bad += 1; continue
        wi, wl = O.aug_units(imgs, msks, units, crop, dataset)
        ok = np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gl.cpu().numpy(), wl)
        if not ok:
            d = np.argwhere(gi.cpu().numpy() != wi)
            print("case %d H=%d W=%d crop=%d L=%d range=%s dataset=%d: MISMATCH at %d pixels, first %s" % (c, H, W, crop, L, rng, dataset, len(d), d[:1]))
            bad += 1
    if verbose:
        print("fuzz: %d cases, %d bad" % (cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 20260929) else 0)
