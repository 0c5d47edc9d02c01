"""Quick timing of selected float ops at [144,3,512,512] (kernel experiments): python scripts/fop_quick.py op1 op2 ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from aadg_amd import _lib
B, size = 144, 512
x = torch.rand(B, 3, size, size, device="cuda")
perm = torch.randperm(B, device="cuda").to(torch.int32)
mags = dict(bench.FLOAT_OPS)
for name in sys.argv[1:]:
    m0 = mags[name]
    mag = None if m0 is None else torch.tensor([m0], device="cuda")
    kw = {"perm": perm} if name == "sample_pairing" else {}
    ts = bench._event_times(lambda: _lib.fop(name, x, mag, **kw), 10)
    nb = (36 if name in bench.STAT_FOPS else 24) * size * size * B
    print("%-18s %.3f ms (min %.3f)  %.0f GB/s  %.2f" % (name, np.median(ts), min(ts), nb / np.median(ts) / 1e6, nb / np.median(ts) / 1e6 / 8000))
