"""Correctness of the big-tile f32x3 1x1 kernel (plain / statistics epilogue / BatchNorm on load) against float64, and its time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib
torch.manual_seed(0)
for (N, Co, Ci, H) in ((3, 256, 256, 16), (2, 512, 512, 32), (5, 1024, 256, 32)):
    x = torch.randn(N, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
    a = _lib.split_weight(w)
    ref = torch.einsum("mk,nkp->nmp", w.double(), x.double().flatten(2)).view(N, Co, H, H)
    y = _lib.conv1x1_nchw_x3(a, x)
    e0 = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    sums = torch.empty(2 * Co + 1, dtype=torch.float64, device="cuda")
    y2 = _lib.conv1x1_nchw_x3(a, x, sums)
    s_ref = torch.cat([torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))], 1).flatten(), torch.tensor([N * H * H], dtype=torch.float64, device="cuda")])
    e1 = ((sums - s_ref).abs().max() / s_ref.abs().max()).item()
    sc, sh = torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda") * 0.3
    ref3 = torch.einsum("mk,nkp->nmp", w.double(), torch.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).flatten(2)).view(N, Co, H, H)
    if Ci <= 512:
        y3 = _lib.conv1x1_nchw_x3(a, x, None, (sc, sh))
        e2 = ((y3.double() - ref3).abs().max() / ref3.abs().max()).item()
    else:
        e2 = float("nan")
    print("N%d %d->%d @%d: plain %.1e (equal with stats: %s)  stats %.1e  pre %.1e" % (N, Ci, Co, H, e0, torch.equal(y, y2), e1, e2), flush=True)
    assert e0 < 2e-5 and e1 < 1e-5 and not (e2 > 2e-5)
print("ok")
