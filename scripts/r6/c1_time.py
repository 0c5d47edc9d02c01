"""The f32x3 1x1 kernel (forward / input gradient) on the backbone's compute-bound shapes, 144 images: python scripts/r6/c1_time.py
(A/B against a variant library: AADG_LIB_PATH=exp_libs/<x>.so PYTHONPATH=scripts/ab/hook python ...)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib
N = 144
torch.manual_seed(0)
for (Co, Ci, H) in ((2048, 512, 32), (512, 2048, 32), (1024, 256, 32), (256, 1024, 32), (256, 2048, 32), (2048, 1024, 32), (512, 128, 64), (128, 512, 64), (256, 64, 128)):
    x = torch.randn(N, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
    a = _lib.split_weight(w)
    f = lambda: _lib.conv1x1_nchw_x3(a, x)
    y = f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for p, q in ev:
        p.record(); f(); q.record()
    torch.cuda.synchronize()
    ref = torch.einsum("mk,nkp->nmp", w.double(), x[:2].double().flatten(2))
    err = (y[:2].double().flatten(2) - ref).abs().max().item()
    print("%4d->%4d @%3d: %.3f ms  (err vs float64 %.1e)" % (Ci, Co, H, sorted(p.elapsed_time(q) for p, q in ev)[3], err))
