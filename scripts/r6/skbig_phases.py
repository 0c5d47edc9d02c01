"""Cost build / sweeps of the large-cloud Sinkhorn apart (3 x 4096^2 points): python scripts/r6/skbig_phases.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib
n, E, D = 4096, 128, 3
torch.manual_seed(0)
x = torch.nn.functional.leaky_relu(torch.randn(D * n, E, device="cuda") * 0.5 + torch.randn(D, 1, E, device="cuda").repeat(1, n, 1).view(-1, E), 0.2)
rows = torch.arange(D * n, dtype=torch.int32, device="cuda")
off = torch.arange(0, (D + 1) * n, n, dtype=torch.int32, device="cuda")
pxy = torch.tensor([0, 1, 0, 2, 1, 2], dtype=torch.int32, device="cuda")
def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
out = _lib.sinkhorn_divergence(x, rows, off, pxy, n)
print("S =", out.cpu().numpy(), " whole %.3f ms  cost build %.3f ms  sweeps %.3f ms" % (
    timed(lambda: _lib.sinkhorn_divergence(x, rows, off, pxy, n)), timed(lambda: _lib.sinkhorn_divergence_phases(x, rows, off, pxy, n, 1)),
    timed(lambda: _lib.sinkhorn_divergence_phases(x, rows, off, pxy, n, 2))))
