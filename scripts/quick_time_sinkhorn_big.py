"""Ad-hoc: scaled synthetic Sinkhorn: 3 domain pairs of 4096-point clouds, E = 128."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib
n, E, D = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 128, 3
torch.manual_seed(0)
x = torch.nn.functional.leaky_relu(torch.randn(D * n, E, device="cuda") * 0.5 + torch.randn(D, 1, E, device="cuda").repeat(1, n, 1).view(-1, E), 0.2)
rows = torch.arange(D * n, dtype=torch.int32, device="cuda")
off = torch.arange(0, (D + 1) * n, n, dtype=torch.int32, device="cuda")
pxy = torch.tensor([0, 1, 0, 2, 1, 2], dtype=torch.int32, device="cuda")
for _ in range(2): out = _lib.sinkhorn_divergence(x, rows, off, pxy, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): out = _lib.sinkhorn_divergence(x, rows, off, pxy, n)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("3 problems of %dx%d points, E=%d: %.3f ms per call; S = %s" % (n, n, E, ms, out.cpu().numpy()))
print("cost build: 3 x 4 GEMMs = %.1f GFLOP; one sweep reads %.0f MB" % (3 * 4 * 2 * n * n * E / 1e9, 3 * 4 * n * n * 4 / 1e6))
