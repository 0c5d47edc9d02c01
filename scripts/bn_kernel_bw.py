"""Ad-hoc: runs the BatchNorm kernels on one shape (C, S from argv) so that a rocprofv3 --kernel-trace --stats run gives their
durations; prints the tensor size for the bandwidth arithmetic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib
N = int(os.environ.get("NB", "144"))
C, S = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(N, C, S, S, device="cuda").bfloat16().requires_grad_(True)
w = torch.ones(C, device="cuda", requires_grad=True); b = torch.zeros(C, device="cuda", requires_grad=True)
rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
g = torch.randn_like(x); g2 = torch.randn_like(x)
r = torch.randn_like(x).requires_grad_(True)
for _ in range(6):
    y = _lib.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, 1)
    y.backward(g)
for _ in range(6):
    ya, yb = _lib.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, 1, r, handles=2)
    torch.autograd.backward([ya, yb], [g, g2])
torch.cuda.synchronize()
print("tensor %.3f GB" % (x.numel() * 2 / 1e9))
