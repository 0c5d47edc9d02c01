"""One BatchNorm + ReLU layer (and one with a residual) forward + backward on layer-1-sized float32 tensors, a few times: the target of
    rocprofv3 --kernel-trace --stats -- python scripts/r6/bn_time.py      (per-kernel durations -> TB/s by hand: tensor = N C H W x 4 bytes)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib
torch.manual_seed(0)
for (N, C, H) in ((144, 256, 128), (144, 64, 128), (144, 512, 64), (144, 1024, 32)):
    x = torch.randn(N, C, H, H, device="cuda", requires_grad=True)
    r = torch.randn(N, C, H, H, device="cuda", requires_grad=True)
    w = torch.ones(C, device="cuda", requires_grad=True); b = torch.zeros(C, device="cuda", requires_grad=True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    g = torch.randn(N, C, H, H, device="cuda")
    print("tensor %d x %d x %d x %d = %.0f MB" % (N, C, H, H, x.numel() * 4 / 1e6))
    for it in range(3):
        y = _lib.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, _lib.ACT_RELU)
        y.backward(g)
        y = _lib.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, _lib.ACT_RELU, r)
        y.backward(g)
    torch.cuda.synchronize()
    del x, r, g, y
    torch.cuda.empty_cache()
