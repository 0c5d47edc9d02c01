"""Ad-hoc: depthwise 3x3 kernels vs torch on the DeepLabV3+ head shapes, bf16 autocast, N=144 @512x512."""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib
N = int(os.environ.get("NB", "144"))
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
for C, S, d in [(2048, 32, 12), (2048, 32, 24), (2048, 32, 36), (256, 32, 1), (304, 128, 1)]:
    x = torch.randn(N, C, S, S, device="cuda").bfloat16().requires_grad_(True)
    w = torch.randn(C, 1, 3, 3, device="cuda", requires_grad=True)
    g = torch.randn_like(x)
    def ours():
        x.grad = None; w.grad = None
        _lib.dwconv3x3(x, w, d).backward(g)
    def ref():
        x.grad = None; w.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            F.conv2d(x, w, None, 1, d, d, C).backward(g)
    def ours_f():
        with torch.no_grad(): _lib._DepthwiseConv3x3.apply(x, w, d)
    to, tr, tf = bench(ours), bench(ref), bench(ours_f)
    gb = x.numel() * 2 / 1e9
    print("C=%4d %3dx%-3d d=%2d %.2f GB | fwd+bwd ours %.2f ms (%.0f GB/s of 6 passes) torch %.2f ms | fwd ours %.3f ms (%.0f GB/s)" %
          (C, S, S, d, gb, to, gb * 6 / to * 1e3, tr, tf, gb * 2 / tf * 1e3), flush=True)
