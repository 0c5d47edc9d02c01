"""Ad-hoc: the SegFormer head's resize + add (csrc/upsample_sum.hip) on the shape of one of 8 ranks: [48 x 768, 128, 128] bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib
N, C, H = 48, 768, 128
full = torch.randn(N, C, H, H, device="cuda", dtype=torch.bfloat16, requires_grad=True)
lows = [torch.randn(N, C, H // f, H // f, device="cuda", dtype=torch.bfloat16, requires_grad=True) for f in (2, 4, 8)]
g = torch.randn_like(full)
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = [None]
def fwd():
    out[0] = _lib.upsample_sum(full, lows)
def bwd():
    out[0].backward(g, retain_graph=True)
fwd()
gb = full.numel() * 2 / 1e9
print("forward  %.3f ms (%.2f GB in + out)" % (bench(fwd), 2 * gb))
print("backward %.3f ms (%.2f GB read)" % (bench(bwd), gb))
