"""Per-layer table of the 1x1 convolutions of the headline step (ResNet-50 / ASPP / decoder at N = 144, bfloat16): the route the model takes
(own kernel or library GEMM) for forward / input gradient, the own weight-gradient kernel, each against the layer's traffic floor
((Ci + Co) HW N 2 bytes at 5.5 TB/s) and its matrix-core floor (2 N HW Ci Co at 1.25 PFLOP/s).  python scripts/ab/conv1x1_layers.py"""
import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import _lib
N = int(os.environ.get("NB", "144"))
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = [(64, 64, 128, 1), (64, 256, 128, 4), (256, 64, 128, 2), (256, 128, 128, 1), (128, 512, 64, 4), (512, 128, 64, 3), (512, 256, 64, 1), (256, 1024, 32, 6),
         (1024, 256, 32, 5), (1024, 512, 32, 1), (512, 2048, 32, 3), (2048, 512, 32, 2), (1024, 2048, 32, 1), (2048, 256, 32, 4), (1280, 256, 32, 1), (256, 48, 128, 1),
         (304, 256, 128, 1), (256, 256, 128, 1), (256, 512, 64, 1), (512, 1024, 32, 1)]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "floor": 0.0}
print("%-28s %8s | %-18s %-18s %-12s" % ("layer", "floor us", "fwd us (route, x)", "dgrad us (route, x)", "wgrad us (x)"))
for Ci, Co, S, cnt in cases:
    x = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, Co, S, S, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05).to(torch.bfloat16)
    w2, wt = w.view(Co, Ci).contiguous(), w.view(Co, Ci).t().contiguous()
    HW = S * S
    floor = max((Ci + Co) * HW * N * 2 / 5.5e12, 2.0 * N * HW * Ci * Co / 1.25e15) * 1e6
    own_f, own_d = _lib._own_gemm_1x1(Co, Ci, HW, N), _lib._own_gemm_1x1(Ci, Co, HW, N)
    f = bench((lambda: _lib.conv1x1_nchw(w2, x)) if own_f else (lambda: F.conv2d(x, w)))
    d = bench((lambda: _lib.conv1x1_nchw(wt, dy)) if own_d else
              (lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]))
    g = bench(lambda: _lib.conv1x1_wgrad(dy, x))
    tot["fwd"] += cnt * f; tot["dgrad"] += cnt * d; tot["wgrad"] += cnt * g; tot["floor"] += cnt * floor
    print("Ci=%4d Co=%4d %3dx%-3d x%d %8.0f | %7.0f (%s %.2f)   %7.0f (%s %.2f)   %7.0f (%.2f)" % (
        Ci, Co, S, S, cnt, floor, f, "own" if own_f else "lib", f / floor, d, "own" if own_d else "lib", d / floor, g, g / floor), flush=True)
print("per step, ms: floor %.2f each | fwd %.2f dgrad %.2f wgrad %.2f" % (tot["floor"] / 1e3, tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3))
