"""Ad-hoc: forward and input gradient of the two stride-2 3x3 convolutions of the backbone: own kernel vs the library (bf16, NB images)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib
N = int(os.environ.get("NB", "144"))
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for C, So in ((128, 64), (256, 32)):
    x = torch.randn(N, C, 2 * So, 2 * So, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, C, So, So, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(C, C, 3, 3, device="cuda", dtype=torch.bfloat16)
    a9t = w.permute(2, 3, 1, 0).reshape(9, C, C).contiguous()
    lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    ours = lambda: _lib.conv3x3s2_dgrad(a9t, dy)
    a9 = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
    lib_f = lambda: torch.ops.aten.convolution(x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1)
    ours_f = lambda: _lib.conv3x3s2_nchw(a9, x)
    t2, t3 = bench(lib_f), bench(ours_f)
    print("C=%d out %dx%d  forward: library %.3f ms | own %.3f ms (%.0f TFLOP/s)" % (C, So, So, t2, t3, 2.0 * N * So * So * C * C * 9 / t3 / 1e9))
    t0, t1 = bench(lib), bench(ours)
    print("C=%d out %dx%d  library %.3f ms | own %.3f ms (%.0f TFLOP/s)" % (C, So, So, t0, t1, 2.0 * N * So * So * C * C * 9 / t1 / 1e9))
