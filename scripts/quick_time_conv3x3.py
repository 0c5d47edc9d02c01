"""Ad-hoc: 3x3 convolution forward / input gradient: own kernel (csrc/conv3x3_fwd.hip) vs the library on the backbone's shapes."""
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
cases = [(64, 128, 1, 3), (128, 64, 1, 3), (256, 32, 1, 5), (512, 32, 2, 3)]       # (channels, size, dilation, count in the backbone)
tot = [0.0, 0.0, 0.0]
for C, S, d, cnt in cases:
    x = torch.randn(N, C, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, C, S, S, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(C, C, 3, 3, device="cuda", dtype=torch.bfloat16)
    a9 = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
    def lib_f():
        return torch.ops.aten.convolution(x, w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1)
    def lib_b():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1, [True, False, False])[0]
    def ours():
        return _lib.conv3x3_nchw(a9, x, d)
    skip = os.environ.get("SKIP_LIB")
    t0 = 0.0 if skip else bench(lib_f)
    t2 = 0.0 if skip else bench(lib_b)
    t1 = bench(ours)
    tot[0] += t0 * cnt; tot[1] += t1 * cnt; tot[2] += t2 * cnt
    fl = 2.0 * N * S * S * C * C * 9
    print("C=%4d %3dx%-3d d=%d  library fwd %.3f bwd %.3f ms | own %.3f ms (%.0f TFLOP/s)  x%d" % (C, S, S, d, t0, t2, t1, fl / t1 / 1e9, cnt), flush=True)
print("per step and direction: library fwd %.2f ms, own %.2f ms, library bwd %.2f ms" % tuple(tot))
