"""Ad-hoc: 1x1 conv weight gradient: MFMA kernel vs MIOpen on the backbone's shapes (bf16, N = 144)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib
N = int(os.environ.get("NB", "144"))
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
cases = [(64, 64, 128), (64, 256, 128), (256, 64, 128), (256, 128, 128), (128, 512, 64), (512, 128, 64), (512, 256, 64), (256, 1024, 32), (1024, 256, 32),
         (1024, 512, 32), (512, 2048, 32), (2048, 512, 32), (2048, 256, 32), (1280, 256, 32), (256, 48, 128), (304, 256, 128)]
tot = [0.0, 0.0]
for Ci, Co, S in cases:
    x = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, Co, S, S, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(Co, Ci, 1, 1, device="cuda", dtype=torch.bfloat16)
    def miopen():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    def ours():
        return _lib.conv1x1_wgrad(dy, x)
    if os.environ.get("SKIP_LIB"):
        err, t0, t1 = 0.0, 0.0, bench(ours, 20)
    else:
        a = miopen().float().view(Co, Ci); b = ours()
        err = (a - b).abs().max().item() / max(1e-6, a.abs().max().item())
        t0, t1 = bench(miopen), bench(ours)
    tot[0] += t0; tot[1] += t1
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    print("Ci=%4d Co=%4d %3dx%-3d miopen %.3f ms | mfma %.3f ms (%.0f GB/s, %.0f TFLOP/s) | rel diff %.1e" %
          (Ci, Co, S, S, t0, t1, gb / t1 * 1e3, 2.0 * N * S * S * Ci * Co / t1 / 1e9, err), flush=True)
print("sum miopen %.2f  mfma %.2f" % tuple(tot))
