"""Ad-hoc: stem convolution forward, MFMA kernel vs the library (bf16, N = 144 x 3 x 512 x 512)."""
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
x = torch.randn(N, 3, 512, 512, device="cuda").to(torch.bfloat16)
w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
wq = w.to(torch.bfloat16)
a = _lib.stem_conv7x7(x, w); b = F.conv2d(x, wq, stride=2, padding=3)
print("max diff %.3e of %.3e" % ((a.float() - b.float()).abs().max().item(), b.float().abs().max().item()))
t0 = bench(lambda: F.conv2d(x, wq, stride=2, padding=3)); t1 = bench(lambda: _lib.stem_conv7x7(x, w))
gb = (x.numel() + a.numel()) * 2 / 1e9
print("library %.3f ms | mfma %.3f ms (%.0f GB/s, %.0f TFLOP/s)" % (t0, t1, gb / t1 * 1e3, 2.0 * a.numel() * 147 / t1 / 1e9))
dy = torch.randn_like(a)
ref = lambda: torch.ops.aten.convolution_backward(dy, x, wq, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
def ours():
    dw = torch.empty_like(w)
    _lib._check(_lib.load().aadg_stem_conv7x7_wgrad_bf16(x.data_ptr(), 1, dy.data_ptr(), dw.data_ptr(), N, 512, 512, _lib._stream()), "wgrad")
    return dw
ga, gb_ = ours(), ref().float()
print("wgrad max diff %.3e of %.3e" % ((ga - gb_).abs().max().item(), gb_.abs().max().item()))
print("wgrad library %.3f ms | mfma %.3f ms" % (bench(ref), bench(ours)))
