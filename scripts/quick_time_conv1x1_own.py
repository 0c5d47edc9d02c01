"""Ad-hoc: forward / input gradient of the backbone's 1x1 convolutions: matrix-core kernel with LDS transpose reads
(csrc/conv1x1_fwd.hip) vs the library (MIOpen -> hipBLASLt), bf16, N = 144."""
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
cases = [(64, 64, 128, 1), (64, 256, 128, 4), (256, 64, 128, 2), (256, 128, 128, 1), (128, 512, 64, 4), (512, 128, 64, 3), (512, 256, 64, 1), (256, 1024, 32, 6),
         (1024, 256, 32, 5), (1024, 512, 32, 1), (512, 2048, 32, 3), (2048, 512, 32, 2), (1024, 2048, 32, 1), (2048, 256, 32, 4), (1280, 256, 32, 1), (256, 48, 128, 1),
         (304, 256, 128, 1), (256, 256, 128, 1), (256, 512, 64, 1), (512, 1024, 32, 1)]
tot = [0.0] * 4
for Ci, Co, S, cnt in cases:
    x = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, Co, S, S, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05).to(torch.bfloat16)
    w2, wt = w.view(Co, Ci), w.view(Co, Ci).t().contiguous()
    fwd = lambda: F.conv2d(x, w)
    fwd_o = lambda: _lib.conv1x1_nchw(w2, x)
    bwd = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    bwd_o = lambda: _lib.conv1x1_nchw(wt, dy)
    a, b = fwd().float(), fwd_o().float()
    e1 = (a - b).abs().max().item() / a.abs().max().item()
    a, b = bwd().float(), bwd_o().float()
    e2 = (a - b).abs().max().item() / a.abs().max().item()
    t = [bench(fwd), bench(fwd_o), bench(bwd), bench(bwd_o)]
    for i in range(4): tot[i] += cnt * t[i]
    print("Ci=%4d Co=%4d %3dx%-3d x%d  fwd lib %.3f own %.3f | dgrad lib %.3f own %.3f | rel diff %.1e %.1e" % (Ci, Co, S, S, cnt, *t, e1, e2), flush=True)
print("weighted: fwd lib %.2f own %.2f | dgrad lib %.2f own %.2f" % tuple(tot))
