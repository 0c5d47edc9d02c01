"""Library route of the 1x1 convolutions that stay with the library at N = 144: aten.convolution (MIOpen -> hipBLASLt, what the model calls)
against torch.matmul of the same operands (W [Co, Ci] @ x [N, Ci, HW]), optionally under TunableOp (PYTORCH_TUNABLEOP_ENABLED=1)."""
import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import _lib
N = int(os.environ.get("NB", "144"))
def bench(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
# (Ci, Co, S, count per step) of the ResNet-50 / ASPP 1x1 convolutions
cases = [(64, 64, 128, 1), (64, 256, 128, 4), (256, 64, 128, 2), (256, 128, 128, 1), (128, 512, 64, 4), (512, 128, 64, 3), (512, 256, 64, 1), (256, 1024, 32, 6),
         (1024, 256, 32, 5), (1024, 512, 32, 1), (512, 2048, 32, 3), (2048, 512, 32, 2), (1024, 2048, 32, 1), (2048, 256, 32, 4), (1280, 256, 32, 1), (256, 48, 128, 1),
         (304, 256, 128, 1), (256, 256, 128, 1), (256, 512, 64, 1), (512, 1024, 32, 1)]
tot = [0.0] * 4
for Ci, Co, S, cnt in cases:
    x = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, Co, S, S, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05).to(torch.bfloat16)
    w2, wt = w.view(Co, Ci), w.view(Co, Ci).t().contiguous()
    line = "Ci=%4d Co=%4d %3dx%-3d x%d " % (Ci, Co, S, S, cnt)
    if not _lib._own_gemm_1x1(Co, Ci, S * S, N):
        a = bench(lambda: F.conv2d(x, w))
        b = bench(lambda: torch.matmul(w2, x.view(N, Ci, S * S)))
        tot[0] += cnt * a; tot[1] += cnt * b
        line += " fwd conv %.3f matmul %.3f" % (a, b)
    if not _lib._own_gemm_1x1(Ci, Co, S * S, N):
        a = bench(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0])
        b = bench(lambda: torch.matmul(wt, dy.view(N, Co, S * S)))
        tot[2] += cnt * a; tot[3] += cnt * b
        line += " | dgrad conv %.3f matmul %.3f" % (a, b)
    print(line, flush=True)
print("weighted per step: fwd conv %.2f matmul %.2f | dgrad conv %.2f matmul %.2f" % tuple(tot))
