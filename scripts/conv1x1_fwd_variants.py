"""Ad-hoc: forward / input-gradient of the 1x1 convolutions as (a) the library convolution, (b) torch.matmul with a broadcast
weight over NCHW images (hipBLASLt strided-batched GEMM chosen by its own heuristics)."""
import os, sys, time, torch
import torch.nn.functional as F
N = int(os.environ.get("NB", "144"))
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
cases = [(64, 256, 128), (256, 64, 128), (128, 512, 64), (512, 128, 64), (256, 1024, 32), (1024, 256, 32), (512, 2048, 32), (2048, 512, 32), (304, 256, 128)]
for Ci, Co, S in cases:
    x = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, Co, S, S, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(Co, Ci, 1, 1, device="cuda", dtype=torch.bfloat16)
    w2 = w.view(Co, Ci)
    wt = w2.t().contiguous()
    fwd = lambda: F.conv2d(x, w)
    fwd_mm = lambda: torch.matmul(w2, x.view(N, Ci, S * S))
    bwd = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    bwd_mm = lambda: torch.matmul(wt, dy.view(N, Co, S * S))
    try:
        a, b = fwd(), fwd_mm().view(N, Co, S, S)
        err = (a.float() - b.float()).abs().max().item() / a.float().abs().max().item()
        t = [bench(fwd), bench(fwd_mm), bench(bwd), bench(bwd_mm)]
        print("Ci=%4d Co=%4d %3dx%-3d  fwd conv %.3f matmul %.3f | dgrad conv %.3f matmul %.3f | rel diff %.1e" % (Ci, Co, S, S, *t, err), flush=True)
    except Exception as e:
        print("Ci=%4d Co=%4d failed: %s" % (Ci, Co, str(e)[:100]), flush=True)
