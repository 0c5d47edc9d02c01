"""Ad-hoc: weight gradient of the backbone's 1x1 convolutions: MIOpen (NHWC igemm + transposes) vs bmm + sum."""
import os, sys, time, torch
N = int(os.environ.get("NB", "144"))
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
cases = [(64, 64, 128), (64, 256, 128), (256, 64, 128), (256, 128, 128), (128, 512, 64), (512, 128, 64), (512, 256, 64), (256, 1024, 32), (1024, 256, 32),
         (1024, 512, 32), (512, 2048, 32), (2048, 512, 32), (2048, 256, 32), (1280, 256, 32), (256, 48, 128)]
tot = [0.0, 0.0, 0.0]
for Ci, Co, S in cases:
    x = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, Co, S, S, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(Co, Ci, 1, 1, device="cuda", dtype=torch.bfloat16)
    def miopen():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    def bmm():
        return torch.bmm(dy.view(N, Co, S * S), x.view(N, Ci, S * S).transpose(1, 2)).sum(0, dtype=torch.float32)
    def mm_cat():   # K = HW per image, accumulate with baddbmm over chunks of images
        return torch.einsum("nok,nck->oc", dy.view(N, Co, S * S), x.view(N, Ci, S * S))
    a = miopen().float().view(Co, Ci); b = bmm()
    err = (a - b).abs().max().item() / max(1e-6, a.abs().max().item())
    t0, t1, t2 = bench(miopen), bench(bmm), bench(mm_cat)
    tot[0] += t0; tot[1] += t1; tot[2] += t2
    print("Ci=%4d Co=%4d %3dx%-3d miopen %.3f ms | bmm+sum %.3f ms | einsum %.3f ms | rel diff %.1e" % (Ci, Co, S, S, t0, t1, t2, err), flush=True)
print("sum miopen %.2f  bmm %.2f  einsum %.2f" % tuple(tot))
