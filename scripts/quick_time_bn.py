"""Ad-hoc: fused BatchNorm(+ReLU) kernels vs torch (MIOpen) on the backbone's shapes, bf16, N=144 @512x512."""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib

N = int(os.environ.get("NB", "144"))
shapes = [(64, 256), (64, 128), (256, 128), (128, 64), (512, 64), (256, 32), (1024, 32), (512, 32), (2048, 32), (256, 128 * 0 + 128)]
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
if os.environ.get("ONLY"):
    shapes = [shapes[int(i)] for i in os.environ["ONLY"].split(",")]
tot = [0, 0]
for C, S in shapes:
    x = torch.randn(N, C, S, S, device="cuda").bfloat16().requires_grad_(True)
    w = torch.ones(C, device="cuda", requires_grad=True); b = torch.zeros(C, device="cuda", requires_grad=True)
    rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    g = torch.randn_like(x)
    def ours():
        y = _lib.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, 1)
        y.backward(g)
    def ref():
        y = F.relu(F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5))
        y.backward(g)
    def ours_f():
        with torch.no_grad(): _lib._BatchNormAct.apply(x, None, w, b, rm, rv, 0.1, 1e-5, 1, False)
    def ref_f():
        with torch.no_grad(): F.relu(F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5))
    r = torch.randn_like(x).requires_grad_(True)
    g2 = torch.randn_like(x)
    def ours_res():          # a residual block's last BN: + identity, ReLU, output consumed twice (two gradients summed in-kernel)
        ya, yb = _lib.batch_norm_act(x, w, b, rm, rv, True, 0.1, 1e-5, 1, r, handles=2)
        torch.autograd.backward([ya, yb], [g, g2])
    to, tr, tof, trf = bench(ours), bench(ref), bench(ours_f), bench(ref_f)
    tres = bench(ours_res)
    gb = x.numel() * 2 / 1e9
    print("C=%4d %3dx%-3d  %.2f GB | fwd+bwd ours %.2f ms (%.0f GB/s of 8 passes) torch %.2f ms | fwd ours %.2f ms (%.0f GB/s of 3 passes) torch %.2f ms"
          % (C, S, S, gb, to, gb * 8 / to * 1e3, tr, tof, gb * 3 / tof * 1e3, trf), flush=True)
    print("        residual + 2 handles fwd+bwd %.2f ms (%.0f GB/s of 12 passes)" % (tres, gb * 12 / tres * 1e3), flush=True)
    tot[0] += to; tot[1] += tr
print("sum ours %.1f ms  torch %.1f ms" % tuple(tot))
