"""Does a BatchNorm layer run faster when its channels are processed in chunks whose tensors fit the Infinity Cache (statistics pass, then
the elementwise pass re-reading the chunk from the cache)?  Emulation with the existing entry points: one [N, C, H, W] layer against C / CC
separate contiguous [N, CC, H, W] layers run one after the other (forward with fused residual + ReLU, and backward)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import _lib

def run(N, C, H, CC, reps=5):
    parts = C // CC
    xs = [torch.randn(N, CC, H, H, device="cuda").bfloat16().requires_grad_(True) for _ in range(parts)]
    rs = [torch.randn(N, CC, H, H, device="cuda").bfloat16().requires_grad_(True) for _ in range(parts)]
    gs = [torch.randn(N, CC, H, H, device="cuda").bfloat16() for _ in range(parts)]
    w = [torch.ones(CC, device="cuda").requires_grad_(True) for _ in range(parts)]
    b = [torch.zeros(CC, device="cuda").requires_grad_(True) for _ in range(parts)]
    rm = [torch.zeros(CC, device="cuda") for _ in range(parts)]
    rv = [torch.ones(CC, device="cuda") for _ in range(parts)]
    def fwd():
        return [_lib.batch_norm_act(xs[i], w[i], b[i], rm[i], rv[i], True, 0.1, 1e-5, 1, rs[i]) for i in range(parts)]
    def both():
        ys = fwd()
        torch.cuda.synchronize()
        t = time.time()
        for i in range(parts):
            ys[i].backward(gs[i])
        torch.cuda.synchronize()
        return time.time() - t
    for _ in range(2): fwd()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): fwd()
    torch.cuda.synchronize()
    tf = (time.time() - t) / reps * 1e3
    both()
    tb = sum(both() for _ in range(reps)) / reps * 1e3
    gb = N * C * H * H * 2 / 1e9
    print("N=%d C=%d %dx%d (%.2f GB per tensor) in chunks of %3d channels (%2d launches pairs): fwd %.3f ms, bwd %.3f ms" % (N, C, H, H, gb, CC, parts, tf, tb), flush=True)

for (C, H) in ((256, 128), (512, 64), (2048, 32)):
    for CC in (C, C // 4, C // 8, C // 16, C // 32):
        run(144, C, H, CC)
