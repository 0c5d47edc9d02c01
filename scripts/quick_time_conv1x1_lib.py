"""Ad-hoc: library (MIOpen -> hipBLASLt/Tensile) forward and input-gradient of the backbone's 1x1 convolutions, bf16, N = 144:
achieved bytes/s against the activation traffic (read input + write output)."""
import os, sys, time, torch
import torch.nn.functional as F
N = int(os.environ.get("NB", "144"))
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
cases = [(64, 64, 128, 1), (64, 256, 128, 4), (256, 64, 128, 2), (256, 128, 128, 1), (128, 512, 64, 4), (512, 128, 64, 3), (512, 256, 64, 1), (256, 1024, 32, 6),
         (1024, 256, 32, 5), (1024, 512, 32, 1), (512, 2048, 32, 3), (2048, 512, 32, 2), (1024, 2048, 32, 1), (2048, 256, 32, 4), (1280, 256, 32, 1), (256, 48, 128, 1),
         (304, 256, 128, 1), (256, 256, 128, 1)]
tot = [0.0, 0.0, 0.0]
for Ci, Co, S, cnt in cases:
    x = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, Co, S, S, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(Co, Ci, 1, 1, device="cuda", dtype=torch.bfloat16)
    fwd = lambda: F.conv2d(x, w)
    bwd = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    t0, t1 = bench(fwd), bench(bwd)
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    ideal = gb / 5.0          # ms at 5 TB/s
    tot[0] += cnt * t0; tot[1] += cnt * t1; tot[2] += cnt * ideal
    print("Ci=%4d Co=%4d %3dx%-3d x%d  fwd %.3f ms (%.0f GB/s)  dgrad %.3f ms (%.0f GB/s)  | %.3f ms at 5 TB/s" %
          (Ci, Co, S, S, cnt, t0, gb / t0 * 1e3, t1, gb / t1 * 1e3, ideal), flush=True)
print("weighted by layer count: fwd %.2f ms  dgrad %.2f ms  | at 5 TB/s %.2f ms each" % tuple(tot))
