"""Ad-hoc: ResNet-50 3x3 convolutions fwd + bwd, contiguous (NCHW) vs channels_last tensors, bf16, N = 144."""
import os, sys, time, torch
import torch.nn.functional as F
N = int(os.environ.get("NB", "144"))
cases = [(64, 128, 1, 1), (128, 64, 1, 1), (256, 32, 1, 1), (512, 32, 1, 2), (128, 128, 2, 1)]
for cl in (0, 1):
    for C, S, stride, dil in cases:
        x = torch.randn(N, C, S, S, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(C, C, 3, 3, device="cuda", dtype=torch.bfloat16) * 0.05
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            w = w.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True); w.requires_grad_(True)
        def it():
            x.grad = None; w.grad = None
            y = F.conv2d(x, w, None, stride, dil, dil)
            y.backward(torch.ones_like(y))
            return y
        t0 = time.time(); y = it(); torch.cuda.synchronize(); t1 = time.time()
        for _ in range(3): it()
        torch.cuda.synchronize(); t2 = time.time()
        print("cl=%d C=%d %dx%d s=%d d=%d first %.1fs iter %.2f ms  y_cl=%s" % (cl, C, S, S, stride, dil, t1 - t0, (t2 - t1) / 3 * 1e3,
              y.is_contiguous(memory_format=torch.channels_last)), flush=True)
