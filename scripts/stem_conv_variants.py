"""Ad-hoc: the 7x7/2 stem convolution (3 -> 64) fwd + weight-gradient with the input padded to 3 / 4 / 8 channels."""
import os, sys, time, torch
import torch.nn.functional as F
N = int(os.environ.get("NB", "144"))
for cin in (3, 4, 8):
    x = torch.randn(N, cin, 512, 512, device="cuda")
    w = torch.randn(64, cin, 7, 7, device="cuda", requires_grad=True)
    def it():
        w.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = F.conv2d(x, w, None, 2, 3)
        y.backward(torch.ones_like(y))
    t0 = time.time(); it(); torch.cuda.synchronize(); t1 = time.time()
    for _ in range(3): it()
    torch.cuda.synchronize(); t2 = time.time()
    print("cin=%d first %.1fs iter %.2f ms" % (cin, t1 - t0, (t2 - t1) / 3 * 1e3), flush=True)
