"""Standalone times of the float32 depthwise 3x3 passes at the headline step's shapes: python scripts/r6/dw_time.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib
_lib.load()
junk = torch.empty(1 << 28, device="cuda")
def timed(fn, n=12):
    ts = []
    for i in range(n):
        junk.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for (N, C, H, W, d) in ((144, 304, 128, 128, 1), (144, 256, 128, 128, 1), (144, 2048, 32, 32, 12), (144, 2048, 32, 32, 24)):
    x = torch.randn(N, C, H, W, device="cuda", requires_grad=True)
    w = torch.randn(C, 1, 3, 3, device="cuda", requires_grad=True)
    y = _lib.dwconv3x3(x, w, d)
    g = torch.randn_like(y)
    nb = x.numel() * 4
    t_f = timed(lambda: _lib.dwconv3x3(x.detach(), w.detach(), d))
    lib = _lib.load()
    dx = torch.empty_like(x)
    t_d = timed(lambda: lib.aadg_dwconv3x3(g.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, W, d, 1, 0, _lib._stream()))
    dw = torch.empty_like(w)
    ws = torch.empty(lib.aadg_dwconv3x3_workspace_bytes(C), dtype=torch.uint8, device="cuda")
    t_w = timed(lambda: lib.aadg_dwconv3x3_wgrad(x.data_ptr(), g.data_ptr(), dw.data_ptr(), N, C, H, W, d, 0, ws.data_ptr(), ws.numel(), _lib._stream()))
    print("[%d,%d,%d,%d] d=%d  %.2f GB/tensor | fwd %.3f ms %.2f TB/s | dgrad %.3f ms %.2f TB/s | wgrad %.3f ms %.2f TB/s"
          % (N, C, H, W, d, nb / 1e9, t_f, 2 * nb / t_f / 1e9, t_d, 2 * nb / t_d / 1e9, t_w, 2 * nb / t_w / 1e9), flush=True)
    del x, y, g, dx
