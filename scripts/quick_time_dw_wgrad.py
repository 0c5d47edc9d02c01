"""Ad-hoc: depthwise 3x3 weight gradient on the decoder / ASPP shapes (bf16, NB images): register-window kernel vs the LDS one
(AADG_DW_WGRAD_LDS=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd import _lib
N = int(os.environ.get("NB", "144"))
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for C, S, d in ((304, 128, 1), (256, 128, 1), (2048, 32, 12)):
    x = torch.randn(N, C, S, S, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, C, S, S, device="cuda", dtype=torch.bfloat16)
    dw = torch.empty(C, 1, 3, 3, device="cuda")
    lib = _lib.load()
    ws = torch.empty(lib.aadg_dwconv3x3_workspace_bytes(C), dtype=torch.uint8, device="cuda")
    def run():
        rc = lib.aadg_dwconv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, C, S, S, d, 1, ws.data_ptr(), ws.numel(), _lib._stream())
        assert rc == 0
    t = bench(run)
    print("C=%d %dx%d d=%d: %.3f ms (%.2f TB/s)" % (C, S, S, d, t, 2.0 * x.numel() * 2 / t / 1e9))
