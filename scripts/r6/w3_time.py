"""The dilation-2 weight gradient of the 3x3 convolution (k_wgrad3x3<W, 2>): time and result checksum on the layer-4 shape and a 64-pixel one.
    python scripts/r6/w3_time.py            (A/B against a variant library: AADG_LIB_PATH=exp_libs/<x>.so PYTHONPATH=scripts/ab/hook python ...)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib
torch.manual_seed(0)
for (Co, Ci, H, d, N, on_load) in ((512, 512, 32, 2, 144, False), (256, 256, 64, 2, 36, False), (512, 512, 32, 1, 144, False),
                                   (256, 256, 32, 1, 144, False), (256, 256, 32, 1, 144, True), (512, 512, 32, 2, 144, True), (128, 128, 64, 1, 144, True)):
    x = torch.randn(N, Ci, H, H, device="cuda")
    dy = torch.randn(N, Co, H, H, device="cuda")
    pre = (torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda") * 0.1) if on_load else None     # the layer's BatchNorm + ReLU on operand load
    f = lambda: _lib.conv3x3_wgrad_x3(dy, x, d, pre=pre)
    out = f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
    for p, q in ev:
        p.record(); f(); q.record()
    torch.cuda.synchronize()
    xin = x[:4].double() if pre is None else torch.relu(x[:4].double() * pre[0].double()[None, :, None, None] + pre[1].double()[None, :, None, None])
    ref = torch.nn.grad.conv2d_weight(xin, (Co, Ci, 3, 3), dy[:4].double(), padding=d, dilation=d)
    got = _lib.conv3x3_wgrad_x3(dy[:4].contiguous(), x[:4].contiguous(), d, pre=pre).double()
    got = got.reshape(ref.shape) if got.numel() == ref.numel() and got.shape != ref.shape else got
    print("%d->%d @%d d%d N%d%s: %.3f ms   max err vs float64 (4 images) %.2e  (|ref| max %.1f)" % (
        Ci, Co, H, d, N, " on load" if on_load else "", sorted(p.elapsed_time(q) for p, q in ev)[4], (got - ref).abs().max().item(), ref.abs().max().item()))
# stride 2 (k_wgrad3x3_s2): layer2 / layer3's first blocks
lib = _lib.load()
for (Co, Ci, Ho, N) in ((128, 128, 64, 144), (256, 256, 32, 144)):
    x = torch.randn(N, Ci, 2 * Ho, 2 * Ho, device="cuda")
    dy = torch.randn(N, Co, Ho, Ho, device="cuda")
    dw9 = torch.empty((9, Co, Ci), dtype=torch.float32, device="cuda")
    def f():
        rc = lib.aadg_conv3x3s2_wgrad_f32x3(dy.data_ptr(), x.data_ptr(), dw9.data_ptr(), N, Co, Ci, Ho, Ho, _lib._stream())
        assert rc == 0, rc
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
    for p, q in ev:
        p.record(); f(); q.record()
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x[:2].double(), (Co, Ci, 3, 3), dy[:2].double(), stride=2, padding=1)
    g9 = torch.empty_like(dw9)
    lib.aadg_conv3x3s2_wgrad_f32x3(dy[:2].contiguous().data_ptr(), x[:2].contiguous().data_ptr(), g9.data_ptr(), 2, Co, Ci, Ho, Ho, _lib._stream())
    got = g9.permute(1, 2, 0).reshape(Co, Ci, 3, 3).double()
    print("stride 2 %d->%d @%d N%d: %.3f ms   max err vs float64 (2 images) %.2e  (|ref| max %.1f)" % (
        Ci, Co, Ho, N, sorted(p.elapsed_time(q) for p, q in ev)[4], (got - ref).abs().max().item(), ref.abs().max().item()))
