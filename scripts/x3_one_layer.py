"""One f32x3 convolution kernel on one shape, a few launches (target of rocprofv3 --pmc runs): python scripts/x3_one_layer.py 1x1|3x3 Co Ci H W [d] [N]"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from aadg_amd import _lib  # noqa: E402
kind, Co, Ci, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
d = int(sys.argv[6]) if len(sys.argv) > 6 else 1
N = int(sys.argv[7]) if len(sys.argv) > 7 else 144
x = torch.randn(N, Ci, H, W, device="cuda")
dy = torch.randn(N, Co, H, W, device="cuda")
if kind == "1x1":
    w = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
    a = _lib.split_weight(w)
    for _ in range(3):
        _lib.conv1x1_nchw_x3(a, x)
        _lib.conv1x1_wgrad_x3(dy, x)
else:
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (9 * Ci) ** 0.5
    a9 = _lib.split_weight(w.permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous())
    for _ in range(3):
        _lib.conv3x3_nchw_x3(a9, x, d)
        _lib.conv3x3_wgrad_x3(dy, x, d)
torch.cuda.synchronize()
