"""Per-layer times of the float32-precision (f32x3) convolution kernels on the shapes of DeepLabV3+/ResNet-50 at N = 144 x 512 x 512
(BASELINE configs[1]): forward, input gradient and weight gradient of every distinct 1x1 / 3x3 convolution, with the two rooflines that
bound each -- HBM (float32 in + out once, 6.3 TB/s achievable) and the bfloat16 matrix cores at three products per multiply (2.5 PF / 3).

    python scripts/x3_layer_times.py [N] [--lib]       (--lib: also time the library's float32 convolution)
"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from aadg_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 144
LIB = "--lib" in sys.argv
# (name, count per step, kind, Co, Ci, H, W, dilation)
L = [
    ("l1.conv1 256->64", 2, "1x1", 64, 256, 128, 128, 1), ("l1.conv1 64->64", 1, "1x1", 64, 64, 128, 128, 1),
    ("l1.conv3 64->256", 3, "1x1", 256, 64, 128, 128, 1), ("l1.down 64->256", 1, "1x1", 256, 64, 128, 128, 1),
    ("l1.conv2 3x3 64", 3, "3x3", 64, 64, 128, 128, 1),
    ("l2.conv1 256->128 @128", 1, "1x1", 128, 256, 128, 128, 1), ("l2.conv1 512->128", 3, "1x1", 128, 512, 64, 64, 1),
    ("l2.conv3 128->512", 4, "1x1", 512, 128, 64, 64, 1), ("l2.down 256->512", 1, "1x1", 512, 256, 64, 64, 1),
    ("l2.conv2 3x3 128", 3, "3x3", 128, 128, 64, 64, 1),
    ("l3.conv1 512->256 @64", 1, "1x1", 256, 512, 64, 64, 1), ("l3.conv1 1024->256", 5, "1x1", 256, 1024, 32, 32, 1),
    ("l3.conv3 256->1024", 6, "1x1", 1024, 256, 32, 32, 1), ("l3.down 512->1024", 1, "1x1", 1024, 512, 32, 32, 1),
    ("l3.conv2 3x3 256", 5, "3x3", 256, 256, 32, 32, 1),
    ("l4.conv1 1024->512", 1, "1x1", 512, 1024, 32, 32, 1), ("l4.conv1 2048->512", 2, "1x1", 512, 2048, 32, 32, 1),
    ("l4.conv3 512->2048", 3, "1x1", 2048, 512, 32, 32, 1), ("l4.down 1024->2048", 1, "1x1", 2048, 1024, 32, 32, 1),
    ("l4.conv2 3x3 512 d2", 3, "3x3", 512, 512, 32, 32, 2),
    ("aspp 2048->256", 4, "1x1", 256, 2048, 32, 32, 1), ("aspp.project 1280->256", 1, "1x1", 256, 1280, 32, 32, 1),
    ("aspp.sep 256->256", 1, "1x1", 256, 256, 32, 32, 1), ("skip 256->48", 1, "1x1", 48, 256, 128, 128, 1),
    ("fuse 304->256", 1, "1x1", 256, 304, 128, 128, 1),
]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


def main():
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    floor = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("%-26s %3s | %-27s | %-27s | %-27s" % ("layer", "n", "fwd  ms   TB/s  TF(x1)", "dgrad ms  TB/s  TF(x1)", "wgrad ms  TB/s  TF(x1)"))
    for name, cnt, kind, Co, Ci, H, W, d in L:
        x = torch.randn(N, Ci, H, W, device="cuda")
        dy = torch.randn(N, Co, H, W, device="cuda")
        taps = 1 if kind == "1x1" else 9
        w = torch.randn(Co, Ci, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device="cuda") / (taps * Ci) ** 0.5
        flops = 2.0 * N * H * W * Co * Ci * taps
        byt = 4.0 * N * H * W * (Co + Ci)
        if kind == "1x1":
            a = _lib.split_weight(w.view(Co, Ci))
            at = _lib.split_weight(w.view(Co, Ci).t().contiguous())
            fns = {"fwd": lambda: _lib.conv1x1_nchw_x3(a, x), "dgrad": lambda: _lib.conv1x1_nchw_x3(at, dy),
                   "wgrad": lambda: _lib.conv1x1_wgrad_x3(dy, x)}
        else:
            a9 = _lib.split_weight(w.permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous())
            a9t = _lib.split_weight(w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, Ci, Co).contiguous())
            fns = {"fwd": lambda: _lib.conv3x3_nchw_x3(a9, x, d), "dgrad": lambda: _lib.conv3x3_nchw_x3(a9t, dy, d),
                   "wgrad": lambda: _lib.conv3x3_wgrad_x3(dy, x, d)}
        cols = []
        for k in ("fwd", "dgrad", "wgrad"):
            ms = timed(fns[k])
            tot[k] += cnt * ms
            fl = max(byt / 6.3e12, 3 * flops / 2.5e15) * 1e3
            floor[k] += cnt * fl
            cols.append("%6.3f %5.2f %6.0f (fl %5.3f)" % (ms, byt / ms / 1e9, flops / ms / 1e9, fl))
        line = "%-26s %3d | %s | %s | %s" % (name, cnt, cols[0], cols[1], cols[2])
        if LIB:
            pad = d if taps == 9 else 0
            line += " | lib fwd %6.3f" % timed(lambda: F.conv2d(x, w, padding=pad, dilation=d))
        print(line, flush=True)
        del x, dy
    print("per step (count-weighted): fwd %.1f ms (floor %.1f)  dgrad %.1f (%.1f)  wgrad %.1f (%.1f)" %
          (tot["fwd"], floor["fwd"], tot["dgrad"], floor["dgrad"], tot["wgrad"], floor["wgrad"]))


if __name__ == "__main__":
    main()
