import copy, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from aadg_amd import _lib as hip
from aadg_amd.models import deeplab
from test_gpu_syncbn import Jacobi
hip.load()
torch.manual_seed(5)
enc = deeplab.ResNet50Encoder()
stage = enc.layer1.cuda()
for m in stage.modules():
    if isinstance(m, (deeplab.Conv1x1, deeplab.Conv3x3)):
        m.f32x3 = True
deeplab.mark_bn_producers(stage)
stage.train()
x = torch.randn(5, 64, 32, 32, device="cuda"); g = torch.randn(5, 256, 32, 32, device="cuda"); cuts = [0, 3, 5]
def run(mod, xin, gout):
    xin = xin.detach().clone().requires_grad_(True)
    y = mod(xin); y = y[0] if isinstance(y, tuple) else y
    y.backward(gout)
    return y.detach(), xin.grad.detach()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
B = deeplab.Bottleneck
for name, flags in (("all off", (False, False, False, False)), ("lazy only", (True, True, False, False)), ("pair only", (False, False, True, True)), ("all on", (True, True, True, True))):
    B.lazy_bn1, B.lazy_bn2, B.pair_shortcut_bn, B.lazy_shortcut = flags
    full = copy.deepcopy(stage)
    y_full, dx_full = run(full, x, g)
    y2, dx2 = run(copy.deepcopy(stage), x, g)
    fake = Jacobi(2); deeplab.set_bn_sync(True); hip.BN_SYNC_REDUCE = fake
    hist = []
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 26):
        fake.next_pass()
        outs = []
        for r in range(2):
            fake.start(r)
            outs.append(run(copy.deepcopy(stage), x[cuts[r]:cuts[r + 1]], g[cuts[r]:cuts[r + 1]]))
        y = torch.cat([o[0] for o in outs]); dx = torch.cat([o[1] for o in outs])
        hist.append((rel(y, y_full), rel(dx, dx_full)))
    hip.BN_SYNC_REDUCE = None; deeplab.set_bn_sync(False)
    print(name, "repeat-noise y %.1e dx %.1e" % (rel(y2, y_full), rel(dx2, dx_full)), "| sync vs full:", " ".join("%.0e/%.0e" % h for h in hist[::3]), flush=True)
