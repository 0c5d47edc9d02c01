import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aadg_amd import _lib as hip
from aadg_amd.models import deeplab
hip.load()
torch.manual_seed(3)
model = deeplab.DeepLabV3Plus('resnet50', 2, aux_pooling=True).cuda()
deeplab.batch_step_bookkeeping(model, f32x3=True)
model.train()
for m in model.modules():
    if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)): m.p = 0.0
x = torch.randn(6, 3, 128, 128, device="cuda")
def run(mode):
    hip.set_wgrad_stream(mode)
    model.zero_grad(set_to_none=True)
    logits, feat = model(x)
    (logits.float().square().mean() + feat.float().mean()).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters()}
a = run(False); b = run(False); c = run(True); d = run(True)
def worst(u, v):
    w = max(((float((u[n]-v[n]).abs().max()) / (float(u[n].abs().max())+1e-12)), n) for n in u)
    return w
print("inline vs inline", worst(a, b)); print("inline vs side", worst(a, c)); print("side vs side", worst(c, d))
for n in ['encoder.stem.0.weight', 'decoder.block1.0.weight' if 'decoder.block1.0.weight' in a else list(a)[-3]]:
    print(n, float((a[n]-b[n]).abs().max()), float((a[n]-c[n]).abs().max()), float(a[n].abs().max()))
