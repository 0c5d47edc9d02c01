"""Ad-hoc: DeepLabV3+/R50 fwd+bwd with the bottleneck 3x3 convolutions fed channels_last tensors (layout conversions as
explicit torch copies): how much MIOpen time (implicit-GEMM + its own transposes) does NHWC-native execution save?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd.models import deeplab
from aadg_amd.models.deeplab import bn_act
cl = os.environ.get("CL2") == "1"
def fwd(self, x):
    idt = x if self.downsample is None else self.downsample(x)
    out = bn_act(self.bn1, self.conv1(x), 'relu')
    if cl:
        out = out.contiguous(memory_format=torch.channels_last)
        out = self.conv2(out).contiguous()
    else:
        out = self.conv2(out)
    out = bn_act(self.bn2, out, 'relu')
    return bn_act(self.bn3, self.conv3(out), 'relu', residual=idt)
deeplab.Bottleneck.forward = fwd
N = int(os.environ.get("NB", "144"))
m = deeplab.DeepLabV3Plus("resnet50", 2).cuda()
x = torch.randn(N, 3, 512, 512, device="cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def it():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, f = m(x)
    loss = y.float().mean() + f.float().mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
t0 = time.time(); it(); torch.cuda.synchronize(); t1 = time.time()
it(); torch.cuda.synchronize(); t2 = time.time()
for _ in range(3): it()
torch.cuda.synchronize(); t3 = time.time()
print("RESULT cl2=%d warm=%.1fs iter=%.1fms" % (cl, t1 - t0, (t3 - t2) / 3 * 1e3), flush=True)
