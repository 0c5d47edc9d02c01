"""Ad-hoc: a few fwd+bwd iterations of DeepLabV3+/R50 at N x 512 x 512 for rocprofv3 (layout from env CL=0/1)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aadg_amd.models.deeplab import DeepLabV3Plus
cl = os.environ.get("CL") == "1"
nb = int(os.environ.get("NB", "144"))
m = DeepLabV3Plus("resnet50", 2).cuda()
x = torch.randn(nb, 3, 512, 512, device="cuda")
if cl:
    m = m.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def it():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, f = m(x)
    loss = y.float().mean() + f.float().mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
t0 = time.time(); it(); torch.cuda.synchronize(); t1 = time.time()
it(); torch.cuda.synchronize()
t2 = time.time()
for _ in range(3): it()
torch.cuda.synchronize(); t3 = time.time()
print("RESULT cl=%d warm=%.1fs iter=%.1fms mem=%.1fGB" % (cl, t1 - t0, (t3 - t2) / 3 * 1e3, torch.cuda.max_memory_allocated() / 2**30), flush=True)
