"""Ad-hoc: time DeepLabV3+/R50 fwd+bwd at N=144x512x512 under different MIOpen / layout settings."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from aadg_amd.models.deeplab import DeepLabV3Plus
cl = os.environ.get("CL") == "1"
torch.backends.cudnn.benchmark = os.environ.get("BENCH") == "1"
m = DeepLabV3Plus("resnet50", 2).cuda()
x = torch.randn(int(os.environ.get("NB", "144")), 3, 512, 512, device="cuda")
if cl:
    m = m.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def it():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, f = m(x)
    loss = y.float().mean() + f.float().mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
t0 = time.time(); it(); it(); torch.cuda.synchronize(); t1 = time.time()
for _ in range(3): it()
torch.cuda.synchronize(); t2 = time.time()
print("RESULT warm=%%.1fs iter=%%.1fms mem=%%.1fGB" %% (t1 - t0, (t2 - t1) / 3 * 1e3, torch.cuda.max_memory_allocated() / 2**30))
''' % ROOT

for name, env in [("nchw_immediate", dict(CL="0", BENCH="0")),
                  ("cl_immediate", dict(CL="1", BENCH="0")),
                  ("cl_find_fast", dict(CL="1", BENCH="1", MIOPEN_FIND_MODE="FAST")),
                  ("cl_find_hybrid", dict(CL="1", BENCH="1", MIOPEN_FIND_MODE="DYNAMIC_HYBRID"))]:
    e = dict(os.environ); e.update(env)
    t0 = time.time()
    try:
        out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=420)
        res = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(name, res[0] if res else ("FAILED: " + out.stderr[-400:]), "total %.0fs" % (time.time() - t0), flush=True)
    except subprocess.TimeoutExpired:
        print(name, "TIMEOUT >420s", flush=True)
