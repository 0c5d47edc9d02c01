"""Ad-hoc: input gradient of a bottleneck's first 1x1 convolution fused with the residual-branch gradient:
   (a) library bwd-data + in-place add   vs   (b) one baddbmm (beta = 1) per block shape; each variant in its own process."""
import subprocess, sys
CHILD = r'''
import sys, time, torch
N, Ci, Cm, S, which = [int(v) for v in sys.argv[1:5]] + [sys.argv[5]]
torch.manual_seed(0)
g1 = torch.randn(N, Cm, S, S, device="cuda", dtype=torch.bfloat16)        # gradient w.r.t. conv1's output
gid = torch.randn(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)       # gradient of the identity branch
w = (torch.randn(Cm, Ci, 1, 1, device="cuda") * 0.05).bfloat16()
x = torch.empty(N, Ci, S, S, device="cuda", dtype=torch.bfloat16)
def lib():
    t = torch.ops.aten.convolution_backward(g1, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    return t.add_(gid)
wt = w.view(Cm, Ci).t().contiguous()                                       # [Ci, Cm]
def fused():
    return torch.baddbmm(gid.view(N, Ci, S * S), wt.unsqueeze(0).expand(N, Ci, Cm), g1.view(N, Cm, S * S)).view(N, Ci, S, S)
fn = lib if which == "lib" else fused
ref = (torch.einsum("mc,nmk->nck", w.view(Cm, Ci).float(), g1.float().view(N, Cm, -1)) + gid.float().view(N, Ci, -1)).view(N, Ci, S, S)
out = fn(); torch.cuda.synchronize()
err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
for _ in range(2): fn()
torch.cuda.synchronize(); t = time.time()
for _ in range(5): fn()
torch.cuda.synchronize()
print("%-5s Ci=%4d Cm=%3d %3dx%-3d  %.3f ms  rel err %.1e" % (which, Ci, Cm, S, S, (time.time() - t) / 5 * 1e3, err), flush=True)
'''
for Ci, Cm, S in [(256, 64, 128), (512, 128, 64), (1024, 256, 32), (2048, 512, 32)]:
    for which in ("lib", "fused"):
        r = subprocess.run([sys.executable, "-c", CHILD, "144", str(Ci), str(Cm), str(S), which], capture_output=True, text=True, timeout=300)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "no output"
        print(line + ("" if r.returncode == 0 else "  [rc=%d]" % r.returncode), flush=True)
