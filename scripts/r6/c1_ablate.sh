# timing-only ablations of the f32x3 1x1 kernel (results WRONG in the variants): bash scripts/r6/c1_ablate.sh tag...
cd $GRAFT_REPO_ROOT
cat > /tmp/c1_abl.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from aadg_amd import _lib
N = 144
for (Co, Ci, H) in ((2048, 512, 32), (512, 2048, 32), (1024, 256, 32), (256, 2048, 32), (256, 64, 128)):
    x = torch.randn(N, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
    a = _lib.split_weight(w)
    f = lambda: _lib.conv1x1_nchw_x3(a, x)
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for p, q in ev:
        p.record(); f(); q.record()
    torch.cuda.synchronize()
    print("%d->%d@%d %.3f" % (Ci, Co, H, sorted(p.elapsed_time(q) for p, q in ev)[3]), end="  ")
print()
PY
echo -n "tree: "; python /tmp/c1_abl.py 2>/dev/null
for t in "$@"; do echo -n "$t: "; AADG_LIB_PATH=exp_libs/$t.so PYTHONPATH=scripts/ab/hook:$PYTHONPATH python /tmp/c1_abl.py 2>/dev/null; done
echo -n "tree again: "; python /tmp/c1_abl.py 2>/dev/null
