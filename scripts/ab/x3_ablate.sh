# timing-only ablations of a f32x3 kernel on one layer shape (results are WRONG in the variants):  bash scripts/ab/x3_ablate.sh "<layer args>" tag...
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
cat > /tmp/x3_abl.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from aadg_amd import _lib
kind, Co, Ci, H, W, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
N = 144
x = torch.randn(N, Ci, H, W, device="cuda")
w = torch.randn(Co, Ci, 3, 3, device="cuda") / (9 * Ci) ** 0.5
a9 = _lib.split_weight(w.permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous())
f = lambda: _lib.conv3x3_nchw_x3(a9, x, d)
f(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
for a, b in ev:
    a.record(); f(); b.record()
torch.cuda.synchronize()
print("%.3f ms" % sorted(a.elapsed_time(b) for a, b in ev)[3])
PY
echo -n "tree: "; python /tmp/x3_abl.py $ARGS
for t in "$@"; do echo -n "$t: "; AADG_LIB_PATH=exp_libs/$t.so PYTHONPATH=scripts/ab/hook:$PYTHONPATH python /tmp/x3_abl.py $ARGS; done
echo -n "tree again: "; python /tmp/x3_abl.py $ARGS
