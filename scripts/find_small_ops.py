"""Ad-hoc: which Python lines issue the small ATen launches of one search step (fill_ / zero_ / copy_ / add ...): one step under
torch.profiler with stacks, grouped by (op, innermost repo frame).  usage: python scripts/find_small_ops.py [--shard_of 8] [op substrings...]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

argv = sys.argv[1:]
shard = []
if "--shard_of" in argv:
    i = argv.index("--shard_of"); shard = argv[i:i + 2]; del argv[i:i + 2]
wanted = argv or ["fill_", "zero_", "zeros"]
sys.argv = [sys.argv[0], "--legs", "none"] + shard
a = bench.parse()
cfg, st = bench.build_state(a, 0, 1)
for i in range(3):
    st.search_step(i, max_iters=1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    st.search_step(3, max_iters=1)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    kn = [k.name for k in (e.kernels or [])]
    if not (any(w in e.name for w in wanted) or any(w in k for k in kn for w in wanted)):
        continue
    frame = "?"
    for s in e.stack or []:
        if "/aadg_amd/" in s or "bench.py" in s:
            frame = s.replace(ROOT + "/", "")
            break
    cnt[(e.name + (" -> " + kn[0][:40] if kn else ""), frame)] += 1
for (name, frame), n in cnt.most_common(40):
    print("%4d  %-70s %s" % (n, name, frame))
