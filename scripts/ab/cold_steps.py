"""Wall time of the first policy-search steps of a fresh process (allocator pools, side-stream pool, kernel caches warming up):
    python scripts/ab/cold_steps.py [steps] [--no_wgrad_stream]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 15
for f in (random.seed, np.random.seed, torch.manual_seed):
    f(1023)
a = bench.Args()
a.cfg, a.backbone, a.batch, a.size = os.path.join("experiments", "optic_sinkhorn", "diversity.yaml"), "resnet50", 8, 512
a.backbone_dtype, a.no_sync_bn, a.placement, a.no_dropout = "f32x3", True, "row", False
a.no_wgrad_stream = "--no_wgrad_stream" in sys.argv
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    cfg, st = bench.build_state(a, 0, 1)
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.search_step(i, max_iters=1)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("side" if not a.no_wgrad_stream else "inline", [round(t, 1) for t in ts], "reserved GB %.1f" % (torch.cuda.memory_reserved() / 2**30))
