"""Run-to-run spread of the quantities tests/test_gpu_precision.py::test_bf16_backbone_keeps_the_search_quantities bounds (the warm-up
steps are not bit-reproducible: split-K float atomics).   python scripts/ab/precision_spread.py [backbone] [runs]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import contextlib
import bench
backbone = sys.argv[1] if len(sys.argv) > 1 else "mobilenet_v2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for r in range(runs):
    a = bench.Args()
    a.cfg, a.backbone, a.batch, a.size = os.path.join("experiments", "optic_sinkhorn", "diversity.yaml"), backbone, 4, 256
    a.backbone_dtype, a.no_sync_bn, a.placement, a.no_dropout = "bf16", True, "row", True
    with contextlib.redirect_stdout(sys.stderr):
        cfg, st = bench.build_state(a, 0, 1)
        for i in range(5):
            st.search_step(i, max_iters=1)
        _, st32 = bench.build_state(a, 0, 1, backbone_dtype="fp32")
        p = bench.precision_check(st, st32, st.M, len(cfg.DATASET.DG.TRAIN), 4)
    rows.append((float(np.median([b["reward_rel"] for b in p["per_batch"]])), p["reward_rel_max_diff"], p["bce_rel_max_diff"], p["dice_abs_max_diff"]))
    print("run %2d: reward_rel median %.4f max %.4f | bce_rel max %.5f | dice_abs max %.5f" % ((r,) + rows[-1]), flush=True)
m = np.array(rows)
print("max over runs:", m.max(0), " 90th percentile:", np.percentile(m, 90, axis=0))
