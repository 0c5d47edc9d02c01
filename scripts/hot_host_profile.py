"""Where the host time of the hot-path step goes (bench.py: hot_step without the backbone): cProfile of 200 steps."""
import cProfile, os, pstats, random, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from aadg_amd import _lib
from aadg_amd.data import transform as T
from aadg_amd.data.policy import DGMultiPolicy, parse_policies
a = bench.Args()
a.cfg, a.backbone, a.batch, a.size = os.path.join("experiments", "optic_sinkhorn", "diversity.yaml"), "resnet50", 8, 512
a.backbone_dtype, a.no_sync_bn, a.placement, a.no_dropout, a.force_dist = "bf16", True, "row", False, False
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    cfg, st = bench.build_state(a, 0, 1)
M, D = st.M, 3
plan = T.row_plan(D, a.batch, M)
z = torch.randn(plan.n_local, 2, a.size, a.size, device="cuda")
fe = torch.nn.functional.leaky_relu(torch.randn(D * a.batch * M, 128, device="cuda"), 0.2)
rewards = torch.zeros(M, device="cuda")
state = {}
def hot_step():
    nxt, state['p'] = state.get('p'), None
    if nxt is None:
        policies, _, _, log_probs, entropies = st.graphed.sample()
        host = policies.cpu().numpy()
    else:
        policies, _, _, log_probs, entropies, fetch = nxt
        host = fetch()
    parsed = parse_policies(host, cfg, None)
    st.train_loader.dataset.transforms.transforms[0] = DGMultiPolicy(parsed)
    sample = next(iter(st.train_loader))
    main = torch.cuda.current_stream()
    side = st._controller_stream(main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        rewards.zero_()
        _lib.sinkhorn_rewards(fe, D, a.batch, M, rewards=rewards)
        st.graphed.update(_lib.normalize_rewards(rewards), entropies)
        state['p'] = st._sample_policies(async_host=True)
    _lib.seg_bce_dice(z, sample['aug_labels'], M, want_grad=True)
    main.wait_stream(side)
    st.train_loader.predraw(fresh_policies=True)
for _ in range(5): hot_step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(100): hot_step()
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) * 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): hot_step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
