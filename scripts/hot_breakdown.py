"""Ad-hoc: wall-clock breakdown of the backbone-free hot step (same pieces as bench.py's hot_step)."""
import os, sys, time, random, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from aadg_amd import _lib
from aadg_amd.data import transform as T
from aadg_amd.data.policy import DGMultiPolicy, parse_policies

class A: pass
a = A(); a.size, a.batch, a.backbone, a.backbone_dtype, a.sync_bn = 512, 8, "mobilenet_v2", "bf16", False
torch.cuda.set_device(0)
for f in (random.seed, np.random.seed, torch.manual_seed): f(1023)
with contextlib.redirect_stdout(sys.stderr):
    cfg, st = bench.build_state(a, 0, 1)
M, D = st.M, 3
n_rows = D * a.batch * M
z = torch.randn(n_rows, 2, a.size, a.size, device="cuda", requires_grad=True)
fe = torch.nn.functional.leaky_relu(torch.randn(n_rows, 128, device="cuda"), 0.2)
rewards = torch.zeros(M, device="cuda")
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
for it in range(25):
    if it == 5: acc.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    policies, _, _, log_probs, entropies = st.graphed.sample(); t = tick("1 controller sample (graph)", t)
    pol = policies.cpu().numpy(); t = tick("2 policies D2H", t)
    parsed = parse_policies(pol, cfg, None)
    st.train_loader.dataset.transforms.transforms[0] = DGMultiPolicy(parsed); t = tick("3 parse + inject", t)
    order = np.random.permutation(len(st.train_loader.dataset))
    items = [st.train_loader.dataset[int(i)] for i in order[:a.batch]]; t = tick("4 host draw (24 samples x 7 refs)", t)
    batch, refs, _ = T.collect_refs(items, True)
    units = T.refs_to_units(refs); t = tick("5 refs -> unit records", t)
    img, lbl = T.materialize(refs); t = tick("6 aug kernels (+validate, H2D)", t)
    loss, _, _ = _lib.policy_bce_loss(z, lbl[24:], M); t = tick("7 BCE/Dice kernel fwd", t)
    loss.backward(); t = tick("8 BCE backward (autograd)", t)
    rewards.zero_(); _lib.sinkhorn_rewards(fe, D, a.batch, M, rewards=rewards); nr = _lib.normalize_rewards(rewards); t = tick("9 sinkhorn + normalise", t)
    st.graphed.update(nr, entropies); t = tick("10 PPO update (graph)", t)
tot = sum(acc.values())
for k in sorted(acc, key=lambda s: int(s.split()[0])):
    print("%-40s %7.3f ms" % (k, acc[k] / 20 * 1e3))
print("%-40s %7.3f ms" % ("total (serialised by syncs)", tot / 20 * 1e3))
