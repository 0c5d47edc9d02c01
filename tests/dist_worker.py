"""Worker for tests/test_dist_cpu.py: world_size-2 gloo run of the row-sharded inner-loop exchange.

The HIP kernels cannot run on CPU, so in THIS TEST ONLY the materialising call is replaced by the oracle
(test infrastructure); what is under test is the distributed logic: identical batch plans on every rank, the
row-sharding law, the single all-gather of embeddings, bit-identical rewards on all ranks, list-style
all_gather / all_reduce helpers, policy broadcast."""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def oracle_materialize(pool, masks, units, crop, dataset, out_img=None, out_lbl=None):
    from oracle import oracle as O
    img, lbl = O.aug_units(pool.numpy(), masks.numpy(), units, crop, dataset)
    return torch.from_numpy(img), torch.from_numpy(lbl)


def build_batch(seed, D, B, M, size, crop):
    from helpers import Cfg, synth_pool
    from aadg_amd.data import transform as T
    from aadg_amd.data.basic import DevicePool
    from aadg_amd.data.policy import DGMultiPolicy, parse_policies
    rs = np.random.RandomState(seed)
    imgs, msks = synth_pool(rs, D * 2, size, size)
    pool = DevicePool(torch.from_numpy(imgs), torch.from_numpy(msks))
    pol = np.zeros((M, 20), np.int64)
    pol[:, 0::2] = rs.randint(0, 10, (M, 10))
    pol[:, 1::2] = rs.randint(0, 10, (M, 10))
    tf = T.Compose([DGMultiPolicy(parse_policies(pol, Cfg(M=M), None)), T.DGRandomScaleCrop(crop), T.Normalize_dg('optic'),
                    T.ToTensor('optic')])
    random.seed(seed)
    np.random.seed(seed)
    batch = []
    for _ in range(B):
        per_item = []
        for d in range(D):
            idx = int(np.random.choice(2, 1)[0])
            per_item.append(tf({'image': pool.image(2 * d + idx), 'label': pool.mask(2 * d + idx), 'img_name': 'x', 'dc': d}))
        batch.append(per_item)
    return batch


def embed(images):
    """Deterministic stand-in for backbone + discriminator: fixed random projection of pooled pixels."""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(3 * 16, 128, generator=g)
    pooled = torch.nn.functional.adaptive_avg_pool2d(images, 4).flatten(1)
    return torch.nn.functional.leaky_relu(pooled @ w, 0.2).contiguous()


def check_plan(rank, world, law, D, B, M, size, crop, full, O, adist, T):
    """One placement law: local rows, the single all-gather back into collate order, rewards identical on all ranks."""
    N = D * B * M
    T.set_row_shard(rank, world, law)
    part = T.train_dg_collate_fn(build_batch(11, D, B, M, size, crop))
    plan = part['plan']
    assert plan.n_rows == N and plan.world == world and plan.law == law and sum(plan.counts) == N
    rows = torch.from_numpy(plan.rows)
    assert part['aug_images'].shape[0] == plan.n_local == plan.counts[rank]
    assert torch.equal(part['aug_images'], full['aug_images'][rows])
    assert torch.equal(part['aug_labels'], full['aug_labels'][rows])
    assert torch.equal(part['dc'], full['dc'][rows])
    # every row has exactly one owner
    owners = adist.all_gather([torch.nn.functional.pad(rows, (0, plan.max_count - plan.n_local), value=-1)])[0]
    owned = owners[owners >= 0]
    assert owned.numel() == N and torch.equal(torch.sort(owned)[0], torch.arange(N))
    units = plan.units()
    if law == 'unit':
        U = D * M
        assert plan.n_local == len(units) * B                      # whole (domain, policy) units
        want = [U // world + (1 if r < U % world else 0) for r in range(world)]
        assert [c // B for c in plan.counts] == want               # 18 units over 4 ranks: 5/5/4/4
        for (d, j) in units:                                       # the rows of a unit: all B items of (domain d, policy j)
            assert all(((b * D + d) * M + j) in set(plan.rows.tolist()) for b in range(B))
    else:
        assert max(plan.counts) - min(plan.counts) <= 1            # balanced to the row
    if world == D:
        assert {d for d, _ in units} == {rank}                     # G == D: one source domain per GPU (north_star)
    lo_s, hi_s, S = part['image_rows']
    assert (lo_s, hi_s) == adist.shard_rows(S) and torch.equal(part['image'], full['image'][lo_s:hi_s])
    # the one exchange step: padded all-gather of the local embeddings, back into collate order, reward on the full matrix
    fe_local = embed(part['aug_images'])
    fe_all = plan.gather(fe_local)
    assert torch.equal(fe_all, embed(full['aug_images']))
    rewards = O.sinkhorn_rewards(fe_all.numpy(), D, B, M)
    truth = O.sinkhorn_rewards(embed(full['aug_images']).numpy(), D, B, M)
    assert np.array_equal(rewards, truth)
    gathered = adist.all_gather([torch.from_numpy(rewards)])[0].view(world, M)
    assert all(torch.equal(gathered[0], gathered[r]) for r in range(world))     # identical on every rank
    # gradient all-reduce: the package's reducer over gloo averages; local mean x n_local * G / N  ==  the global mean, uneven splits included
    from aadg_amd.reducer import GradReducer
    torch.manual_seed(5)
    lin = torch.nn.Linear(8, 1)
    ddp = GradReducer(lin)
    x = torch.arange(N * 8, dtype=torch.float32).view(N, 8) / 100.0
    (ddp(x[rows]).mean() * plan.loss_weight).backward()
    ref = torch.nn.Linear(8, 1)
    ref.load_state_dict(lin.state_dict())
    ref(x).mean().backward()
    assert torch.allclose(lin.weight.grad, ref.weight.grad, atol=1e-6) and torch.allclose(lin.bias.grad, ref.bias.grad, atol=1e-6)
    return part


def check_discriminator(rank, world, D, plan):
    """load_ddp_discriminator wraps the online branch in the gradient reducer: after an optimiser step on DIFFERENT local rows the parameters
    (online and, through momentum_update, EMA) are still identical on every rank."""
    from helpers import Cfg
    from aadg_amd.losses import CrossEntropy
    from aadg_amd.models import load_ddp_discriminator
    from aadg_amd import distributed as adist

    class A(object):
        gpu, workers, distributed = None, 0, True
    cfg = Cfg()
    cfg.DISCRIMINATOR = Cfg._C(); cfg.DISCRIMINATOR.NAME = 'momentum_feature'
    cfg.DATASET = Cfg._C(); cfg.DATASET.NAME = 'optic'; cfg.DATASET.DG = Cfg._C(); cfg.DATASET.DG.TRAIN = list(range(D))
    cfg.MODEL = Cfg._C(); cfg.MODEL.NAME = 'unet'; cfg.MODEL.BACKBONE = 'unet'
    cfg.TRAIN = Cfg._C(); cfg.TRAIN.BATCH_SIZE = 2
    torch.manual_seed(77 + rank)                       # deliberately different initial weights: the reducer broadcasts rank 0's
    disc, _, _ = load_ddp_discriminator(1, A(), cfg)
    from aadg_amd.reducer import GradReducer
    assert isinstance(disc, GradReducer)
    bare = disc.module
    bare.synchronize_parameters()
    opt = torch.optim.Adam([p for p in disc.parameters() if p.requires_grad], lr=1e-2)
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(plan.n_rows, 128, generator=g)
    soft = torch.softmax(torch.randn(plan.n_rows, D, generator=g), dim=1)
    rows = torch.from_numpy(plan.rows)
    for _ in range(2):
        loss = CrossEntropy()(disc(feats[rows], momentum=False), soft[rows]) * plan.loss_weight
        opt.zero_grad()
        loss.backward()
        opt.step()
        bare.momentum_update()
    flat = torch.cat([p.detach().reshape(-1) for p in bare.parameters()])
    every = adist.all_gather([flat[None]])[0]
    assert all(torch.equal(every[0], every[r]) for r in range(world)), "discriminator replicas diverged"
    # and they equal single-process training on ALL rows
    torch.manual_seed(77)
    from aadg_amd.models.discriminator import MomentumFeatureDiscriminator
    ref = MomentumFeatureDiscriminator(D, 128)
    for p in list(ref.mom_dis.parameters()) + list(ref.mom_fc.parameters()):
        p.requires_grad_(False)
    ref.synchronize_parameters()
    ropt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=1e-2)
    for _ in range(2):
        loss = CrossEntropy()(ref(feats, momentum=False), soft)
        ropt.zero_grad()
        loss.backward()
        ropt.step()
        ref.momentum_update()
    rflat = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    assert torch.allclose(flat, rflat, atol=2e-5), (flat - rflat).abs().max()


def check_reducer(rank, world, plan):
    """aadg_amd/reducer.py on a small network with BatchNorm buffers: several buckets (tiny capacity), both zero_grad modes, a
    parameter that takes no part, a backward pass that raises, buffers broadcast from rank 0 -- after every step the replicas are
    identical and equal to single-process training on ALL rows with the count-weighted loss."""
    from aadg_amd.reducer import GradReducer
    from aadg_amd import distributed as adist

    def make():
        torch.manual_seed(31)
        class Net(torch.nn.Sequential):
            def __init__(self):
                super().__init__(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                 torch.nn.Linear(16, 3))
                self.unused = torch.nn.Parameter(torch.ones(4, 4))      # requires a gradient, never gets one
        return Net()
    g = torch.Generator().manual_seed(4)
    N = plan.n_rows
    x, y = torch.randn(N, 6, generator=g), torch.randn(N, 3, generator=g)
    rows = torch.from_numpy(plan.rows)
    torch.manual_seed(1000 + rank)
    net = make()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(rank * 0.1)                          # replicas start different: construction broadcasts rank 0's
    assert adist.grad_group() is not None and adist.grad_group() is not adist.small_group()
    red = GradReducer(net, bucket_bytes=600)           # 16x16 floats = 1 KiB: several buckets
    assert red.stats["buckets"] >= 3
    ref = make()
    # BatchNorm over the LOCAL rows differs from the full batch: eval-mode statistics keep the comparison exact
    net[1].eval(); ref[1].eval()
    opt, ropt = torch.optim.SGD(red.parameters(), lr=0.1), torch.optim.SGD(ref.parameters(), lr=0.1)
    for step in range(4):
        none = step % 2 == 0
        opt.zero_grad(set_to_none=none); ropt.zero_grad(set_to_none=none)
        if step == 2:                                   # a pass cut short: its arrivals must not leak into the next one
            class Boom(torch.autograd.Function):
                @staticmethod
                def forward(ctx, t):
                    return t.clone()

                @staticmethod
                def backward(ctx, g):
                    raise ValueError("boom")
            try:
                (Boom.apply(red(x[rows])[:, :1]).sum() + red.module[5].weight.sum()).backward()
                raise AssertionError("expected the backward pass to raise")
            except ValueError:
                pass
            opt.zero_grad(set_to_none=none)
        (((red(x[rows]) - y[rows]) ** 2).mean() * plan.loss_weight).backward()
        ((ref(x) - y) ** 2).mean().backward()
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            if n.startswith("unused"):
                assert p.grad is None or not p.grad.any()
                continue
            assert p.grad is not None and torch.allclose(p.grad, q.grad, atol=2e-6), (step, n, (p.grad - q.grad).abs().max())
        opt.step(); ropt.step()
    assert red.stats["buckets"] >= 3 and red._rebuilt
    # two backward passes into one optimizer step (as DDP without no_sync(): each pass all-reduces, .grad accumulates); the second pass
    # leaves the first layers out -- their gradient is the first pass's average and must come through the second all-reduce unchanged
    for none in (True, False):
        opt.zero_grad(set_to_none=none); ropt.zero_grad(set_to_none=none)
        (((red(x[rows]) - y[rows]) ** 2).mean() * plan.loss_weight).backward()
        ((ref(x) - y) ** 2).mean().backward()
        z = torch.randn(N, 16, generator=torch.Generator().manual_seed(9))
        ((red.module[5](z[rows]) ** 2).mean() * plan.loss_weight).backward()              # the last layer only
        (ref[5](z) ** 2).mean().backward()
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            if not n.startswith("unused"):
                assert torch.allclose(p.grad, q.grad, atol=3e-6), ("two passes", none, n, (p.grad - q.grad).abs().max())
        opt.step(); ropt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    every = adist.all_gather([flat[None]])[0]
    assert all(torch.equal(every[0], every[r]) for r in range(world)), "replicas diverged"
    # buffers: a training forward starts from rank 0's running statistics
    net[1].train()
    with torch.no_grad():
        net[1].running_mean.fill_(float(rank))
    seen = []
    h = net[1].register_forward_pre_hook(lambda m, inp: seen.append(m.running_mean.clone()))
    red(x[rows])
    h.remove()
    assert torch.equal(seen[0], torch.zeros(16))        # every rank starts the forward from rank 0's buffer
    with torch.no_grad():
        net[1].running_mean.fill_(float(rank + 5))
    red.eval()
    red(x[rows])                                        # dirty after the training forward: one more broadcast, then none
    assert torch.equal(net[1].running_mean, torch.full((16,), 5.0))
    with torch.no_grad():
        net[1].running_mean.fill_(float(rank))
    red(x[rows])
    assert torch.equal(net[1].running_mean, torch.full((16,), float(rank)))


def worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aadg_amd import _lib
    from aadg_amd import distributed as adist
    from aadg_amd.data import transform as T
    from oracle import oracle as O
    _lib.aug_u8_forward = oracle_materialize          # test-only: CPU checker stands in for the HIP launch
    D, B, M, size, crop = 3, 2, 6, 24, 24             # 18 (domain, policy) units: uneven over 4 ranks (5/5/4/4)
    if world > D * B:                                  # 8 ranks (3/3/2/2/2/2/2/2 units): the un-augmented images need a row per rank
        B = 3
    # single-process truth (every rank computes it: shard = whole)
    T.set_row_shard(0, 1)
    full = T.train_dg_collate_fn(build_batch(11, D, B, M, size, crop))
    assert not full['plan'].sharded and full['aug_images'].shape[0] == D * B * M
    for law in ('unit', 'row'):
        part = check_plan(rank, world, law, D, B, M, size, crop, full, O, adist, T)
    check_discriminator(rank, world, D, part['plan'])
    check_reducer(rank, world, part['plan'])
    # test batches are NOT sharded: every rank scores the whole test set (validate() then agrees everywhere)
    T.set_row_shard(rank, world, 'unit')
    tb = build_batch(11, D, B, M, size, crop)
    items = [{k: v for k, v in item[0].items() if k not in ('aug_images', 'aug_labels', 'dc_single')} for item in tb]
    for it in items:
        it['dc'] = it['dc'][0]
    test = T.test_dg_collate_fn(items)
    assert test['image'].shape[0] == B and 'plan' not in test
    # list helpers
    t = [torch.full((3,), float(rank + 1)), torch.full((2, 2), float(rank))]
    adist.all_reduce(t, average=True)
    assert torch.allclose(t[0], torch.full((3,), (1 + world) / 2.0)) and torch.allclose(t[1], torch.full((2, 2), (world - 1) / 2.0))
    # replicated controller: rank 0's policies are authoritative
    from helpers import Cfg
    from aadg_amd.models.controller import Controller
    cfg = Cfg(M=4)
    cfg.CONTROLLER.T, cfg.CONTROLLER.C = 2, 2.5
    torch.manual_seed(1023)
    ctrl = Controller(cfg)
    torch.manual_seed(100 + rank)                      # deliberately different draws per rank
    M = 4
    policies, _, _, log_probs, _ = ctrl(M)
    dist.broadcast(policies, 0)
    lp = ctrl.evaluate(policies, M)
    all_lp = adist.all_gather([lp.detach()])[0].view(world, M)
    assert all(torch.allclose(all_lp[0], all_lp[r], atol=1e-6) for r in range(world))
    # round 4: the small collectives' own process group (a second communicator over the same ranks), the policy broadcast and the
    # BatchNorm-statistics all-reduce through it, what bench.py prints about the job, and the GPU-side timer switched off on CPU
    g = adist.small_group()
    assert g is not None and g is adist.small_group() and g is not dist.group.WORLD
    d = adist.describe()
    assert d["initialized"] and d["world_size"] == world and d["rank"] == rank and d["backend"] == "gloo"
    assert "own process group" in d["small_collectives_group"]
    v = torch.full((5,), float(rank + 1), dtype=torch.float64)
    adist.small_all_reduce(v, kind="batchnorm_statistics_all_reduce")
    assert torch.equal(v, torch.full((5,), world * (world + 1) / 2.0, dtype=torch.float64))
    p2 = torch.full((3,), rank, dtype=torch.int64)
    adist.small_broadcast(p2, 0, kind="policy_broadcast")
    assert torch.equal(p2, torch.zeros(3, dtype=torch.int64))
    from aadg_amd import _lib as L
    before = L.BN_SYNC_COLLECTIVES[0]
    w = torch.ones(4, dtype=torch.float64) * (rank + 1)
    L._bn_sync_reduce(w)                                # what the BatchNorm kernels' wrapper calls between its two phases
    assert torch.equal(w, torch.full((4,), world * (world + 1) / 2.0, dtype=torch.float64)) and L.BN_SYNC_COLLECTIVES[0] == before + 1
    adist.USE_SMALL_GROUP = False
    assert adist.small_group() is None                  # opt-out: everything on the default group
    adist.USE_SMALL_GROUP = True
    open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    dist.destroy_process_group()
    adist.reset_groups()


if __name__ == "__main__":
    worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
