"""CPU suite, part 2: host logic (controller, PPO/REINFORCE, discriminator, config, sharding law)
against golden vectors produced by the reference (tests/golden/make_golden.py: gen_controller)."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, Cfg


def _cfg(excl):
    cfg = Cfg(EXCLUDE_OPS=excl)
    cfg.CONTROLLER.T, cfg.CONTROLLER.C, cfg.CONTROLLER.PENALTY, cfg.CONTROLLER.LOSS = 2, 2.5, 1e-5, 'ppo'
    return cfg


def _load(tag, z, controller):
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "_sd_")}
    controller.load_state_dict(sd, strict=True)  # same key names as the reference's state_dict


@pytest.mark.parametrize("tag,excl", [("full", []), ("excl", ["Cutout"])])
def test_controller_and_search_losses_match_reference(tag, excl):
    from aadg_amd.models.controller import Controller
    from aadg_amd import losses
    z = np.load(os.path.join(GOLDEN, "controller.npz"))
    cfg = _cfg(excl)
    c = Controller(cfg)
    assert sum(p.numel() for p in c.parameters()) == sum(z[k].size for k in z.files if k.startswith(tag + "_sd_"))
    _load(tag, z, c)
    # same torch seed -> same multinomial draws -> the very same policies (CPU generator)
    torch.manual_seed(7)
    policies, op_probs, mag_probs, log_probs, entropies = c(6)
    assert policies.dtype == torch.int64 and tuple(policies.shape) == (6, 20)
    assert np.array_equal(policies.numpy(), z[tag + "_policies"])           # bit-exact policy indexing
    assert np.allclose(op_probs.detach().numpy(), z[tag + "_op_probs"], atol=1e-6)
    assert np.allclose(mag_probs.detach().numpy(), z[tag + "_mag_probs"], atol=1e-6)
    assert np.allclose(log_probs.detach().numpy(), z[tag + "_log_probs"], atol=1e-5)
    assert np.allclose(entropies.detach().numpy(), z[tag + "_entropies"], atol=1e-5)
    ev = c.evaluate(torch.from_numpy(z[tag + "_policies"]), 6)
    assert np.allclose(ev.detach().numpy(), z[tag + "_evaluate"], atol=1e-5)
    assert np.allclose(ev.detach().numpy(), log_probs.detach().numpy(), atol=1e-5)  # evaluate == sample log-prob
    reward = torch.from_numpy(z[tag + "_reward"])
    opt = torch.optim.Adam(c.parameters(), lr=0.00035)
    crit = losses.search_loss(cfg)
    assert isinstance(crit, losses.ProximalPolicyOptimization)
    crit.register_optimizer(opt)
    loss, score, ent = crit(c, policies, log_probs, entropies, reward)
    assert np.allclose([loss.item(), score.item(), ent.item()], z[tag + "_ppo"], atol=1e-5)
    assert np.allclose(c.evaluate(policies, 6).detach().numpy(), z[tag + "_ppo_evaluate_after"], atol=1e-4)
    # REINFORCE
    c2 = Controller(cfg)
    _load(tag, z, c2)
    torch.manual_seed(7)
    policies, _, _, log_probs, entropies = c2(6)
    cfg.CONTROLLER.LOSS = 'reinforce'
    crit2 = losses.search_loss(cfg)
    crit2.register_optimizer(torch.optim.Adam(c2.parameters(), lr=0.00035))
    loss, score, ent = crit2(c2, policies, log_probs, entropies, reward)
    assert np.allclose([loss.item(), score.item(), ent.item()], z[tag + "_reinforce"], atol=1e-5)
    assert np.allclose(c2.evaluate(policies, 6).detach().numpy(), z[tag + "_reinforce_evaluate_after"], atol=1e-4)


def test_momentum_discriminator_and_soft_ce_match_reference():
    from aadg_amd.models.discriminator import MomentumFeatureDiscriminator
    from aadg_amd.losses import CrossEntropy
    z = np.load(os.path.join(GOLDEN, "controller.npz"))
    d = MomentumFeatureDiscriminator(3, 64)
    d.load_state_dict({k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("disc_sd_")}, strict=True)
    x = torch.from_numpy(z["disc_x"])
    logits, fe = d(x, momentum=True, return_feature=True)
    assert not fe.requires_grad
    assert np.allclose(fe.numpy(), z["disc_mom_fe"], atol=1e-6) and np.allclose(logits.numpy(), z["disc_mom_logits"], atol=1e-6)
    assert np.allclose(d(x).detach().numpy(), z["disc_logits"], atol=1e-6)
    ce = CrossEntropy()(d(x), torch.from_numpy(z["ce_target"]))
    assert abs(ce.item() - float(z["ce_value"])) < 1e-6
    d.momentum_update()
    _, fe2 = d(x, momentum=True, return_feature=True)
    assert np.allclose(fe2.numpy(), z["disc_mom_fe_after_update"], atol=1e-6)
    # synchronize_parameters aliases the EMA twin to the online weights (reference quirk, kept)
    d.synchronize_parameters()
    with torch.no_grad():
        d.dis[0].weight.add_(1.0)
    assert torch.equal(d.dis[0].weight, d.mom_dis[0].weight)


def test_reference_style_yaml_configs_load_unchanged():
    from aadg_amd.config.defaults import get_default_config, update_config

    class Args:
        output_dir, seed = "out", 1023

    root = os.path.join(os.path.dirname(GOLDEN), "..", "experiments")
    files = sorted(glob.glob(os.path.join(root, "*", "*.yaml")))
    assert files, "experiments/*.yaml missing"
    for f in files:
        cfg = get_default_config()
        a = Args()
        a.cfg = f
        update_config(cfg, a)
        assert cfg.is_frozen() and cfg.SEED == 1023 and cfg.OUTPUT_DIR == "out"
        assert cfg.CONTROLLER.M == (1 if "fixed" in f else 6) and cfg.DATASET.NAME in ("optic", "rvs")
        with pytest.raises(AttributeError):
            cfg.SEED = 1
        cfg.CONTROLLER.EXCLUDE_OPS.append("x")  # list mutation is allowed on a frozen node, as in yacs
    bad = get_default_config()
    with pytest.raises(KeyError):
        bad._merge({"NOT_A_KEY": 1}, [])


def test_shard_rows_law():
    from aadg_amd.distributed import shard_rows
    for G in (1, 2, 3, 4, 5, 6, 8):
        spans = [shard_rows(144, r, G) for r in range(G)]
        assert spans[0][0] == 0 and spans[-1][1] == 144
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_row_plan_placement_laws():
    """domain-major (domain, policy) units cut contiguously -- G = 3: one domain per GPU; G = 4: 5/5/4/4 units;
    G = 8: 3/3/2/2/2/2/2/2 units ('unit') or 18 rows each ('row'); `take` restores collate order; weights sum to G."""
    import numpy as np
    from aadg_amd.distributed import RowPlan
    D, B, M = 3, 8, 6
    N = D * B * M
    one = RowPlan(D, B, M)
    assert not one.sharded and np.array_equal(one.rows, np.arange(N)) and one.loss_weight == 1.0
    for law in ('unit', 'row'):
        for G in (2, 3, 4, 8):
            plans = [RowPlan(D, B, M, r, G, law) for r in range(G)]
            rows = np.concatenate([p.rows for p in plans])
            assert np.array_equal(np.sort(rows), np.arange(N))
            assert abs(sum(p.loss_weight for p in plans) - G) < 1e-12
            # emulate the padded all-gather: rank r's rows land at r * max_count + i
            flat = np.full(G * plans[0].max_count, -1, np.int64)
            for r, p in enumerate(plans):
                flat[r * p.max_count:r * p.max_count + p.n_local] = p.rows
            assert np.array_equal(flat[plans[0].take], np.arange(N))
            if G == 3:
                for r, p in enumerate(plans):
                    assert {d for d, _ in p.units()} == {r}            # one source domain per GPU
                    assert np.array_equal(np.sort(p.rows) // M % D, np.full(p.n_local, r))
    assert [c // B for c in RowPlan(D, B, M, 0, 4, 'unit').counts] == [5, 5, 4, 4]
    assert [c // B for c in RowPlan(D, B, M, 0, 8, 'unit').counts] == [3, 3, 2, 2, 2, 2, 2, 2]
    assert RowPlan(D, B, M, 0, 8, 'row').counts == [18] * 8 and RowPlan(D, B, M, 0, 4, 'row').counts == [36] * 4
    with pytest.raises(ValueError):
        RowPlan(D, B, M, 0, 19, 'unit')


def test_separable_conv_folds_a_dilation_that_reaches_past_the_map():
    """ASPP rate 36 on a 32 x 32 map: only the centre tap of the depthwise kernel sees data, so SeparableConv2d folds it into
    the pointwise weights -- same output and gradients as the two convolutions, zero gradient on the eight unused taps."""
    from aadg_amd.models.deeplab import SeparableConv2d
    torch.manual_seed(0)
    m = SeparableConv2d(6, 4, dilation=36).double()
    x = torch.randn(2, 6, 32, 32, dtype=torch.float64, requires_grad=True)
    y = m(x)
    y.square().sum().backward()
    got = (y.detach(), x.grad.clone(), m[0].weight.grad.clone(), m[1].weight.grad.clone())
    x.grad = None
    m.zero_grad()
    y2 = m[1](m[0](x))                                         # the two convolutions, unfolded
    y2.square().sum().backward()
    want = (y2.detach(), x.grad, m[0].weight.grad, m[1].weight.grad)
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-10)
    g = got[2].clone()
    g[:, 0, 1, 1] = 0
    assert g.abs().max().item() == 0.0
    # a map larger than the dilation takes the ordinary path
    m2 = SeparableConv2d(6, 4, dilation=12).double()
    x2 = torch.randn(1, 6, 32, 32, dtype=torch.float64)
    assert torch.allclose(m2(x2), m2[1](m2[0](x2)))


def test_global_avg_pool_f32_matches_mean():
    from aadg_amd.models.deeplab import global_avg_pool_f32
    torch.manual_seed(1)
    x = torch.randn(3, 5, 4, 6, dtype=torch.float64, requires_grad=True)
    w = torch.randn(3, 5)
    y = global_avg_pool_f32(x)
    (y * w).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = x2.mean(dim=(2, 3), dtype=torch.float32)
    (y2 * w).sum().backward()
    assert y.dtype == torch.float32 and torch.equal(y, y2) and torch.allclose(x.grad, x2.grad, rtol=1e-6, atol=1e-8)


def test_fast_draw_equals_object_path():
    """transform.fast_train_units (the standard pipeline without per-image objects) makes the same random draws, in the same order,
    as synthetic.__getitem__ -> Policy.__call__ -> DGRandomScaleCrop -> ToTensor: identical unit records, domain codes and names over
    several batches (the CutMix queues fill up after 10 calls), and identical python / numpy generator states afterwards."""
    import random
    import numpy as np
    import torch
    from helpers import Cfg
    from aadg_amd.data import transform as T
    from aadg_amd.data.policy import DGMultiPolicy, parse_policies
    from aadg_amd.data.synthetic import SyntheticDGSegmentation
    for name, D in (('optic', 3), ('rvs', 3), ('optic', 8)):
        tr, _ = T.get_dg_segtransform(name, 48, D)
        ds = SyntheticDGSegmentation(D, 3, 48 if name == 'optic' else 96, name, 'train', tr, device='cpu')
        for seed in range(4):
            pol = np.random.RandomState(seed).randint(0, 10, (6, 20))
            runs = []
            for fast in (False, True):
                random.seed(seed)
                np.random.seed(seed)
                ds.transforms.transforms[0] = DGMultiPolicy(parse_policies(pol, Cfg(), None))
                res = []
                for _ in range(3):
                    if fast:
                        units, dc, dcs, names, M, kind = T.fast_train_units(ds, 3)
                    else:
                        flat, refs, M = T.collect_refs([ds[0] for _ in range(3)], True)
                        units = T.refs_to_units(refs)
                        dc = torch.cat([b['dc'] for b in flat], 0).numpy()
                        dcs = torch.stack([b['dc_single'] for b in flat], 0).numpy()
                        names = [b['img_name'] for b in flat]
                    res.append((units, dc, dcs, names))
                runs.append((res, random.random(), np.random.rand()))
            (a, ra, na), (b, rb, nb) = runs
            assert ra == rb and na == nb
            for (u1, d1, s1, n1), (u2, d2, s2, n2) in zip(a, b):
                assert u1.tobytes() == u2.tobytes() and np.array_equal(d1, d2) and np.array_equal(s1, s2) and n1 == n2
    # a non-standard pipeline is refused (the loader then takes the object path)
    ds.transforms.transforms[0] = T.Identity()
    assert T.fast_train_units(ds, 2) is None


def test_batch_drawn_ahead_equals_batch_drawn_in_place():
    """Round 4: predraw_train_batch draws the python-generator part of the NEXT batch (sub-policy choices, scale / crop geometry, soft
    codes) before the policies are known -- these draws do not depend on what the policies contain.  The batch completed later must
    be the batch drawn in one go: same records, codes, names and generator states, for freshly injected policies (the search loop:
    a new DGMultiPolicy per epoch) and for policies that stay (their CutMix queues carry over); a draw that does not fit the pipeline
    state it is consumed in is refused."""
    import random
    import numpy as np
    from helpers import Cfg
    from aadg_amd.data import transform as T
    from aadg_amd.data.policy import DGMultiPolicy, parse_policies
    from aadg_amd.data.synthetic import SyntheticDGSegmentation
    tr, _ = T.get_dg_segtransform('optic', 48, 3)
    ds = SyntheticDGSegmentation(3, 3, 48, 'optic', 'train', tr, device='cpu')
    for seed in range(3):
        runs = []
        for ahead in (False, True):
            random.seed(seed)
            np.random.seed(seed)
            res = []
            for step in range(4):
                pol = np.random.RandomState(100 * seed + step).randint(0, 10, (6, 20))      # policies with and without Cutout
                if ahead and step > 0:
                    pass                                            # drawn at the end of the previous step (below)
                if step % 2 == 0 or step == 1:
                    ds.transforms.transforms[0] = DGMultiPolicy(parse_policies(pol, Cfg(), None))      # fresh policies: empty queues
                res.append(T.fast_train_units(ds, 3))
                if ahead:
                    # the next step injects fresh policies unless it is step 3 (which keeps step 2's objects: queues carry over)
                    assert T.predraw_train_batch(ds, 3, fresh_policies=(step + 1) != 3)
            if ahead:
                ds._predrawn = None                                 # the last draw ahead is never consumed: drop it, compare states before it
            runs.append(res)
        a, b = runs
        for (u1, d1, s1, n1, M1, k1), (u2, d2, s2, n2, M2, k2) in zip(a, b):
            assert u1.tobytes() == u2.tobytes() and np.array_equal(d1, d2) and np.array_equal(s1, s2) and n1 == n2
    # consumed in another state than it was drawn for: the draw is taken back (generator restored) and redone in place -- the batch is
    # the one a run without any drawing ahead produces
    def batch(ahead, n_ahead, fresh):
        random.seed(5)
        np.random.seed(5)
        ds._predrawn = None
        ds.transforms.transforms[0] = DGMultiPolicy(parse_policies(np.random.RandomState(0).randint(0, 10, (6, 20)), Cfg(), None))
        first = T.fast_train_units(ds, 3)                             # queues now hold 9 entries
        if ahead:
            assert T.predraw_train_batch(ds, n_ahead, fresh_policies=fresh)
        second = T.fast_train_units(ds, 3)
        return first[0].tobytes() + second[0].tobytes(), random.random(), np.random.rand()
    want = batch(False, 0, False)
    assert batch(True, 3, True) == want                               # assumed fresh policies, but the old ones stayed
    assert batch(True, 2, False) == want                              # another batch size
    assert batch(True, 3, False) == want                              # and a draw that fits
    # a third party draws from python's generator between the draw ahead and a consumer in ANOTHER pipeline state: the draw
    # is dropped, but the generator is NOT rewound -- the third party's numbers are not handed out a second time
    random.seed(9)
    np.random.seed(9)
    ds._predrawn = None
    ds.transforms.transforms[0] = DGMultiPolicy(parse_policies(np.random.RandomState(0).randint(0, 10, (6, 20)), Cfg(), None))
    T.fast_train_units(ds, 3)
    assert T.predraw_train_batch(ds, 2, fresh_policies=False)         # does not fit a batch of 3
    theirs = [random.random() for _ in range(4)]                      # e.g. validate()'s pipeline
    state = random.getstate()
    T.fast_train_units(ds, 3)
    later = [random.random() for _ in range(50000)]
    assert not any(later[i:i + 4] == theirs for i in range(0, len(later) - 4))
    random.setstate(state)
    assert random.random() not in theirs
    # a draw ahead for a pipeline that became non-standard is cleared, not kept forever
    ds._predrawn = None
    ds.transforms.transforms[0] = DGMultiPolicy(parse_policies(np.random.RandomState(0).randint(0, 10, (6, 20)), Cfg(), None))
    assert T.predraw_train_batch(ds, 3)
    keep = ds.transforms.transforms[0]
    ds.transforms.transforms[0] = T.Identity()
    assert T.fast_train_units(ds, 3) is None and ds._predrawn is None
    ds.transforms.transforms[0] = keep
    assert T.predraw_train_batch(ds, 3)
    ds._predrawn = None


def test_launch_plan_classes_statistics_and_late_units():
    """aadg_amd._lib.launch_plan mirrors unit_flow() / stats_by_pushforward() of csrc/aug_u8.hip on the host: hand-built unit records,
    one per rule (class by scale and Sharpness count; which slots need a pixel pass; which units are 'late')."""
    from aadg_amd import _lib
    H = W = crop = 64
    def unit(ops, sw=64, sh=64):
        u = np.zeros(1, _lib.UNIT_DTYPE)
        u['rect'][:, :, 2:] = -1
        u['n_ops'] = len(ops)
        for k, (op, f) in enumerate(ops):
            u['op'][0, k] = op
            u['farg'][0, k] = np.float32(f)
        u['scaled_w'], u['scaled_h'] = sw, sh
        return u
    AC, INV, EQ, SOL, POS, CON, COL, BRI, SHA, CUT = range(10)
    units = np.concatenate([
        unit([]),                                   # 0 plain, no statistics
        unit([(AC, 0)]),                            # 1 slot-0 statistics only: raw histogram, not late
        unit([(INV, 0), (EQ, 0)]),                  # 2 push-forward at slot 1: raw histogram, not late
        unit([(COL, 1.3), (AC, 0)]),                # 3 Color breaks the byte-map chain: pixel pass at slot 1 -> late
        unit([(INV, 0), (CON, 1.2)]),               # 4 Contrast needs the mean of L of the image after Invert: pixel pass -> late
        unit([(SHA, 1.5), (BRI, 1.2)]),             # 5 Sharpness stencil: the 'sharp' tile class, no statistics
        unit([(SHA, 1.0)]),                         # 6 Sharpness with factor 1 is the identity: plain class
        unit([(BRI, 1.1)], sw=40, sh=70),           # 7 shrinks x by < 2: generic class
        unit([(AC, 0), (SHA, 1.2), (EQ, 0)], 20, 64),   # 8 shrinks by > 2: staged; staged units never push forward
        unit([(SHA, 1.2), (SHA, 1.3), (SHA, 1.4)]),     # 9 three stencils: staged
        unit([(SHA, 1.2)], sw=64, sh=40),               # 10 shrinks y by < 2 with a stencil: generic, listed behind the plain generic units
        unit([(BRI, 1.1)], sw=70, sh=40),               # 11 shrinks y only: generic, two passes
        unit([(SHA, 1.2)], sw=40, sh=70),               # 12 shrinks x only but chains a stencil: generic with a stencil (two passes)
    ])
    classes, stats_mask, order, counts, stat_lists, late, _, n_wonly, _late_counts = _lib.launch_plan(units, H, W, crop)
    assert classes == 1 | 2 | 4
    assert counts == (6, 1, 4, 2) and n_wonly == 1                   # ABI 9: unit 7 (shrinks the width only, no stencil) leads the generic run
    assert sorted(order[:6].tolist()) == [0, 1, 2, 3, 4, 6] and order[6] == 5 and order[7:11].tolist() == [7, 11, 10, 12] and sorted(order[11:].tolist()) == [8, 9]
    assert stat_lists[0].tolist() == [1, 2, 8]                 # raw histograms: slot-0 statistics and push-forward sources
    assert stat_lists[1].tolist() == [3, 4] and stat_lists[2].tolist() == [8] and stat_lists[3].size == 0
    assert stats_mask == 0b111
    assert late.tolist() == [3, 4, 8]
    assert _lib.launch_hints(units, H, W, crop)[3] == counts
    # sizes that are not multiples of 4: everything is staged
    assert _lib.launch_plan(units, 63, 63, 61)[0] == 2


def test_shard_rows_refuses_more_ranks_than_rows():
    """an empty [lo, hi) slice would give that rank a NaN mean loss, which DDP then all-reduces into every replica."""
    from aadg_amd.distributed import shard_rows
    assert shard_rows(6, 1, 3) == (2, 4)
    with pytest.raises(ValueError):
        shard_rows(2, 0, 3)


def test_fast_draw_path_is_for_the_synthetic_dataset_only():
    """fast_train_units reads SyntheticDGSegmentation internals; any other dataset takes the object path."""
    from aadg_amd.data import transform as T

    class Other(object):
        phase = 'train'
        transforms = None
    assert T.fast_train_units(Other(), 2) is None


def test_pillow_tap_count_bound_behind_axis_taps():
    """csrc/aug_u8.hip: axis_taps gives the down-scaling passes 3 taps while 3 out > 2 in, 4 while 2 out > in, else Pillow's 5 slots.
    Pillow's precompute_coeffs (Resample.c) uses xmax - xmin = int(c + s + 0.5) - int(c - s + 0.5) coefficients per output (clipped to the
    image), in double arithmetic; checked here for every (in, out) with in <= 300 and for the bench sizes."""
    import numpy as np
    ins = list(range(8, 301)) + [512, 768, 1000, 1024]
    for n_in in ins:
        for n_out in range((n_in + 1) // 2, n_in):
            scale = n_in / n_out
            support = 1.0 * max(scale, 1.0)
            c = (np.arange(n_out, dtype=np.float64) + 0.5) * scale
            xmin = np.maximum((c - support + 0.5).astype(np.int64), 0)
            xmax = np.minimum((c + support + 0.5).astype(np.int64), n_in)
            taps = int((xmax - xmin).max())
            bound = 3 if 3 * n_out > 2 * n_in else (4 if 2 * n_out > n_in else 5)
            assert taps <= bound, (n_in, n_out, taps, bound)


def test_c_planner_draws_pythons_stream_draw_for_draw():
    """Round 6: aadg_draw_python_stream (csrc/host_draw.hip) advances a copy of the interpreter's Mersenne-Twister state
    exactly as the Python statement (_draw_python_stream_py: the reference pipeline's draws from `random`, data/policy.py:17-23,
    data/transform.py:38-53,104-131,260-274) -- same sub-policy choices, geometry, soft codes, queue lengths and the same generator state
    afterwards, over scale ranges that up- and down-scale, padded crops, ragged sub-policy counts, queues at and below their cap; an empty
    crop range raises what python's randint raises."""
    import random
    import time
    from aadg_amd.data import transform as T

    class Crop(object):
        def __init__(self, size, padding):
            self.size, self.padding = size, padding

    class SC(object):
        def __init__(self, lo, hi, size, padding):
            self.scale_range, self.crop = (lo, hi), Crop((size, size), padding)
    rs = np.random.RandomState(3)
    t_c = t_py = 0.0
    for case in range(40):
        n_items, D, M = int(rs.randint(1, 9)), int(rs.randint(1, 9)), int(rs.randint(1, 7))
        nsub = tuple(int(v) for v in rs.randint(1, 7, M))
        qlens = tuple(int(v) for v in rs.randint(0, 12, M))
        lo, hi = [(1.0, 1.5), (0.5, 2.0), (0.75, 1.25), (1.0, 1.0)][case % 4]
        W0 = int(rs.choice([64, 100, 256, 512]))
        size = int(rs.choice([W0, W0 // 2, W0 + 24]))
        sc = SC(lo, hi, size, int(rs.choice([0, 0, 4])))
        n_code = D + int(rs.randint(0, 3))
        random.seed(1000 + case)
        for _ in range(case):                                   # any position inside the 624-word block, and across a refill
            random.random()
        st = random.getstate()
        t0 = time.perf_counter()
        a = T._draw_python_stream_py(n_items, D, M, nsub, qlens, sc, n_code, W0, W0)
        t_py += time.perf_counter() - t0
        end_py = random.getstate()
        random.setstate(st)
        t0 = time.perf_counter()
        b = T._draw_python_stream(n_items, D, M, nsub, qlens, sc, n_code, W0, W0)
        t_c += time.perf_counter() - t0
        assert random.getstate() == end_py, case
        assert isinstance(b['R'], np.ndarray), "the library's planner did not run"
        assert list(a['R']) == b['R'].tolist() and np.array_equal(a['geo'], b['geo']) and a['queue_after'] == b['queue_after'], case
        assert np.array_equal(np.array(a['dcs'], dtype=np.float64), b['dcs']), case          # bit-equal float64 codes
    assert t_c < t_py, (t_c, t_py)
    # an empty crop range: the crop is larger than the scaled image can ever be padded to -> python's randint raises; so does the fast path
    sc = SC(1.0, 1.0, 64, 0)
    sc.crop.size = (64, 200)                                    # (h, w): no padding branch (w >= crop_h, h >= crop_w fails -> pad), then m2 <= 0
    for fn in (T._draw_python_stream_py, T._draw_python_stream):
        random.seed(5)
        try:
            fn(1, 1, 1, (1,), (0,), sc, 1, 64, 300)
            raised = False
        except ValueError:
            raised = True
        assert raised, fn
