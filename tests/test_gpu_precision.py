"""GPU: the precision contract of the headline number.  Round 5: bench.py's headline step runs the
backbone in float32 tensors on the f32x3 kernels and ASSERTS north_star's 1e-4 (first test below); the bfloat16-autocast step is a
labelled secondary figure whose looser bounds the remaining tests keep.  History: until round 4 the headline ran the convolutions
under bfloat16 autocast; BASELINE configs[1] is float32.  On IDENTICAL weights and inputs (forward only, dropout off) the quantities
the policy search consumes must agree between the two: raw Sinkhorn rewards per policy, per-policy BCE, Dice.  The bounds asserted
here (at a reduced 256 x 256 config, weights a few search steps old: median reward within 10 %, per-policy BCE within 1.5 %, Dice within 0.02)
are the ones bench.py states in its `dtype` field; the values measured at the headline config travel in its `precision` block
(0.1 % / 0.04 % / 0.0007 in round 3).  Round 4: three seeded batches, and `within_north_star_1e-4` flags -- the bf16 backbone moves the raw
rewards by 0.75e-4 .. 3.2e-4 and Dice by 0.9e-4 .. 6e-4 (absolute, headline config, runs of round 4): around, not reliably inside, the 1e-4
north_star states for Dice / Sinkhorn, which is why the flags are REPORTED per run and the assertion stays at the bounds below."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("size,batch,steps", [(256, 4, 5), (512, 8, 3)])
def test_f32x3_headline_backbone_is_inside_north_star_tolerance(hip, size, batch, steps):
    """Round 5: what bench.py's headline CLAIMS is asserted, not only reported.  The headline backbone runs float32
    tensors through the own convolution kernels' f32x3 instantiations (three bfloat16 matrix-core products per pair of (hi, lo)-split
    operands, float32 accumulation).  Against the library's float32 convolutions on identical weights (a few search steps old) and three
    seeded batches -- at the headline config itself, 512 x 512, B = 8 (144 images), and at 256 x 256 -- raw Sinkhorn rewards, per-policy
    BCE and Dice must ALL stay inside north_star's 1e-4 (measured on the first run of round 5: 3.6e-7 / 9e-8 / 1.1e-6)."""
    sys.path.insert(0, ROOT)
    import torch
    import bench
    a = bench.Args()
    a.cfg, a.backbone, a.batch, a.size = os.path.join("experiments", "optic_sinkhorn", "diversity.yaml"), "resnet50", batch, size
    a.backbone_dtype, a.no_sync_bn, a.placement, a.no_dropout = "f32x3", True, "row", True
    cfg, st = bench.build_state(a, 0, 1)
    for i in range(steps):
        st.search_step(i, max_iters=1)
    torch.cuda.synchronize()
    _, st32 = bench.build_state(a, 0, 1, backbone_dtype="fp32")
    p = bench.precision_check(st, st32, st.M, len(cfg.DATASET.DG.TRAIN), batch, label="f32x3")
    print("f32x3 precision at %d^2, B = %d:" % (size, batch), {k: v for k, v in p.items() if k not in ("per_batch", "what")})
    assert p["batches"] == 3 and all(p["within_north_star_1e-4"].values()), p
    assert p["reward_abs_max_diff"] <= 1e-4 and p["dice_abs_max_diff"] <= 1e-4 and p["bce_abs_max_diff"] <= 1e-4, p
    assert p["reward_ranking_equal"]


@pytest.mark.parametrize("backbone,size,batch", [("resnet50", 256, 4), ("mobilenet_v2", 256, 4)])
def test_bf16_backbone_keeps_the_search_quantities(hip, backbone, size, batch):
    sys.path.insert(0, ROOT)
    import bench
    a = bench.Args()
    a.cfg, a.backbone, a.batch, a.size = os.path.join("experiments", "optic_sinkhorn", "diversity.yaml"), backbone, batch, size
    a.backbone_dtype, a.no_sync_bn, a.placement, a.no_dropout = "bf16", True, "row", True
    cfg, st = bench.build_state(a, 0, 1)
    for i in range(5):                                  # as in bench.py: the comparison runs on weights a few search steps old
        st.search_step(i, max_iters=1)
    _, st32 = bench.build_state(a, 0, 1, backbone_dtype="fp32")
    p = bench.precision_check(st, st32, st.M, len(cfg.DATASET.DG.TRAIN), batch)
    # a reduced config (256 x 256, clouds of 4 points per domain): measured 0.3 - 2.7 % on the rewards; the headline config (512 x 512, clouds of
    # 8 points) measures 0.1 % (bench.py `precision`)
    # three seeded batches since round 4: the bound holds for the median batch, the worst of three may reach twice as far (tiny clouds).
    # The five warm-up steps are not bit-reproducible (split-K float atomics in the weight gradients), so the compared weights differ from
    # run to run: 30 runs of round 5 gave medians of 0.3 - 5.2 % (one run in ten above 5 %) -- the bounds are twice that.
    assert float(np.median([b["reward_rel"] for b in p["per_batch"]])) <= 1e-1 and p["reward_rel_max_diff"] <= 2e-1, p
    # (scripts/ab/precision_spread.py, 30 runs of the mobilenet case: per-policy BCE 0.28 - 0.76 %, Dice 0.001 - 0.0103 -- same reason)
    assert p["bce_rel_max_diff"] <= 1.5e-2, p
    assert p["dice_abs_max_diff"] <= 2e-2, p
    assert p["batches"] == 3 and len(p["per_batch"]) == 3
    for b in p["per_batch"]:
        assert np.isfinite(b["rewards_bf16"]).all() and len(b["rewards_bf16"]) == st.M
    assert set(p["within_north_star_1e-4"]) == {"rewards", "dice", "bce"}


def test_bf16_segformer_keeps_the_search_quantities(hip):
    """BASELINE configs[4] (SegFormer-B2, 8 source domains) keeps its residual stream in bfloat16 under autocast since round 3
    (csrc/layernorm.hip stores s = x + branch in the tokens' dtype; torch's autocast would add and normalise in float32).  End-to-end
    bound of that choice on the quantities the search consumes -- bfloat16 autocast against float32 on identical weights and batches,
    forward only, after a few training steps through the fused residual / DropPath path -- and a train-mode gradient comparison."""
    sys.path.insert(0, ROOT)
    import torch
    import bench
    a = bench.Args()
    a.cfg, a.backbone, a.batch, a.size = os.path.join("experiments", "merged_sinkhorn", "segformer_b2_d8.yaml"), "mit_b2", 2, 128
    a.backbone_dtype, a.no_sync_bn, a.placement, a.no_dropout = "bf16", True, "row", True
    cfg, st = bench.build_state(a, 0, 1)
    for i in range(3):
        st.search_step(i, max_iters=1)
    _, st32 = bench.build_state(a, 0, 1, backbone_dtype="fp32")
    p = bench.precision_check(st, st32, st.M, len(cfg.DATASET.DG.TRAIN), a.batch)
    print("segformer precision", {k: v for k, v in p.items() if k not in ("per_batch", "what")})
    # measured (round 4, random-init MiT-B2, clouds of 2 points per domain, 28 domain pairs per policy): rewards within 5-18 %
    # (0.7-1.0 absolute on sums of ~14; the ranking of the 6 policies changes), per-policy BCE within 1-3 %, Dice within 0.006-0.026 --
    # an order of magnitude looser than the DeepLab backbones above (0.3 % / 0.04 %): 16 blocks of bfloat16 residual adds in front of a
    # pooled 512-vector whose cosine distances between single points are the reward.  BASELINE names configs[4] as a bf16 config; the
    # bound is recorded here so that it cannot drift unnoticed
    assert float(np.median([b["reward_rel"] for b in p["per_batch"]])) <= 0.15 and p["reward_rel_max_diff"] <= 0.30, p
    assert p["bce_rel_max_diff"] <= 5e-2 and p["dice_abs_max_diff"] <= 4e-2, p
    # train mode (DropPath off through --no_dropout's twin below, same weights): parameter gradients of one batch, bf16 against float32
    from aadg_amd.search_dg import _autocast, _bare
    from aadg_amd import _lib
    sample = next(iter(st.train_loader))
    grads = []
    for s_ in (st, st32):
        m = _bare(s_.model)
        m.train()
        for mod in m.modules():
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        m.zero_grad(set_to_none=True)
        with _autocast(s_.args):
            seg, _ = s_.model(sample["aug_images"])
        _lib.policy_bce_backward(seg, sample["aug_labels"], st.M)
        grads.append({n: q.grad.float().clone() for n, q in m.named_parameters() if q.grad is not None})
    num = sum(float((grads[0][n] - grads[1][n]).double().pow(2).sum()) for n in grads[1])
    den = sum(float(grads[1][n].double().pow(2).sum()) for n in grads[1])
    rel = (num / den) ** 0.5
    print("segformer bf16 vs fp32 gradient, relative L2 over all parameters: %.4f" % rel)
    assert set(grads[0]) == set(grads[1]) and rel <= 0.15, rel
