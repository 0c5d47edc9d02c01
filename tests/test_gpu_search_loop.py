"""GPU: the search driver end to end through the reference CLI (run.py --cfg ...), short synthetic run:
warm-up epoch -> EMA sync -> 2 policy-search epochs (controller graphs, fused augmentation, BCE/Dice and
Sinkhorn kernels, PPO) -> validation -> the reference's on-disk artefacts (search_dg.py:388-407)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_py_search_smoke(hip, tmp_path):
    import run
    from aadg_amd.config.defaults import _C
    args = ["--cfg", os.path.join(ROOT, "experiments", "optic_sinkhorn", "smoke.yaml"), "--output_dir", str(tmp_path / "out"),
            "--crop_size", "64", "--epoch_items", "4", "--backbone_dtype", "bf16"]
    _C.defrost()
    _C.LOG_DIR = str(tmp_path / "log")
    best = run.main(args)
    assert set(best) >= {"epoch", "avg_dsc", "cup_dsc", "disc_dsc"}
    outs = glob.glob(str(tmp_path / "out" / "optic" / "smoke_*"))
    assert len(outs) == 1
    d = outs[0]
    for f in ("final_model_state.pth", "final_controller_state.pth", "mag_probs_trajectory.npy",
              "op_probs_trajectory.npy", "final_result.json", "train.log"):
        assert os.path.exists(os.path.join(d, f)), f
    assert np.load(os.path.join(d, "op_probs_trajectory.npy")).shape == (2, 10)
    assert np.load(os.path.join(d, "mag_probs_trajectory.npy")).shape == (2, 10)
    sd = torch.load(os.path.join(d, "final_controller_state.pth"), map_location="cpu")
    assert set(k.split(".")[0] for k in sd) == {"embedding", "lstm", "outop", "outmag"}
    json.load(open(os.path.join(d, "final_result.json")))
    log = open(os.path.join(d, "train.log")).read()
    assert "OT" in log and "controller loss" in log
    assert all(np.isfinite(float(x)) for x in [best["avg_dsc"]])


def test_run_py_search_smoke_rvs_reinforce(hip, tmp_path):
    """The single-class RVS driver (the reference's search_dg_2d.py) with the REINFORCE criterion, scale range
    [0.5, 2] (generic fused tiles) and fp32 backbone."""
    import run
    from aadg_amd.config.defaults import _C
    args = ["--cfg", os.path.join(ROOT, "experiments", "rvs_sinkhorn", "smoke.yaml"), "--output_dir", str(tmp_path / "out"),
            "--crop_size", "64", "--epoch_items", "4"]
    _C.defrost()
    _C.LOG_DIR = str(tmp_path / "log")
    best = run.main(args)
    outs = glob.glob(str(tmp_path / "out" / "rvs" / "smoke_*"))
    assert len(outs) == 1 and os.path.exists(os.path.join(outs[0], "final_result.json"))
    assert np.isfinite(best["avg_dsc"])


def test_run_py_config0_unet_fixed_policy(hip, tmp_path):
    """BASELINE configs[0] through the reference CLI: MODEL.NAME unet (models/unet.py:58-72 contract: (logits, pooled
    bottleneck)), ONE vessel source domain, fixed policy [Contrast .5, Sharpness .5] instead of the controller search
    (run.py --fixed_policy), 256x256 crops, batch 2."""
    import run
    from aadg_amd.config.defaults import _C
    from aadg_amd.models import build_model
    from aadg_amd.models.deeplab import UNetSmall
    args = ["--cfg", os.path.join(ROOT, "experiments", "rvs_sinkhorn", "unet_fixed.yaml"), "--output_dir", str(tmp_path / "out"),
            "--crop_size", "256", "--epoch_items", "4", "--fixed_policy"]
    _C.defrost()
    _C.LOG_DIR = str(tmp_path / "log")
    best = run.main(args)
    assert isinstance(build_model(_C), UNetSmall)
    outs = glob.glob(str(tmp_path / "out" / "rvs" / "unet_fixed_*"))
    assert len(outs) == 1
    for f in ("final_model_state.pth", "final_result.json", "train.log"):
        assert os.path.exists(os.path.join(outs[0], f)), f
    sd = torch.load(os.path.join(outs[0], "final_model_state.pth"), map_location="cpu")
    assert any(k.startswith("d1.") for k in sd) and any(k.startswith("mid.") for k in sd)
    assert np.load(os.path.join(outs[0], "op_probs_trajectory.npy")).size == 0          # no controller statistics: nothing was searched
    assert np.isfinite(best["avg_dsc"])
    log = open(os.path.join(outs[0], "train.log")).read()
    assert "Seg Loss" in log and "OT 0.00000" in log                                    # one source domain: no domain pair, zero reward


def test_baseline_config0_fixed_policy_vessel(hip, oracle):
    """BASELINE configs[0]: single-domain vessel data, FIXED policy [Contrast .5, Sharpness .5] (no controller
    search), 256x256, batch 2 -- the reference's CPU-runnable plumbing case, HIP vs oracle, bit-exact."""
    import random
    from aadg_amd.data import transform as T
    from aadg_amd.data.basic import DevicePool
    from aadg_amd.data.policy import DGMultiPolicy
    from aadg_amd.data.synthetic import make_pool
    imgs, msks = make_pool(7, 1, 4, 256, 256, 'rvs')
    pool = DevicePool(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda())
    fixed = [[[('Contrast', 0.5), ('Sharpness', 0.5)]]]                       # M = 1 policy, Q = 1 sub-policy, L = 2
    tf = T.Compose([DGMultiPolicy(fixed), T.DGRandomScaleCrop(256, scale_range=[0.5, 2]), T.Normalize_dg('vessel'), T.ToTensor('vessel')])
    random.seed(3)
    np.random.seed(3)
    batch = [[tf({'image': pool.image(i), 'label': pool.mask(i), 'img_name': 'v%d' % i, 'dc': 0})] for i in range(2)]
    flat, refs, M = T.collect_refs(batch, nested=True)
    units = T.refs_to_units(refs)
    got_img, got_lbl = T.materialize(refs)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, 256, 1)
    assert got_lbl.shape[1] == 1
    assert np.array_equal(got_img.cpu().numpy(), want_img) and np.array_equal(got_lbl.cpu().numpy(), want_lbl)


def test_early_controller_update_is_the_same_search(hip, tmp_path, monkeypatch):
    """Issuing the PPO update and the next epoch's sampling before the last backward pass (search_step) changes the schedule,
    not the search.  With the training epoch replaced by a deterministic reward source (the real one carries run-to-run
    last-bit noise from split-K atomics, which Adam amplifies) both orders give bit-identical controllers and trajectories."""
    import random
    import run
    from aadg_amd import search_dg
    from aadg_amd.config.defaults import _C

    def fake_train(config, loader, model, disc, mc, dc, mo, do, M, epoch, wd, logger, args=None, max_iters=None, on_last_rewards=None):
        g = torch.Generator(device="cpu").manual_seed(100 + epoch)
        nr = hip.normalize_rewards(torch.randn(M, generator=g).cuda())
        if on_last_rewards is not None:
            on_last_rewards(nr)
        return nr
    monkeypatch.setattr(search_dg, "train", fake_train)
    results = []
    for early in (True, False):
        orig = search_dg.SearchState.__init__

        def init(self, *a, _orig=orig, _early=early, **k):
            _orig(self, *a, **k)
            self.args.early_controller_update = _early
        monkeypatch.setattr(search_dg.SearchState, "__init__", init)
        out = tmp_path / ("out_%d" % early)
        args = ["--cfg", os.path.join(ROOT, "experiments", "optic_sinkhorn", "smoke.yaml"), "--output_dir", str(out),
                "--crop_size", "64", "--epoch_items", "4", "--backbone_dtype", "fp32"]
        _C.defrost()
        _C.LOG_DIR = str(tmp_path / ("log_%d" % early))
        for seed_fn in (random.seed, np.random.seed, torch.manual_seed):      # the driver itself seeds nothing (as the reference)
            seed_fn(77)
        run.main(args)
        monkeypatch.setattr(search_dg.SearchState, "__init__", orig)
        d = glob.glob(str(out / "optic" / "smoke_*"))[0]
        results.append((torch.load(os.path.join(d, "final_controller_state.pth"), map_location="cpu"),
                        np.load(os.path.join(d, "op_probs_trajectory.npy")), np.load(os.path.join(d, "mag_probs_trajectory.npy"))))
    a, b = results
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), k


def test_run_py_reference_yaml_without_flags_takes_the_own_kernels(hip, tmp_path, capfd):
    """`python run.py --cfg experiments/optic_sinkhorn/diversity.yaml` -- the reference's yaml, NO arithmetic flag --
    runs the measured path: --backbone_dtype defaults to f32x3, the convolutions of the model the yaml names (DeepLabV3+/MobileNetV2)
    take the own float32-precision kernels, the weight-gradient stream is on, and layers outside the kernels' tiles are listed with the
    reason (library float32).  (--max_epochs / --epoch_items / --crop_size only shorten the run.)"""
    import run
    from aadg_amd import _lib
    from aadg_amd.config.defaults import _C
    from aadg_amd.models import deeplab
    assert run.parse_args(["--cfg", "x"]).backbone_dtype == "f32x3"
    args = ["--cfg", os.path.join(ROOT, "experiments", "optic_sinkhorn", "diversity.yaml"), "--output_dir", str(tmp_path / "out"),
            "--max_epochs", "1", "--epoch_items", "8", "--crop_size", "128"]     # one warm-up batch of TRAIN.BATCH_SIZE = 8 items
    _C.defrost()
    _C.LOG_DIR = str(tmp_path / "log")
    best = run.main(args)
    assert np.isfinite(best["avg_dsc"])
    own, lib = deeplab.f32x3_coverage()
    assert len(own) >= 35, (len(own), lib)                    # MobileNetV2's pointwise convolutions, ASPP, decoder
    assert all(v.startswith("library: ") for v in lib.values())
    assert _lib.wgrad_stream_enabled()
    err = capfd.readouterr().err
    assert "f32x3: %d convolution layers on the own float32-precision matrix-core kernels" % len(own) in err
