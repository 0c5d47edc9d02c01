"""Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE in the build container.

Run (build container only; /root/reference does not exist on the GPU box):

    python tests/golden/make_golden.py

What travels is data only: inputs (seeded synthetic uint8 images, policy matrices, RNG seeds) and
the outputs the reference's own functions produced for them.  Third-party modules the reference
imports but this image lacks are replaced by empty stand-ins *that the exercised code paths never
call* (cv2: only GammaCorrection; torchvision: only `transforms.Compose`, re-implemented in 6 lines;
kornia: only the geometric/hue float ops, which are NOT fixture material; segmentation_models_pytorch:
only the model factory).  The pixel arithmetic itself is executed by the reference's code + Pillow.
"""
import json
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    cv2 = types.ModuleType("cv2")
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvtt = types.ModuleType("torchvision.transforms.transforms")

    class Compose(object):
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tvtt.Compose = Compose
    tvt.transforms = tvtt
    tvt.Compose = Compose
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.transforms"] = tvtt
    sys.modules["kornia"] = types.ModuleType("kornia")
    sys.modules["segmentation_models_pytorch"] = types.ModuleType("segmentation_models_pytorch")
    sys.path.insert(0, REF)


def synth_image(rs, H, W, kind):
    """Seeded synthetic RGB uint8 image (smooth field + colour cast + noise) and a 0/128/255 mask."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    field = 90 + 60 * np.sin(xx / (3.0 + kind)) * np.cos(yy / (4.0 + kind))
    cast = np.array([1.0, 0.8 - 0.1 * kind, 0.5 + 0.15 * kind])
    img = field[..., None] * cast + rs.randint(0, 40, (H, W, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    cy, cx = H * (0.4 + 0.1 * rs.rand()), W * (0.4 + 0.1 * rs.rand())
    rr = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
    mask = np.full((H, W), 255, np.uint8)
    mask[rr < 0.35 * H] = 128
    mask[rr < 0.18 * H] = 0
    return img, mask


class Cfg(object):
    """Attribute bag with the CONTROLLER keys parse_policies reads."""

    class _C(object):
        pass

    def __init__(self, L=2, NUM_MAGS=10, EXCLUDE_OPS=(), EXCLUDE_OPS_NUM=0, SEED=1023):
        self.CONTROLLER = Cfg._C()
        self.CONTROLLER.L = L
        self.CONTROLLER.NUM_MAGS = NUM_MAGS
        self.CONTROLLER.EXCLUDE_OPS = list(EXCLUDE_OPS)
        self.CONTROLLER.EXCLUDE_OPS_NUM = EXCLUDE_OPS_NUM
        self.SEED = SEED


class NullLogger(object):
    def info(self, *a):
        pass


def gen_ops():
    """Each of the 10 registry ops x 10 magnitude levels, via the reference's apply_augment."""
    from PIL import Image
    import PIL
    from data.basic import apply_augment, augment_list
    rs = np.random.RandomState(1023)
    out = {"pillow_version": np.array(PIL.__version__)}
    names = [fn.__name__ for fn, _, _ in augment_list()]
    out["op_names"] = np.array(names)
    for tag, (H, W) in (("a", (32, 32)), ("b", (24, 40))):
        img, mask = synth_image(rs, H, W, 0 if tag == "a" else 2)
        out["img_" + tag], out["mask_" + tag] = img, mask
        pim, pmask = Image.fromarray(img), Image.fromarray(mask)
        res = np.zeros((10, 10, H, W, 3), np.uint8)
        mres = np.zeros((10, 10, H, W), np.uint8)
        for oi, name in enumerate(names):
            for li in range(10):
                np.random.seed(1000 * oi + li)  # Cutout draws np.random.uniform twice
                o, m = apply_augment(pim, pmask, name, li / 9)
                res[oi, li] = np.asarray(o)
                mres[oi, li] = np.asarray(m)
        out["out_" + tag], out["mout_" + tag] = res, mres
    np.savez_compressed(os.path.join(OUT, "u8_ops.npz"), **out)


def gen_parse():
    from data.policy import parse_policies
    rs = np.random.RandomState(7)
    cases = []
    for L, excl in ((2, []), (2, ["Cutout"]), (3, ["Invert", "Equalize"]), (1, [])):
        n_ops = 10 - len(excl)
        pol = np.zeros((6, 5 * L * 2), np.int64)
        pol[:, 0::2] = rs.randint(0, n_ops, (6, 5 * L))
        pol[:, 1::2] = rs.randint(0, 10, (6, 5 * L))
        parsed = parse_policies(pol, Cfg(L=L, EXCLUDE_OPS=excl), NullLogger())
        cases.append({"L": L, "exclude": excl, "policies": pol.tolist(),
                      "parsed": [[[[n, float(v)] for n, v in sp] for sp in p] for p in parsed]})
    with open(os.path.join(OUT, "parse_policies.json"), "w") as f:
        json.dump(cases, f)


def gen_pipeline():
    """Whole live pipeline: DGMultiPolicy -> DGRandomScaleCrop -> Normalize_dg -> ToTensor -> collate."""
    from PIL import Image
    from data.policy import DGMultiPolicy, parse_policies
    from data import transform as T
    # the 256 float32 values Normalize_dg can produce, from the reference's own arithmetic
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    ramp3 = np.stack([ramp] * 3, -1)
    lut, _ = T.Normalize_dg('vessel').normalize(Image.fromarray(ramp3), Image.fromarray(ramp))
    lut = np.ascontiguousarray(lut[..., 0].reshape(256))
    out = {"lut256": lut}
    cfgs = [
        # name, dataset, src HxW, crop, scale range, D, items, seed
        ("optic_up", "optic", (40, 40), 32, [1, 1.5], 3, 2, 11),
        ("optic_pad", "optic", (28, 28), 32, [1, 1.5], 3, 1, 12),
        ("rvs_down", "vessel", (48, 48), 32, [0.5, 2], 3, 1, 13),
        ("optic_rect", "optic", (36, 44), 32, [1, 1.5], 2, 1, 14),
    ]
    meta = []
    for name, ds, (H, W), crop, sr, D, items, seed in cfgs:
        rs = np.random.RandomState(seed)
        pool_img, pool_msk = [], []
        for d in range(D):
            for _ in range(2):
                im, mk = synth_image(rs, H, W, d)
                if ds == "vessel":
                    mk = ((mk == 128) * 255).astype(np.uint8)
                pool_img.append(im)
                pool_msk.append(mk)
        pol = np.zeros((6, 20), np.int64)
        pol[:, 0::2] = rs.randint(0, 10, (6, 10))
        pol[:, 1::2] = rs.randint(0, 10, (6, 10))
        parsed = parse_policies(pol, Cfg(), NullLogger())
        tf = T.transforms.Compose([DGMultiPolicy(parsed), T.DGRandomScaleCrop(crop, scale_range=sr),
                                   T.Normalize_dg(ds), T.ToTensor(ds)])
        random.seed(seed)
        np.random.seed(seed)
        batch, picks = [], []
        for it in range(items):
            per_item = []
            for d in range(D):
                idx = int(np.random.choice(2, 1)[0])  # FundusSegmentation.__getitem__ draw (data/optic.py:84)
                picks.append(2 * d + idx)
                s = {'image': Image.fromarray(pool_img[2 * d + idx]), 'label': Image.fromarray(pool_msk[2 * d + idx]),
                     'img_name': 'im%d_%d' % (d, idx), 'dc': d}
                per_item.append(tf(s))
            batch.append(per_item)
        nb = T.train_dg_collate_fn(batch)

        def codes(x):  # float32 image tensor -> index into lut256 (exact)
            x = x.numpy()
            c = np.rint((x + 1.0) * 127.5).astype(np.int64)
            assert np.array_equal(lut[c], x)
            return c.astype(np.uint8)

        out[name + "_pool_img"] = np.stack(pool_img)
        out[name + "_pool_msk"] = np.stack(pool_msk)
        out[name + "_policies"] = pol
        out[name + "_picks"] = np.array(picks)
        out[name + "_aug_images"] = codes(nb['aug_images'])
        out[name + "_aug_labels"] = nb['aug_labels'].numpy().astype(np.uint8)
        out[name + "_image"] = codes(nb['image'])
        out[name + "_label"] = nb['label'].numpy().astype(np.uint8)
        out[name + "_dc"] = nb['dc'].numpy()
        assert np.array_equal(out[name + "_aug_labels"].astype(np.float32), nb['aug_labels'].numpy())
        meta.append({"name": name, "dataset": ds, "H": H, "W": W, "crop": crop, "scale_range": sr, "D": D,
                     "items": items, "seed": seed})
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "pipeline.npz"), **out)


def gen_controller():
    """Controller sample/evaluate, PPO and REINFORCE updates, momentum discriminator (CPU; the
    reference's hard-wired .cuda() calls are neutralised)."""
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.controller import Controller
    from models.discriminator import MomentumFeatureDiscriminator
    import losses as RL
    out = {}
    for tag, excl in (("full", []), ("excl", ["Cutout"])):
        cfg = Cfg(EXCLUDE_OPS=excl)
        cfg.CONTROLLER.T, cfg.CONTROLLER.C, cfg.CONTROLLER.PENALTY, cfg.CONTROLLER.LOSS = 2, 2.5, 1e-5, 'ppo'
        torch.manual_seed(1023)
        c = Controller(cfg)
        for k, v in c.state_dict().items():
            out["%s_sd_%s" % (tag, k)] = v.numpy().copy()
        torch.manual_seed(7)
        policies, op_probs, mag_probs, log_probs, entropies = c(6)
        out[tag + "_policies"] = policies.numpy()
        out[tag + "_op_probs"] = op_probs.detach().numpy()
        out[tag + "_mag_probs"] = mag_probs.detach().numpy()
        out[tag + "_log_probs"] = log_probs.detach().numpy()
        out[tag + "_entropies"] = entropies.detach().numpy()
        out[tag + "_evaluate"] = c.evaluate(policies, 6).detach().numpy()
        reward = torch.tensor([0.3, -1.2, 0.8, 1.1, -0.4, -0.6])
        out[tag + "_reward"] = reward.numpy()
        opt = torch.optim.Adam(c.parameters(), lr=0.00035)
        crit = RL.ProximalPolicyOptimization(cfg)
        crit.register_optimizer(opt)
        loss, score, ent = crit(c, policies, log_probs, entropies, reward)
        out[tag + "_ppo"] = np.array([loss.item(), score.item(), ent.item()], np.float64)
        out[tag + "_ppo_evaluate_after"] = c.evaluate(policies, 6).detach().numpy()
        # REINFORCE on a fresh controller
        torch.manual_seed(1023)
        c2 = Controller(cfg)
        torch.manual_seed(7)
        policies, _, _, log_probs, entropies = c2(6)
        opt2 = torch.optim.Adam(c2.parameters(), lr=0.00035)
        crit2 = RL.Reinforce(cfg)
        crit2.register_optimizer(opt2)
        loss, score, ent = crit2(c2, policies, log_probs, entropies, reward)
        out[tag + "_reinforce"] = np.array([loss.item(), score.item(), ent.item()], np.float64)
        out[tag + "_reinforce_evaluate_after"] = c2.evaluate(policies, 6).detach().numpy()
    torch.manual_seed(3)
    d = MomentumFeatureDiscriminator(3, 64)
    x = torch.randn(12, 64)
    out["disc_x"] = x.numpy()
    for k, v in d.state_dict().items():
        out["disc_sd_" + k] = v.numpy().copy()
    logits, fe = d(x, momentum=True, return_feature=True)
    out["disc_mom_logits"], out["disc_mom_fe"] = logits.numpy(), fe.numpy()
    out["disc_logits"] = d(x, momentum=False).detach().numpy()
    d.momentum_update()
    logits, fe = d(x, momentum=True, return_feature=True)
    out["disc_mom_fe_after_update"] = fe.numpy()
    t = torch.softmax(torch.randn(12, 3), dim=1)
    out["ce_target"] = t.numpy()
    out["ce_value"] = np.array(RL.CrossEntropy()(d(x), t).item())
    np.savez_compressed(os.path.join(OUT, "controller.npz"), **out)


def gen_functional():
    """Non-kornia float ops of data/functional.py (+ data/kernels.py) on torch-CPU."""
    import torch
    from data import functional as RF
    from data import kernels as RK
    out = {}
    torch.manual_seed(1023)
    img = torch.rand(2, 3, 20, 24)
    img[0, :, :4, :4] = 0.0
    img[1, 1] *= 0.5                      # a channel with a reduced range
    out["img"] = img.numpy().copy()
    permag = torch.tensor([0.25, 0.8])
    for sg in (0.5, 1.0, 2.0):
        out["gauss3_%g" % sg] = RK.get_gaussian_3x3kernel(torch.tensor([sg])).numpy()
    out["sharp_kernel"] = RK.get_sharpness_kernel().numpy()
    cases = []
    for name in ("invert", "gray", "auto_contrast", "equalize", "hflip", "vflip"):
        cases.append((name, None))
    for name in ("solarize", "contrast", "saturate", "brightness", "sharpness"):
        for m in (0.0, 0.3, 1.0):
            cases.append((name, torch.tensor([m])))
        cases.append((name, permag))
    cases.append(("posterize", torch.tensor([0.5])))
    cases.append(("gaussian_blur3x3", torch.tensor([0.7])))
    cases.append(("gaussian_blur3x3", torch.tensor([1.5])))
    keys = []
    for i, (name, mag) in enumerate(cases):
        fn = getattr(RF, name)
        res = fn(img.clone()) if mag is None else fn(img.clone(), mag.clone())
        key = "f%02d_%s" % (i, name)
        keys.append(key)
        out[key] = res.detach().numpy().copy()
        out[key + "_mag"] = np.zeros(0, np.float32) if mag is None else mag.numpy()
    for i, m in enumerate((torch.tensor([0.4]), permag)):
        torch.manual_seed(55 + i)
        res = RF.sample_pairing(img.clone(), m.clone())
        torch.manual_seed(55 + i)
        out["sp%d_perm" % i] = torch.randperm(2).numpy()
        out["sp%d" % i] = res.numpy().copy()
        out["sp%d_mag" % i] = m.numpy()
    out["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(OUT, "functional.npz"), **out)


def gen_operations():
    """`_Operation.forward` (data/operations.py:73-100) of every module whose arithmetic lives in the reference itself (the kornia
    geometric ops and Hue are not fixture material), in training mode (RelaxedBernoulli mask blend) and eval mode (Bernoulli
    mask, in-place on the selected samples).  The random draws are FIXED and recorded -- the mask returned by get_mask(), the
    0/1 draw behind the magnitude sign flip (torch.randint) and SamplePairing's permutation (torch.randperm := next sample,
    cyclic) -- because the
    CPU and GPU generators produce different streams; the test injects the same values."""
    import torch
    from data import operations as RO
    out = {}
    torch.manual_seed(4242)
    B = 4
    img = torch.rand(B, 3, 16, 16)
    img[0, :, :5, :5] = 0.0
    img[2, 2] *= 0.4
    out["img"] = img.numpy().copy()
    names = ["Invert", "Solarize", "Posterize", "Gray", "Contrast", "AutoContrast", "Saturate", "Brightness", "SamplePairing",
             "Equalize", "Sharpness", "HorizontalFlip", "VerticalFlip"]
    soft_mask = torch.tensor([0.02, 0.97, 0.5, 0.8]).view(B, 1, 1, 1)           # what RelaxedBernoulli(T=0.1) typically returns
    hard_masks = [torch.tensor([1., 0., 1., 1.]), torch.tensor([0., 0., 0., 0.]), torch.tensor([1., 1., 1., 1.])]
    signs01 = torch.tensor([1., 0., 0., 1.])
    real_randint, real_randperm = torch.randint, torch.randperm
    keys = []
    try:
        torch.randint = lambda *a, **k: signs01.clone()
        torch.randperm = lambda n, **k: (torch.arange(n) + 1) % n          # SamplePairing's partner: the next sample (cyclic)
        for name in names:
            cls = getattr(RO, name)
            for ci, (mag0, prob0) in enumerate(((0.3, 0.5), (0.8, 0.9), (1.3, 0.1))):
                try:
                    op = cls(initial_magnitude=mag0, initial_probability=prob0)
                    has_mag = True
                except TypeError:
                    op = cls(initial_probability=prob0)
                    has_mag = False
                if not has_mag and ci > 0:
                    continue
                for mode in (("train", "eval0", "eval1", "eval2"), ("train", "eval0"), ("train",))[ci]:
                    if mode == "train":
                        op.train()
                        mask = soft_mask.clone()
                    else:
                        op.eval()
                        mask = hard_masks[int(mode[-1])].view(B, 1, 1, 1).clone()
                    op.get_mask = lambda batch_size=None, m=mask: m.clone()
                    with torch.no_grad():
                        res = op(img.clone())
                    key = "%s_%d_%s" % (name, ci, mode)
                    keys.append(key)
                    out[key] = res.numpy().copy()
                    out[key + "_mask"] = mask.numpy().copy()
                    out[key + "_cfg"] = np.array([mag0 if has_mag else np.nan, prob0], np.float64)
    finally:
        torch.randint, torch.randperm = real_randint, real_randperm
    out["signs01"] = signs01.numpy()
    out["perm_rule"] = np.array("randperm(n) := (arange(n) + 1) % n")
    out["keys"] = np.array(keys)
    # the straight-through estimator (data/functional.py:21-46): forward value, gradient routing and sum_to_size
    from data import functional as RF
    a = torch.rand(3, 2, 4, 5)
    bparam = torch.rand(3, 1, 1, 1, requires_grad=True)
    y = RF.ste(a, bparam)
    g = torch.rand_like(y)
    y.backward(g)
    out["ste_a"], out["ste_b"], out["ste_y"], out["ste_g"], out["ste_grad_b"] = (a.numpy(), bparam.detach().numpy(), y.detach().numpy(),
                                                                              g.numpy(), bparam.grad.numpy())
    # solarize / posterize route the gradient to the magnitude through ste (data/functional.py:163,181)
    m = torch.tensor([0.4, 0.7, 0.2, 0.9], requires_grad=True)
    r = RF.solarize(img.clone(), m)
    gg = torch.rand_like(r)
    r.backward(gg)
    out["sol_mag"], out["sol_out"], out["sol_g"], out["sol_grad_mag"] = m.detach().numpy(), r.detach().numpy(), gg.numpy(), m.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "operations.npz"), **out)


if __name__ == "__main__":
    install_stubs()
    gen_operations()
    gen_functional()
    gen_ops()
    gen_parse()
    gen_pipeline()
    gen_controller()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
