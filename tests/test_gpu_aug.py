"""GPU parity tests for the uint8 augmentation path: HIP (through the C ABI) vs the oracle and vs
the reference's golden vectors.  Bit-exact everywhere (integer / Pillow fixed-point work; the float32
outputs are a pure function of a byte)."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, draw_batch, load_pipeline_golden, random_units, synth_pool

pytestmark = pytest.mark.gpu


def test_eager_ops_vs_reference_golden(hip):
    """aadg_op_u8 (the registry op entry point) against reference apply_augment outputs."""
    from aadg_amd.data import basic
    z = np.load(os.path.join(GOLDEN, "u8_ops.npz"))
    names = [str(n) for n in z["op_names"]]
    for tag in ("a", "b"):
        img = torch.from_numpy(z["img_" + tag]).cuda()
        msk = torch.from_numpy(z["mask_" + tag]).cuda()
        for oi, name in enumerate(names):
            for li in range(10):
                np.random.seed(1000 * oi + li)
                o, m = basic.apply_augment(img, msk, name, li / 9)
                assert np.array_equal(o.cpu().numpy(), z["out_" + tag][oi, li]), (tag, name, li)
                assert np.array_equal(m.cpu().numpy(), z["mout_" + tag][oi, li]), (tag, name, li)


@pytest.mark.parametrize("case", range(4))
def test_pipeline_seed_for_seed_vs_reference(hip, case):
    """Whole collate on the GPU, seeded like the reference run: identical tensors."""
    from aadg_amd.data import transform as T
    z, meta = load_pipeline_golden()
    m = meta[case]
    name = m["name"]
    pool, flat, refs, M = draw_batch(z[name + "_pool_img"], z[name + "_pool_msk"], z[name + "_policies"], m, "cuda")
    img, lbl = T.materialize(refs)
    S = len(flat)
    lut = z["lut256"]
    img, lbl = img.cpu().numpy(), lbl.cpu().numpy()
    assert np.array_equal(img[:S], lut[z[name + "_image"]])
    assert np.array_equal(lbl[:S], z[name + "_label"].astype(np.float32))
    assert np.array_equal(img[S:], lut[z[name + "_aug_images"]])
    assert np.array_equal(lbl[S:], z[name + "_aug_labels"].astype(np.float32))


@pytest.mark.parametrize("H,W,crop,sr,kind,N", [
    (64, 64, 64, (1.0, 1.5), 0, 96),      # optic-like, upscale
    (40, 52, 48, (1.0, 1.5), 0, 64),      # non-square source, pad on one axis
    (96, 96, 64, (0.5, 2.0), 1, 96),      # rvs-like, antialiased downscale, K=1
    (30, 30, 50, (0.5, 2.0), 1, 48),      # always padded
    (33, 35, 31, (1.0, 1.5), 0, 40),      # odd sizes: scalar (non-vector) code paths
    (256, 256, 256, (1.0, 1.5), 0, 36),   # the reference's own size
    (256, 256, 256, (0.5, 2.0), 1, 36),   # RVS scale range at the reference's size: generic (down-scaling) fused tiles
    (128, 128, 96, (0.34, 0.6), 1, 24),   # scale < 1/2: staged flow with up to 7 taps
])
def test_random_units_vs_oracle(hip, oracle, H, W, crop, sr, kind, N):
    rs = np.random.RandomState(H * 1000 + W)
    P = 5
    imgs, msks = synth_pool(rs, P, H, W, vessel=(kind == 1))
    imgs[1] = rs.randint(0, 256, imgs[1].shape)   # one pure-noise image
    imgs[2][...] = 93                              # one constant image (degenerate histograms)
    units = random_units(rs, N, P, H, W, crop, sr)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, kind)
    got_img, got_lbl = hip.aug_u8_forward(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda(), units, crop, kind)
    got_img, got_lbl = got_img.cpu().numpy(), got_lbl.cpu().numpy()
    bad = [i for i in range(N) if not np.array_equal(got_img[i], want_img[i])]
    assert not bad, "image mismatch in units %s: %s" % (bad[:5], units[bad[:1]])
    assert np.array_equal(got_lbl, want_lbl)


def test_three_and_four_op_chains(hip, oracle):
    rs = np.random.RandomState(3)
    imgs, msks = synth_pool(rs, 4, 48, 48)
    units = random_units(rs, 64, 4, 48, 48, 48, (1.0, 1.5), L=4)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, 48, 0)
    got_img, got_lbl = hip.aug_u8_forward(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda(), units, 48, 0)
    assert np.array_equal(got_img.cpu().numpy(), want_img)
    assert np.array_equal(got_lbl.cpu().numpy(), want_lbl)


def test_full_size_properties(hip, oracle):
    """BASELINE size (512x512, N=144): identity round trip, determinism, and an oracle spot check."""
    rs = np.random.RandomState(1023)
    H = W = crop = 512
    P, N = 24, 144
    imgs, msks = synth_pool(rs, P, H, W)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    # (a) no ops, no scale, full crop == u8/127.5-1 of the source, exactly
    ident = np.zeros(P, hip.UNIT_DTYPE)
    ident['rect'][:, :, 2:] = -1
    ident['src'] = np.arange(P)
    ident['scaled_w'], ident['scaled_h'] = W, H
    oi, ol = hip.aug_u8_forward(d_img, d_msk, ident, crop, 0)
    lut = (np.arange(256, dtype=np.float32) / np.float32(127.5)) - np.float32(1.0)
    assert np.array_equal(oi.cpu().numpy(), lut[imgs].transpose(0, 3, 1, 2))
    assert np.array_equal(ol[:, 0].cpu().numpy(), (msks <= 50).astype(np.float32))
    assert np.array_equal(ol[:, 1].cpu().numpy(), (msks <= 200).astype(np.float32))
    # (b) Invert twice is the identity (an involution through two full op stages)
    inv2 = ident.copy()
    inv2['n_ops'] = 2
    inv2['op'][:, 0] = inv2['op'][:, 1] = 1
    oi2, _ = hip.aug_u8_forward(d_img, d_msk, inv2, crop, 0)
    assert torch.equal(oi, oi2)
    # (c) the real workload: deterministic, and equal to the oracle on a sample of units
    units = random_units(rs, N, P, H, W, crop, (1.0, 1.5))
    a_img, a_lbl = hip.aug_u8_forward(d_img, d_msk, units, crop, 0)
    b_img, b_lbl = hip.aug_u8_forward(d_img, d_msk, units, crop, 0)
    assert torch.equal(a_img, b_img) and torch.equal(a_lbl, b_lbl)
    sel = np.arange(0, N, 12)
    w_img, w_lbl = oracle.aug_units(imgs, msks, units[sel], crop, 0)
    assert np.array_equal(a_img[torch.from_numpy(sel).cuda()].cpu().numpy(), w_img)
    assert np.array_equal(a_lbl[torch.from_numpy(sel).cuda()].cpu().numpy(), w_lbl)


def test_config2_rvs_1024_spot_check(hip, oracle):
    """BASELINE configs[2] geometry: 1024x1024 sources and crops, RVS scale range [0.5, 2], K = 1.
    Oracle comparison on a handful of units (the oracle needs ~0.2 s per 1024^2 unit)."""
    rs = np.random.RandomState(77)
    H = W = crop = 1024
    imgs, msks = synth_pool(rs, 3, H, W, vessel=True)
    units = random_units(rs, 10, 3, H, W, crop, (0.5, 2.0))
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 1)
    got_img, got_lbl = hip.aug_u8_forward(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda(), units, crop, 1)
    assert np.array_equal(got_img.cpu().numpy(), want_img)
    assert np.array_equal(got_lbl.cpu().numpy(), want_lbl)


def test_excluded_ops_and_magnitude_grid_through_the_pipeline(hip, oracle):
    """CONTROLLER.EXCLUDE_OPS shifts the op indexing (data/policy.py:72-74); NUM_MAGS changes the level grid."""
    import random
    from helpers import Cfg
    from aadg_amd.data import transform as T
    from aadg_amd.data.basic import DevicePool
    from aadg_amd.data.policy import DGMultiPolicy, parse_policies
    rs = np.random.RandomState(5)
    imgs, msks = synth_pool(rs, 3, 48, 48)
    pool = DevicePool(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda())
    cfg = Cfg(L=3, NUM_MAGS=7, EXCLUDE_OPS=["Invert", "Cutout"])
    pol = np.zeros((4, 5 * 3 * 2), np.int64)
    pol[:, 0::2] = rs.randint(0, 8, (4, 15))
    pol[:, 1::2] = rs.randint(0, 7, (4, 15))
    parsed = parse_policies(pol, cfg, None)
    assert all(name not in ("Invert", "Cutout") for p in parsed for sp in p for name, _ in sp)
    tf = T.Compose([DGMultiPolicy(parsed), T.DGRandomScaleCrop(48), T.Normalize_dg('optic'), T.ToTensor('optic')])
    random.seed(9)
    np.random.seed(9)
    batch = [[tf({'image': pool.image(d), 'label': pool.mask(d), 'img_name': 'x', 'dc': d}) for d in range(3)]]
    flat, refs, M = T.collect_refs(batch, nested=True)
    units = T.refs_to_units(refs)
    assert int(units['n_ops'][3:].max()) == 3
    got_img, got_lbl = T.materialize(refs)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, 48, 0)
    assert np.array_equal(got_img.cpu().numpy(), want_img) and np.array_equal(got_lbl.cpu().numpy(), want_lbl)


def test_bad_arguments_raise(hip):
    imgs = torch.zeros((2, 16, 16, 3), dtype=torch.uint8, device="cuda")
    msks = torch.zeros((2, 16, 16), dtype=torch.uint8, device="cuda")
    u = np.zeros(1, hip.UNIT_DTYPE)
    u['scaled_w'] = u['scaled_h'] = 16
    u['src'] = 5
    with pytest.raises(hip.AadgError):
        hip.aug_u8_forward(imgs, msks, u, 16, 0)
    u['src'] = 0
    u['scaled_w'] = 4  # below 1/3 scale
    with pytest.raises(hip.AadgError):
        hip.aug_u8_forward(imgs, msks, u, 16, 0)
    with pytest.raises(hip.AadgError):
        hip.aug_u8_forward(imgs.cpu(), msks.cpu(), u, 16, 0)


def test_class_lists_and_fallback_agree(hip, oracle):
    """aadg_aug_u8_forward_ex2 with the caller's class lists (one workgroup per tile of a unit of the kernel's class) and the
    list-free entry (every unit offered to every kernel, the wrong class returns) produce the same tensors, and both are the
    oracle's -- on a batch that mixes plain, Sharpness (8-row half tiles), generic and staged units, at a crop that is not a
    multiple of the 16-row tile."""
    import ctypes
    from helpers import random_units, synth_pool
    rs = np.random.RandomState(77)
    P, H, crop, N = 6, 72, 72, 40
    imgs, msks = synth_pool(rs, P, H, H)
    units = random_units(rs, N, P, H, H, crop, (0.36, 1.6))
    units['op'][:10, 0] = 8; units['farg'][:10, 0] = np.float32(1.4); units['n_ops'][:10] = np.maximum(units['n_ops'][:10], 1)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 0)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    got_img, got_lbl = hip.aug_u8_forward(d_img, d_msk, units, crop, 0)                     # ex2 with lists
    classes, stats, order, counts, stat_lists = hip.launch_hints(units, H, H, crop)
    assert sum(len(l) for l in stat_lists) > 0
    assert sum(counts[:3]) <= N and counts[0] > 0 and counts[1] > 0 and counts[2] > counts[3] > 0 and sum(counts[:3]) < N     # all five classes present
    assert np.array_equal(got_img.cpu().numpy(), want_img) and np.array_equal(got_lbl.cpu().numpy(), want_lbl)
    lib = hip.load()
    d_units = hip.units_to_device(units, d_img.device)
    o_img = torch.empty_like(got_img); o_lbl = torch.empty_like(got_lbl)
    nb = lib.aadg_aug_u8_workspace_bytes(N, H, H, crop)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    rc = lib.aadg_aug_u8_forward(d_img.data_ptr(), d_msk.data_ptr(), P, H, H, d_units.data_ptr(), N, 4, crop, 0, o_img.data_ptr(),
                                 o_lbl.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(o_img, got_img) and torch.equal(o_lbl, got_lbl)


def test_statistics_by_pushforward_long_chains(hip, oracle):
    """AutoContrast / Equalize behind per-channel byte maps take their histogram from the RAW image's, pushed through the earlier
    stages' LUTs (no pixel pass).  Chains of up to 4 ops: push-forward at slots 1, 2 and 3, chains broken by Color / Cutout / Sharpness
    (pixel pass again), Contrast (needs the mean of L: always a pixel pass), with and without the caller's work lists -- bit-exact vs the oracle."""
    from helpers import synth_pool
    from aadg_amd.data.basic import cutout_rect
    rs = np.random.RandomState(5)
    P, H, crop = 4, 64, 64
    imgs, msks = synth_pool(rs, P, H, H)
    chains = [[(7, 1.4), (0, 0)], [(1, 0), (3, 100), (2, 0)], [(4, 5), (7, 0.6), (5, 1.3), (0, 0)], [(2, 0), (0, 0), (2, 0), (0, 0)],
              [(6, 1.5), (2, 0)], [(7, 1.2), (8, 1.6), (0, 0)], [(9, 0.2), (0, 0)], [(3, 64), (5, 0.7)], [(5, 1.5), (2, 0)], [(0, 0), (1, 0), (4, 4), (2, 0)]]
    N = 2 * len(chains)
    units = np.zeros(N, hip.UNIT_DTYPE)
    units['rect'][:, :, 2:] = -1
    for i in range(N):
        ch = chains[i % len(chains)]
        u = units[i]
        u['src'] = rs.randint(P)
        u['n_ops'] = len(ch)
        for k, (op, v) in enumerate(ch):
            u['op'][k] = op
            if op in (3, 4):
                u['iarg'][k] = int(v)
            elif op in (5, 6, 7, 8):
                u['farg'][k] = np.float32(v)
            elif op == 9:
                u['rect'][k] = cutout_rect(H, H, v * H, 20.0, 30.0)
        w = h = H if i < len(chains) else int(H * 1.3)          # identity scale and up-scaling (fused flow)
        u['scaled_w'], u['scaled_h'] = w, h
        u['crop_x'], u['crop_y'] = (w - crop) // 2, (h - crop) // 2
    classes, stats_mask, order, counts, stat_lists = hip.launch_hints(units, H, H, crop)
    assert len(stat_lists[0]) == 12 and len(stat_lists[3]) == 0           # raw histograms: statistics op in slot 0 or a push-forward source; slot 3 never needs pixels here
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 0)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    got_img, got_lbl = hip.aug_u8_forward(d_img, d_msk, units, crop, 0)
    assert np.array_equal(got_img.cpu().numpy(), want_img) and np.array_equal(got_lbl.cpu().numpy(), want_lbl)
    lib = hip.load()
    d_units = hip.units_to_device(units, d_img.device)
    o_img = torch.empty_like(got_img); o_lbl = torch.empty_like(got_lbl)
    ws = torch.empty(lib.aadg_aug_u8_workspace_bytes(N, H, H, crop), dtype=torch.uint8, device="cuda")
    assert lib.aadg_aug_u8_forward(d_img.data_ptr(), d_msk.data_ptr(), P, H, H, d_units.data_ptr(), N, 4, crop, 0, o_img.data_ptr(),
                                   o_lbl.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(o_img, got_img) and torch.equal(o_lbl, got_lbl)


def test_pool_histograms_are_pil_statistics(hip):
    """aadg_pool_histograms_u8: per channel Image.histogram() and the sum behind ImageStat.Stat(convert('L')).mean, per pool image
    (vector and scalar code paths, noise / constant images)."""
    from helpers import synth_pool
    for (P, H, W) in [(5, 64, 64), (3, 33, 35)]:
        rs = np.random.RandomState(H)
        imgs, _ = synth_pool(rs, P, H, W)
        imgs[1] = rs.randint(0, 256, imgs[1].shape)
        imgs[2][...] = 93
        hist = hip.pool_histograms(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint32)
        assert hist.shape == (P, hip.HIST_STRIDE)
        for p in range(P):
            for c in range(3):
                assert np.array_equal(hist[p, 256 * c:256 * (c + 1)], np.bincount(imgs[p, :, :, c].reshape(-1), minlength=256))
            r, g, b = (imgs[p, :, :, c].astype(np.int64) for c in range(3))
            lsum = int(((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).sum())      # ImagingConvert rgb2l
            assert int(hist[p, 768]) + (int(hist[p, 769]) << 32) == lsum


@pytest.mark.parametrize("H,crop,sr,L", [(64, 64, (1.0, 1.5), 2), (64, 64, (1.0, 1.5), 4), (96, 64, (0.5, 2.0), 3), (72, 72, (0.36, 1.6), 4),
                                         (33, 31, (1.0, 1.5), 3)])
def test_pool_statistics_cache_changes_nothing(hip, oracle, H, crop, sr, L):
    """With pool_hist (statistics of the raw pool images, computed once) the call skips its stage-0 histogram pass, builds every byte
    map that waits for no pixel pass together with the tables and -- when the batch holds no staged unit (the first three cases) --
    re-does only the late units' maps behind the histogram passes; results are the oracle's and bit-identical to the uncached call -- fused, generic and staged units, statistics ops in every slot, push-forward chains, noise and constant
    images."""
    from helpers import random_units, synth_pool
    rs = np.random.RandomState(100 + H)
    P, N = 6, 96
    imgs, msks = synth_pool(rs, P, H, H)
    imgs[1] = rs.randint(0, 256, imgs[1].shape)
    imgs[2][...] = 93
    units = random_units(rs, N, P, H, H, crop, sr, L=L)
    # make sure statistics ops sit in slot 0 and behind byte maps
    units['op'][:12, 0] = np.tile([0, 2, 5], 4); units['farg'][:12, 0] = np.float32(1.3); units['n_ops'][:12] = np.maximum(units['n_ops'][:12], 1)
    units['op'][12:20, 0] = 1; units['op'][12:20, 1] = np.tile([0, 2], 4); units['n_ops'][12:20] = np.maximum(units['n_ops'][12:20], 2)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 0)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    plain = hip.aug_u8_forward(d_img, d_msk, units, crop, 0)
    ph = hip.pool_histograms(d_img)
    late = hip.launch_plan(units, H, H, crop)[5]
    assert late.size > 0
    cached = hip.aug_u8_forward(d_img, d_msk, units, crop, 0, pool_hist=ph)
    assert np.array_equal(cached[0].cpu().numpy(), want_img) and np.array_equal(cached[1].cpu().numpy(), want_lbl)
    assert torch.equal(plain[0], cached[0]) and torch.equal(plain[1], cached[1])
    for _ in range(3):                                   # back to back on one workspace
        again = hip.aug_u8_forward(d_img, d_msk, units[::-1].copy(), crop, 0, pool_hist=ph)
    assert torch.equal(again[0], cached[0].flip(0)) and torch.equal(again[1], cached[1].flip(0))
    with pytest.raises(hip.AadgError):
        hip.aug_u8_forward(d_img, d_msk, units, crop, 0, pool_hist=ph[:2])


def test_device_pool_statistics_follow_in_place_writes(hip):
    """DevicePool.histograms() is computed once per resident pool and recomputed after an in-place write to the images."""
    from helpers import synth_pool
    from aadg_amd.data.basic import DevicePool
    rs = np.random.RandomState(9)
    imgs, msks = synth_pool(rs, 3, 32, 32)
    pool = DevicePool(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda())
    h0 = pool.histograms()
    assert pool.histograms() is h0                                        # cached
    pool.images[1].fill_(7)
    h1 = pool.histograms()
    assert h1 is not h0 and int(h1[1, 7]) == 32 * 32 and int(h1[1, 256 + 7]) == 32 * 32 and torch.equal(h1[0], h0[0])
    cpu = DevicePool(torch.from_numpy(imgs), torch.from_numpy(msks))
    assert cpu.histograms() is None                                       # CPU pools (tests): the statistics passes run per call


def _bench_batch(cfg_rel, size, per_domain):
    """The seeded batch plan of a bench.py leg (same yaml, seed 1023, policies RandomState(1023).randint): pool + unit records."""
    import os
    import random
    from aadg_amd.config.defaults import get_default_config
    from aadg_amd.data import transform as T
    from aadg_amd.data.dataloader import get_seg_dg_dataloader
    from aadg_amd.data.policy import DGMultiPolicy, parse_policies
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_default_config()
    cfg.defrost()
    cfg.merge_from_file(os.path.join(root, cfg_rel))
    cfg.SEED = 1023
    cfg.freeze()

    class A(object):
        pass
    args = A()
    args.crop_size, args.epoch_items = size, 8
    for fn in (random.seed, np.random.seed, torch.manual_seed):
        fn(1023)
    _, loader, _ = get_seg_dg_dataloader(cfg, args, 8, 0, per_domain=per_domain)
    pol = np.random.RandomState(1023).randint(0, 10, (6, 20))
    loader.dataset.transforms.transforms[0] = DGMultiPolicy(parse_policies(pol, cfg, None))
    batch = [loader.dataset[0] for _ in range(8)]
    flat, refs, M = T.collect_refs(batch, nested=True)
    return loader.dataset.pool, T.refs_to_units(refs), (0 if cfg.DATASET.NAME == "optic" else 1)


def test_bench_batch_512_every_unit_vs_oracle(hip, oracle):
    """Round 3: ALL 24 + 144 units of the seeded BASELINE configs[1] batch (512 x 512, Fundus pipeline: scale range [1, 1.5], K = 2)
    against the oracle, bit for bit -- not a sample of them."""
    pool, units, ds = _bench_batch(os.path.join("experiments", "optic_sinkhorn", "diversity.yaml"), 512, 8)
    assert len(units) == 24 + 144
    got_img, got_lbl = hip.aug_u8_forward(pool.images, pool.masks, units, 512, ds, pool_hist=pool.histograms())
    want_img, want_lbl = oracle.aug_units(pool.images.cpu().numpy(), pool.masks.cpu().numpy(), units, 512, ds)
    assert np.array_equal(got_img.cpu().numpy(), want_img) and np.array_equal(got_lbl.cpu().numpy(), want_lbl)


def test_bench_batch_rvs_1024_every_unit_vs_oracle(hip, oracle):
    """Round 4 (round 3 checked every third unit): ALL 168 units of the seeded BASELINE configs[2] batch (RVS pipeline, 1024 x 1024
    crops, scale range [0.5, 2], K = 1) against the oracle, bit for bit -- up-scaling tiles, the two-pass down-scaling flow with and
    without stencils, large pad regions -- in ONE library call (the chunked intermediate, the class lists and the late list as the
    bench uses them); the oracle works through the batch in slices to bound its memory."""
    pool, units, ds = _bench_batch(os.path.join("experiments", "rvs_sinkhorn", "diversity_ex.yaml"), 1024, 8)
    assert len(units) == 168 and ds == 1
    counts = hip.launch_hints(units, 1024, 1024, 1024)[3]
    assert counts[0] > 0 and counts[1] > 0 and counts[2] > counts[3] > 0            # every tile class is in the batch
    got_img, got_lbl = hip.aug_u8_forward(pool.images, pool.masks, units, 1024, ds, pool_hist=pool.histograms())
    imgs, msks = pool.images.cpu().numpy(), pool.masks.cpu().numpy()
    bad = []
    for lo in range(0, 168, 24):
        want_img, want_lbl = oracle.aug_units(imgs, msks, units[lo:lo + 24], 1024, ds)
        gi, gl = got_img[lo:lo + 24].cpu().numpy(), got_lbl[lo:lo + 24].cpu().numpy()
        bad += [lo + i for i in range(len(want_img)) if not (np.array_equal(gi[i], want_img[i]) and np.array_equal(gl[i], want_lbl[i]))]
    assert not bad, bad
    # a sub-batch (other chunking, other list positions) gives the same bytes for the same units
    sel = np.arange(1, 168, 5)
    s_img, s_lbl = hip.aug_u8_forward(pool.images, pool.masks, units[sel], 1024, ds, pool_hist=pool.histograms())
    idx = torch.from_numpy(sel).cuda()
    assert torch.equal(s_img, got_img[idx]) and torch.equal(s_lbl, got_lbl[idx])


def test_tap_class_boundaries_vs_oracle(hip, oracle):
    """Round 3: the down-scaling passes use 3 / 4 / 5 resampling taps by scale (csrc/aug_u8.hip: axis_taps).  Every size around the class
    boundaries of a 96-pixel axis -- 3 w > 2 W (65 | 64), 2 w > W (49 | 48: exactly half keeps Pillow's five slots), the unscaled and the
    up-scaled axis -- on either axis, with and without a Sharpness stencil in front, against the oracle bit for bit."""
    from aadg_amd._lib import UNIT_DTYPE
    rs = np.random.RandomState(21)
    H = W = crop = 96
    P = 4
    imgs, msks = synth_pool(rs, P, H, W, vessel=True)
    sizes = [48, 49, 50, 63, 64, 65, 66, 80, 95, 96, 97, 130, 192]
    recs = []
    for i, (w, h) in enumerate([(a, b) for a in sizes for b in sizes if min(a, b) < 96]):
        u = np.zeros((), UNIT_DTYPE)
        u["rect"][:, 2:] = -1
        u["src"] = i % P
        if i % 3 == 1:
            u["n_ops"] = 1; u["op"][0] = 8; u["farg"][0] = np.float32(1.6)
        elif i % 3 == 2:
            u["n_ops"] = 2; u["op"][0] = 7; u["farg"][0] = np.float32(1.3); u["op"][1] = 8; u["farg"][1] = np.float32(0.4)
        u["scaled_w"], u["scaled_h"] = w, h
        pad = max((crop - w) // 2 + 5, (crop - h) // 2 + 5) if (w < crop or h < crop) else 0
        u["pad"] = pad
        u["crop_x"] = rs.randint(0, w + 2 * pad - crop + 1)
        u["crop_y"] = rs.randint(0, h + 2 * pad - crop + 1)
        recs.append(u)
    units = np.array(recs, dtype=UNIT_DTYPE)
    got_img, got_lbl = hip.aug_u8_forward(torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda(), units, crop, 1)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 1)
    bad = [i for i in range(len(units)) if not np.array_equal(got_img[i].cpu().numpy(), want_img[i])]
    assert not bad, [(int(units[i]["scaled_w"]), int(units[i]["scaled_h"])) for i in bad[:8]]
    assert np.array_equal(got_lbl.cpu().numpy(), want_lbl)


def test_statistics_pass_is_correct_for_any_list_split(hip, oracle):
    """aadg_aug_lists.n_stat_stencil (ABI 7) tells the statistics pass which units of a slot's list -- the first ones -- have a
    Sharpness stencil in front of the op; the promise of include/aadg_hip.h is that a wrong split costs time, never correctness (the
    kernel chooses the data flow from the unit record).  The same late units -- byte map / Color / Cutout / Sharpness in front of
    Contrast / AutoContrast / Equalize, on a 200 x 264 image (partial last block row, a second, narrow column strip) -- with the
    planner's split, with no unit declared a stencil unit and with every unit declared one: identical bytes, equal to the oracle."""
    from aadg_amd._lib import UNIT_DTYPE
    rs = np.random.RandomState(5)
    H, W, crop, P = 200, 264, 200, 5
    imgs = np.stack([synth_pool(rs, 1, H, W)[0][0] for _ in range(P)])
    msks = np.stack([synth_pool(rs, 1, H, W)[1][0] for _ in range(P)])
    recs = []
    from aadg_amd.data.basic import cutout_rect
    for op0 in (7, 6, 9, 8, 3):
        for op1 in (5, 0, 2):
            for rep in range(2):
                u = np.zeros((), UNIT_DTYPE)
                u["rect"][:, 2:] = -1
                u["src"] = rs.randint(P)
                u["n_ops"] = 2
                u["op"][0] = op0
                u["farg"][0] = np.float32(1.4 if op0 != 8 else (1.7, 0.3)[rep])
                if op0 == 9:
                    u["rect"][0] = cutout_rect(W, H, 0.15 * W, rs.uniform(W), rs.uniform(H))
                if op0 == 3:
                    u["iarg"][0] = 100
                u["op"][1] = op1
                u["farg"][1] = np.float32(1.3)
                u["scaled_w"], u["scaled_h"] = (W, H) if rep == 0 else (int(1.2 * W), int(1.3 * H))
                u["crop_x"] = rs.randint(0, int(u["scaled_w"]) - crop + 1)
                u["crop_y"] = rs.randint(0, int(u["scaled_h"]) - crop + 1)
                recs.append(u)
    units = np.array(recs, dtype=UNIT_DTYPE)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    ph = hip.pool_histograms(d_img)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 1)
    base = hip.aug_u8_forward(d_img, d_msk, units, crop, 1, pool_hist=ph)
    assert np.array_equal(base[0].cpu().numpy(), want_img) and np.array_equal(base[1].cpu().numpy(), want_lbl)
    n_sten = hip.launch_plan(units, H, W, crop)[6]
    assert 0 < n_sten[1] < hip.launch_plan(units, H, W, crop)[4][1].size               # a mixed list
    for mode in ("none", "all"):
        got = _aug_with_forced_split(hip, d_img, d_msk, units, crop, ph, mode)
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]), mode


def _aug_with_forced_split(hip, d_img, d_msk, units, crop, ph, mode):
    """aug_u8_forward with aadg_aug_lists.n_stat_stencil overridden ('none': 0 for every slot, 'all': n_stat) -- the planner is wrapped
    for the duration of the call so that its summary carries the forced counts."""
    import ctypes
    from aadg_amd import _lib
    lib = _lib.load()
    real = lib.aadg_aug_u8_plan
    K = _lib.MAX_OPS

    def plan(*args):
        rc = real(*args)
        summary = args[-1]
        for k in range(K):
            summary[8 + K + k] = 0 if mode == "none" else summary[8 + k]
        return rc

    class Proxy(object):
        def __getattr__(self, name):
            return plan if name == "aadg_aug_u8_plan" else getattr(lib, name)
    orig_load = _lib.load
    _lib.load = lambda: Proxy()
    try:
        return hip.aug_u8_forward(d_img, d_msk, units, crop, 1, pool_hist=ph)
    finally:
        _lib.load = orig_load


@pytest.mark.parametrize("H,W,crop,kind", [(176, 320, 200, 0), (256, 256, 256, 1), (96, 520, 132, 1)])
def test_width_only_units_one_pass_tile_vs_oracle(hip, oracle, H, W, crop, kind):
    """Round 5 (ABI 9): a down-scaling unit that shrinks the WIDTH only and chains no Sharpness stencil runs through the one-pass tile
    k_fused3w (128 x 16 outputs; horizontal pass of 3 / 4 / 5 taps into LDS, up-scaling vertical pass) instead of the two passes.  Units of
    every tap class with byte-map / Color / Cutout ops in front, scaled heights from 1x to 2x, crops wider than one tile with a partial last
    tile and pad columns / rows, both datasets -- against the oracle bit for bit, with the planner's lists and through the list-free entry
    (every unit offered to every tile kernel: the two-pass kernels must leave these units alone)."""
    from aadg_amd._lib import UNIT_DTYPE
    rs = np.random.RandomState(H + W + crop)
    P = 4
    imgs, msks = synth_pool(rs, P, H, W, vessel=(kind == 1))
    widths = sorted(set([W // 2, W // 2 + 1, (2 * W) // 3, (2 * W) // 3 + 1, (3 * W) // 4, W - 1, int(W * 0.55), int(W * 0.9)]))
    heights = [H, H + 1, int(H * 1.3), 2 * H]
    recs = []
    for i, (w, h) in enumerate([(a, b) for a in widths for b in heights]):
        u = np.zeros((), UNIT_DTYPE)
        u["rect"][:, 2:] = -1
        u["src"] = i % P
        if i % 4 == 1:
            u["n_ops"] = 2; u["op"][0] = 1; u["op"][1] = 6; u["farg"][1] = np.float32(1.5)          # Invert, Color
        elif i % 4 == 2:
            u["n_ops"] = 2; u["op"][0] = 0; u["op"][1] = 9; u["rect"][1] = (W // 5, H // 4, W // 2, H // 2)     # AutoContrast, Cutout
        elif i % 4 == 3:
            u["n_ops"] = 1; u["op"][0] = 5; u["farg"][0] = np.float32(1.4)                            # Contrast
        u["scaled_w"], u["scaled_h"] = w, h
        pad = max((crop - w) // 2 + 5, (crop - h) // 2 + 5) if (w < crop or h < crop) else 0
        u["pad"] = pad
        u["crop_x"] = rs.randint(0, w + 2 * pad - crop + 1)
        u["crop_y"] = rs.randint(0, h + 2 * pad - crop + 1)
        recs.append(u)
    units = np.array(recs, dtype=UNIT_DTYPE)
    plan = hip.launch_plan(units, H, W, crop)
    assert plan[7] == len(units) and plan[3][2] == len(units)                   # every unit is a width-only generic unit
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    got_img, got_lbl = hip.aug_u8_forward(d_img, d_msk, units, crop, kind)
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, kind)
    bad = [i for i in range(len(units)) if not np.array_equal(got_img[i].cpu().numpy(), want_img[i])]
    assert not bad, [(int(units[i]["scaled_w"]), int(units[i]["scaled_h"])) for i in bad[:8]]
    assert np.array_equal(got_lbl.cpu().numpy(), want_lbl)
    # the list-free entry
    lib = hip.load()
    d_units = hip.units_to_device(units, d_img.device)
    o_img = torch.empty_like(got_img); o_lbl = torch.empty_like(got_lbl)
    ws = torch.empty(lib.aadg_aug_u8_workspace_bytes(len(units), H, W, crop), dtype=torch.uint8, device="cuda")
    rc = lib.aadg_aug_u8_forward(d_img.data_ptr(), d_msk.data_ptr(), P, H, W, d_units.data_ptr(), len(units), 4, crop, kind, o_img.data_ptr(),
                                 o_lbl.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(o_img, got_img) and torch.equal(o_lbl, got_lbl)


@pytest.mark.parametrize("H,crop,sr", [(64, 64, (1.0, 1.5)), (96, 64, (0.5, 2.0))])
def test_helper_stream_fork_is_the_same_call(hip, oracle, H, crop, sr):
    """ABI 12: with cached pool statistics the library runs the late units' chain (histogram pass, byte maps) and their tiles on a helper
    stream beside the tile kernel of the other units.  The planner lists the late units last inside the plain / Sharpness class; the call
    with the fork (default) equals the one-stream call (`_lib.AUG_FORK = False`) and the oracle bit for bit, back to back on one workspace
    and with other work queued on the caller's stream in front of and behind it."""
    from helpers import random_units, synth_pool
    rs = np.random.RandomState(7 + H)
    P, N = 5, 120
    imgs, msks = synth_pool(rs, P, H, H)
    units = random_units(rs, N, P, H, H, crop, sr, L=3)
    # late units of both up-scaling classes: a statistics op behind Color (2: equalize behind 6: color) / behind Sharpness
    units['op'][:30, 0] = 6; units['farg'][:30, 0] = np.float32(1.4); units['op'][:30, 1] = np.tile([0, 2, 5], 10); units['farg'][:30, 1] = np.float32(1.2)
    units['n_ops'][:30] = np.maximum(units['n_ops'][:30], 2)
    units['op'][30:44, 0] = 8; units['farg'][30:44, 0] = np.float32(1.6); units['op'][30:44, 1] = 5; units['farg'][30:44, 1] = np.float32(0.7)
    units['n_ops'][30:44] = np.maximum(units['n_ops'][30:44], 2)
    plan = hip.launch_plan(units, H, H, crop)
    n_plain, n_sharp = plan[3][0], plan[3][1]
    npe, nse = plan[8]
    npl, nsl = n_plain - npe, n_sharp - nse
    assert plan[5].size > 0 and (npl > 0 or nsl > 0) and npe + nse > 0                               # something to fork, something beside it
    order, late = plan[2], set(plan[5].tolist())
    assert all(int(u) in late for u in order[n_plain - npl:n_plain]) and not any(int(u) in late for u in order[:n_plain - npl])
    want_img, want_lbl = oracle.aug_units(imgs, msks, units, crop, 0)
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    ph = hip.pool_histograms(d_img)
    busy = torch.randn(1 << 22, device="cuda")
    outs = {}
    for fork in (True, False, True):
        hip.AUG_FORK = fork
        try:
            for _ in range(3):
                busy.mul_(1.0001)                              # work in front of the call on the caller's stream
                got = hip.aug_u8_forward(d_img, d_msk, units, crop, 0, pool_hist=ph)
                busy.add_(1e-6)                                # ... and behind it
        finally:
            hip.AUG_FORK = True
        torch.cuda.synchronize()
        assert np.array_equal(got[0].cpu().numpy(), want_img) and np.array_equal(got[1].cpu().numpy(), want_lbl), fork
        outs[fork] = got
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
