"""Shared test plumbing: drives aadg_amd's host-side draw code and hands the recorded units either
to the CPU oracle or to the HIP library."""
import json
import os
import random

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Cfg(object):
    class _C(object):
        pass

    def __init__(self, L=2, NUM_MAGS=10, EXCLUDE_OPS=(), EXCLUDE_OPS_NUM=0, SEED=1023, M=6):
        self.CONTROLLER = Cfg._C()
        self.CONTROLLER.L = L
        self.CONTROLLER.NUM_MAGS = NUM_MAGS
        self.CONTROLLER.EXCLUDE_OPS = list(EXCLUDE_OPS)
        self.CONTROLLER.EXCLUDE_OPS_NUM = EXCLUDE_OPS_NUM
        self.CONTROLLER.M = M
        self.SEED = SEED


def load_pipeline_golden():
    z = np.load(os.path.join(GOLDEN, "pipeline.npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


def draw_batch(pool_img, pool_msk, policies, meta, device="cpu"):
    """Re-draws a golden pipeline case with aadg_amd's host code, seeded like make_golden.py.
    Returns (pool, flat_batch, refs, M)."""
    from aadg_amd.data import transform as T
    from aadg_amd.data.basic import DevicePool
    from aadg_amd.data.policy import DGMultiPolicy, parse_policies
    pool = DevicePool(torch.from_numpy(np.ascontiguousarray(pool_img)).to(device),
                      torch.from_numpy(np.ascontiguousarray(pool_msk)).to(device))
    parsed = parse_policies(policies, Cfg(), None)
    ds = meta["dataset"]
    tf = T.Compose([DGMultiPolicy(parsed), T.DGRandomScaleCrop(meta["crop"], scale_range=meta["scale_range"]),
                    T.Normalize_dg(ds), T.ToTensor(ds)])
    random.seed(meta["seed"])
    np.random.seed(meta["seed"])
    batch = []
    for it in range(meta["items"]):
        per_item = []
        for d in range(meta["D"]):
            idx = int(np.random.choice(2, 1)[0])
            s = {'image': pool.image(2 * d + idx), 'label': pool.mask(2 * d + idx), 'img_name': 'im%d_%d' % (d, idx), 'dc': d}
            per_item.append(tf(s))
        batch.append(per_item)
    flat, refs, M = T.collect_refs(batch, nested=True)
    return pool, flat, refs, M


def synth_pool(rs, P, H, W, vessel=False):
    """Synthetic pool like SURVEY 8d: smooth field + colour cast + noise; disc masks."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    imgs = np.empty((P, H, W, 3), np.uint8)
    msks = np.empty((P, H, W), np.uint8)
    for p in range(P):
        k = p % 3
        field = 90 + 60 * np.sin(xx / (3.0 + k + H / 64.0)) * np.cos(yy / (4.0 + k + H / 64.0))
        cast = np.array([1.0, 0.8 - 0.1 * k, 0.5 + 0.15 * k])
        im = field[..., None] * cast + rs.randint(0, 40, (H, W, 3))
        imgs[p] = np.clip(im, 0, 255).astype(np.uint8)
        cy, cx = H * (0.4 + 0.1 * rs.rand()), W * (0.4 + 0.1 * rs.rand())
        rr = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        m = np.full((H, W), 255, np.uint8)
        m[rr < 0.35 * H] = 128
        m[rr < 0.18 * H] = 0
        if vessel:
            m = ((m == 128) * 255).astype(np.uint8)
        msks[p] = m
    return imgs, msks


def random_units(rs, N, P, H, W, crop, scale_range=(1.0, 1.5), L=2, p_scale=0.8):
    """Random but valid unit records (all 10 ops, all magnitude levels), drawn directly."""
    import math
    from aadg_amd._lib import UNIT_DTYPE
    from aadg_amd.data.basic import cutout_rect
    units = np.zeros(N, UNIT_DTYPE)
    units['rect'][:, :, 2:] = -1
    los = [0, 0, 0, 0, 4, .1, .1, .1, .1, 0]
    his = [1, 1, 1, 256, 8, 1.9, 1.9, 1.9, 1.9, .2]
    for i in range(N):
        u = units[i]
        u['src'] = rs.randint(P)
        n_ops = rs.randint(0, L + 1) if i % 7 == 0 else L
        u['n_ops'] = n_ops
        for k in range(n_ops):
            op = rs.randint(10)
            v = rs.randint(10) / 9 * (his[op] - los[op]) + los[op]
            u['op'][k] = op
            if op == 3:
                u['iarg'][k] = int(math.ceil(v))
            elif op == 4:
                u['iarg'][k] = int(v)
            elif op in (5, 6, 7, 8):
                u['farg'][k] = np.float32(v)
            elif op == 9:
                if v > 0:
                    u['rect'][k] = cutout_rect(W, H, v * W, rs.uniform(W), rs.uniform(H))
        w, h = W, H
        if rs.rand() < p_scale:
            w = int(rs.uniform(*scale_range) * W)
            h = int(rs.uniform(*scale_range) * H)
        u['scaled_w'], u['scaled_h'] = w, h
        pad = 0
        if w < crop or h < crop:
            pad = max((crop - w) // 2 + 5, (crop - h) // 2 + 5)
        u['pad'] = pad
        u['crop_x'] = rs.randint(0, w + 2 * pad - crop + 1)
        u['crop_y'] = rs.randint(0, h + 2 * pad - crop + 1)
    return units
