"""Synthetic multi-domain segmentation pools with the reference's sample-dict schema
(data/optic.py:79-103: `__getitem__` returns a LIST with one sample per source domain, each
{'image', 'label', 'img_name', 'dc'} pushed through `self.transforms`).

The real Fundus / RVS images are not available in this environment (datasets are
out of scope); images follow SURVEY.md 8d: per domain a pool of uint8 RGB images = smooth low-frequency
field + per-domain colour cast + uniform noise (non-degenerate histograms), masks = two concentric discs
(0 / 128 / 255, optic) or thin polylines (0 / 255, rvs).  Pools live in HBM as one DevicePool.
"""
import numpy as np
import torch

from .basic import DevicePool


def make_pool(seed, n_domains, per_domain, H, W, dataset='optic'):
    rs = np.random.RandomState(seed)
    P = n_domains * per_domain
    imgs = np.empty((P, H, W, 3), np.uint8)
    msks = np.empty((P, H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for d in range(n_domains):
        cast = np.array([1.0, 0.85 - 0.12 * (d % 4), 0.45 + 0.13 * (d % 5)], np.float32)
        for i in range(per_domain):
            p = d * per_domain + i
            fx, fy = rs.uniform(0.5, 2.0, 2) * (2 * np.pi / max(H, W))
            field = 100 + 55 * np.sin(xx * fx + rs.uniform(0, 6)) * np.cos(yy * fy + rs.uniform(0, 6))
            im = field[..., None] * cast + rs.randint(0, 48, (H, W, 3)).astype(np.float32)
            imgs[p] = np.clip(im, 0, 255).astype(np.uint8)
            if dataset == 'optic':
                cy, cx = H * rs.uniform(0.35, 0.65), W * rs.uniform(0.35, 0.65)
                rr = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
                m = np.full((H, W), 255, np.uint8)
                m[rr < 0.30 * H] = 128
                m[rr < 0.15 * H] = 0
            else:
                m = np.zeros((H, W), np.uint8)
                for _ in range(12):
                    x0, y0 = rs.uniform(0, W), rs.uniform(0, H)
                    ang = rs.uniform(0, np.pi)
                    dist = np.abs((xx - x0) * np.sin(ang) - (yy - y0) * np.cos(ang))
                    m[dist < 1.2] = 255
            msks[p] = m
    return imgs, msks


class SyntheticDGSegmentation(object):
    def __init__(self, n_domains=3, per_domain=32, size=512, dataset='optic', phase='train', transform=None,
                 seed=1023, device=None, length=None):
        self.phase = phase
        self.transforms = transform
        self.n_domains, self.per_domain = n_domains, per_domain
        imgs, msks = make_pool(seed, n_domains, per_domain, size, size, dataset)
        device = device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
        self.pool = DevicePool(torch.from_numpy(imgs).to(device), torch.from_numpy(msks).to(device))
        self.length = length if length is not None else per_domain

    def __len__(self):
        return self.length

    def _sample(self, d, index):
        p = d * self.per_domain + index
        s = {'image': self.pool.image(p), 'label': self.pool.mask(p), 'img_name': 'synth_d%d_%04d' % (d, index), 'dc': d}
        return self.transforms(s) if self.transforms is not None else s

    def __getitem__(self, index):
        if self.phase != 'test':
            # one random image per source domain (data/optic.py:82-90)
            return [self._sample(d, int(np.random.choice(self.per_domain, 1)[0])) for d in range(self.n_domains)]
        return self._sample(index // self.per_domain % self.n_domains, index % self.per_domain)
