"""Loader factory -- API mirror of the reference's data/dataloader.py:10-36
(`get_seg_dg_dataloader(cfg, args, batch_size, workers) -> (train_sampler, train_loader, test_loader)`).

There are no DataLoader worker processes: the per-sample host work is only the random draws (microseconds);
pixels are produced by the fused GPU call inside the collate function.  `loader.dataset.transforms.transforms[0]`
stays the policy injection point (search_dg.py:341).
"""
import numpy as np

from .synthetic import SyntheticDGSegmentation
from .transform import get_dg_segtransform, test_dg_collate_fn, train_dg_collate_fn


class DeviceBatchLoader(object):
    def __init__(self, dataset, batch_size, collate_fn, shuffle=False, drop_last=False):
        self.dataset, self.batch_size, self.collate_fn = dataset, batch_size, collate_fn
        self.shuffle, self.drop_last = shuffle, drop_last
        self.sampler = None

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def predraw(self, fresh_policies=True):
        """Draw the policy-independent part of the NEXT training batch now (python's generator: sub-policy choices, scale / crop
        geometry, soft domain codes -- data/transform.py: predraw_train_batch), e.g. while the GPU is still busy with the previous
        step and the controller's next sample has not reached the host yet.  The batch itself is completed by the next __iter__
        step.  `fresh_policies`: a new DGMultiPolicy will have been injected by then (search_dg.py:341).  No-op (False) for
        pipelines other than the standard training one."""
        if self.collate_fn is train_dg_collate_fn and getattr(self.dataset, 'phase', None) == 'train' and self.drop_last:
            from .transform import predraw_train_batch
            return predraw_train_batch(self.dataset, self.batch_size, fresh_policies)
        return False

    def __iter__(self):
        n = len(self.dataset)
        order = np.random.permutation(n) if self.shuffle else np.arange(n)
        fast = None
        if self.collate_fn is train_dg_collate_fn and getattr(self.dataset, 'phase', None) == 'train':
            from .transform import fast_train_collate as fast       # standard pipeline: same draws, no per-image objects
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            batch = fast(self.dataset, len(idx)) if fast is not None else None
            if batch is None:
                batch = self.collate_fn([self.dataset[int(i)] for i in idx])
            yield batch


def get_seg_dg_dataloader(cfg, args, batch_size, workers, size=None, per_domain=32, device=None):
    name = cfg.DATASET.NAME
    size = size if size is not None else getattr(args, 'crop_size', 256)
    kind = 'optic' if name == 'optic' else 'rvs'
    n_domains = len(cfg.DATASET.DG.TRAIN)
    transform_train, transform_test = get_dg_segtransform(name, size, n_domains)
    src = size if kind == 'optic' else 2 * size if size <= 512 else size
    trainset = SyntheticDGSegmentation(n_domains, per_domain, src if kind == 'rvs' else size, kind, 'train', transform_train,
                                       seed=cfg.SEED or 1023, device=device, length=getattr(args, 'epoch_items', per_domain))
    testset = SyntheticDGSegmentation(1, max(batch_size, 8), size, kind, 'test', transform_test, seed=(cfg.SEED or 1023) + 1,
                                      device=device, length=max(batch_size, 8))
    train_loader = DeviceBatchLoader(trainset, batch_size, train_dg_collate_fn, shuffle=True, drop_last=True)
    test_loader = DeviceBatchLoader(testset, batch_size, test_dg_collate_fn, shuffle=False, drop_last=False)
    return None, train_loader, test_loader
