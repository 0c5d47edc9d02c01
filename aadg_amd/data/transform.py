"""Scale / crop / normalise / tensorise / collate -- host side; API mirror of data/transform.py.

Reference classes kept by name and constructor signature: RandomCrop (data/transform.py:27-55),
DGRandomCrop (:58-69), DGRandomScaleCrop (:97-135), Normalize_dg (:138-186), ToTensor (:208-236),
Identity (:239-241), to_multilabel/ToMultiLabel/SoftLable (:244-274), get_dg_segtransform
(:281-309), train_dg_collate_fn / test_dg_collate_fn (:323-362).

In this implementation the transforms operate on `ImageRef`s: each one makes the reference's random
draws in the reference's order (python `random`, as there) and records the geometry; the pixels of
the WHOLE batch -- ops, Pillow-exact BILINEAR/NEAREST resize, pad, crop, /127.5-1, mask->multilabel,
HWC->CHW -- are produced by one GPU call in the collate function (the kernels replace what
Normalize_dg/ToTensor/collate spent 90 % of the reference's CPU pipeline time on).
"""
import numbers
import random

import numpy as np
import torch

from .. import _lib
from .basic import ImageRef, MaskRef


class Compose(object):
    """torchvision.transforms.Compose stand-in: `.transforms` is the list the search driver patches
    (train_loader.dataset.transforms.transforms[0] = DGMultiPolicy(...), search_dg.py:341)."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


class Identity(object):
    def __call__(self, sample):
        return sample


def _square(size):
    if isinstance(size, numbers.Number):
        return (int(size), int(size))
    return tuple(size)


class RandomCrop(object):
    """Pads (fill 0) when the image is smaller than the target, then crops at a random offset."""

    def __init__(self, size, padding=0):
        self.size = _square(size)  # (h, w)
        self.padding = padding

    def draw(self, w, h):
        """Returns (pad, x1, y1) for an image of PIL size (w, h); draws like data/transform.py:50-51."""
        pad = 0
        if self.padding > 0 or w < self.size[0] or h < self.size[1]:
            pad = int(max(self.padding, (self.size[0] - w) // 2 + 5, (self.size[1] - h) // 2 + 5))
            w, h = w + 2 * pad, h + 2 * pad
        th, tw = self.size
        if w == tw and h == th:
            return pad, 0, 0
        x1 = random.randint(0, w - tw)
        y1 = random.randint(0, h - th)
        return pad, x1, y1

    def apply_(self, img, mask):
        """In-place on refs the caller owns (same draws as __call__)."""
        w, h = img.size
        assert (w, h) == mask.size
        pad, x1, y1 = self.draw(w, h)
        for r in (img, mask):
            if r.scaled is None:
                r.scaled = r.pool.size
            r.pad, r.crop, r.crop_size = pad, (x1, y1), self.size
        return img, mask

    def __call__(self, img, mask):
        return self.apply_(img.copy(), mask.copy())


class DGRandomCrop(object):
    def __init__(self, size, padding=0):
        self.size = _square(size)
        self.padding = padding
        self.crop = RandomCrop(size, padding)

    def __call__(self, sample):
        sample['image'], sample['label'] = self.crop(sample['image'], sample['label'])
        return sample


class DGRandomScaleCrop(object):
    """p=0.8: resize to int(U(s0,s1)*w) x int(U(s0,s1)*h) (BILINEAR image / NEAREST mask), then
    RandomCrop.  Every augmented image gets its own scale and crop draw and is paired with a
    re-scaled copy of the ORIGINAL mask (data/transform.py:122-131)."""

    def __init__(self, size, scale_range=[1, 1.5]):
        self.size = size
        self.scale_range = scale_range
        self.crop = RandomCrop(self.size)

    def scale_(self, img, mask):
        """In-place on refs the caller owns."""
        if random.random() > 0.2:
            sw, sh = img.size
            w = int(random.uniform(self.scale_range[0], self.scale_range[1]) * sw)
            h = int(random.uniform(self.scale_range[0], self.scale_range[1]) * sh)
            img.scaled = mask.scaled = (w, h)
        return img, mask

    def scale(self, img, mask):
        return self.scale_(img.copy(), mask.copy())

    def _scale_crop(self, img, mask):
        # one private copy of each ref, then both steps in place (draw order: scale, then crop -- as the reference)
        return self.crop.apply_(*self.scale_(img.copy(), mask.copy()))

    def __call__(self, sample):
        img, mask = sample['image'], sample['label']
        assert img.size == mask.size
        sample['image'], sample['label'] = self._scale_crop(img, mask)
        if 'aug_images' in sample:
            done = [self._scale_crop(aug, mask) for aug in sample['aug_images']]
            sample['aug_images'] = [d[0] for d in done]
            sample['aug_labels'] = [d[1] for d in done]
        return sample


class Normalize_dg(object):
    """img -> float32 x/127.5 - 1; mask -> multilabel (optic: [cup, disc]) or binary (vessel).
    Deferred: only tags the refs with the dataset kind; the arithmetic runs in the fused kernel."""

    def __init__(self, dataset_name, mean=(0., 0., 0.), std=(1., 1., 1.)):
        self.mean = mean
        self.std = std
        self.dataset_name = dataset_name

    def _tag(self, ref):
        # refs that went through a crop step are private copies of that step: tag them in place
        r = ref if ref.crop_size is not None else ref.copy()
        r.norm = self.dataset_name
        return r

    def __call__(self, sample):
        if 'aug_images' in sample:
            sample['aug_images'] = [self._tag(r) for r in sample['aug_images']]
            sample['aug_labels'] = [self._tag(r) for r in sample['aug_labels']]
        sample['image'] = self._tag(sample['image'])
        sample['label'] = self._tag(sample['label'])
        return sample


def to_multilabel(pre_mask, classes=2):
    mask = np.zeros((*pre_mask.shape, classes))
    mask[pre_mask == 1] = [0, 1]
    mask[pre_mask == 2] = [1, 1]
    return mask


def ToMultiLabel(dc, c):
    new_dc = np.zeros([c])
    new_dc[dc] = 1
    return new_dc


def SoftLable(label):
    """Soft one-hot: true class U[0.8,1], the rest random, summing to 1 (data/transform.py:260-274);
    draws from python `random` in the reference's order."""
    hard = list(label)
    hot = hard.index(1)
    soft = np.array(label, dtype=np.float64)
    soft[hot] = 0.8 + random.random() * 0.2
    used = soft[hot]
    last = len(hard) - 1
    for i in range(len(hard)):
        if i == hot:
            continue
        if i == last:
            soft[i] = 1 - used
        else:
            soft[i] = random.random() * (1 - used)
            used += soft[i]
    return soft


class ToTensor(object):
    """Draws the soft domain code; the HWC->CHW float conversion itself happens in the fused kernel."""

    def __init__(self, dataset_name, n_domains=None) -> None:
        super().__init__()
        # the reference hard-codes 3 source domains (data/transform.py:212-215); n_domains widens the code for configs with
        # more sources (BASELINE configs[4]: 8)
        self.n = 3 if dataset_name in ['optic', 'vessel'] else 2
        if n_domains is not None:
            self.n = max(self.n, int(n_domains))

    def __call__(self, sample):
        soft = SoftLable(ToMultiLabel(sample['dc'], self.n)).astype(np.float32)      # same rounding as .float()
        domain_code = torch.from_numpy(soft)
        sample['dc'] = domain_code
        if 'aug_images' in sample:
            sample['dc'] = torch.from_numpy(np.tile(soft, (len(sample['aug_images']), 1)))
            sample['dc_single'] = domain_code
        return sample


def get_dg_segtransform(dataset, size=256, n_domains=None):
    """Same pipelines as data/transform.py:281-309; `size` generalises the hard-coded 256 crop
    (BASELINE configs run 512 and 1024), `n_domains` the hard-coded 3-wide domain code."""
    if 'optic' in dataset:
        transform_train_imgs = Compose([
            Identity(),  # slot 0: the search driver installs DGMultiPolicy here
            DGRandomScaleCrop(size),
            Normalize_dg('optic'),
            ToTensor('optic', n_domains)
        ])
        transform_test_imgs = Compose([
            DGRandomCrop(size),
            Normalize_dg('optic'),
            ToTensor('optic', n_domains)
        ])
    elif 'rvs' in dataset:
        transform_train_imgs = Compose([
            Identity(),
            DGRandomScaleCrop(size, scale_range=[0.5, 2]),
            Normalize_dg('vessel'),
            ToTensor('vessel', n_domains)
        ])
        transform_test_imgs = Compose([
            Normalize_dg('vessel'),
            ToTensor('vessel', n_domains),
        ])
    else:
        raise NotImplementedError(dataset)
    return transform_train_imgs, transform_test_imgs


# ------------------------------------------------------------------------------------------------
# ImageRef -> aadg_unit records -> one GPU launch
# ------------------------------------------------------------------------------------------------
def refs_to_units(refs):
    """Pack recorded ImageRefs into the C-ABI unit array (include/aadg_hip.h: aadg_unit).  Plain Python lists are
    filled in one pass and assigned column-wise (per-record structured-array field writes cost ~3 us each)."""
    n = len(refs)
    K = _lib.MAX_OPS
    src, n_ops, geo = [0] * n, [0] * n, [None] * n
    op = [[0] * K for _ in range(n)]
    iarg = [[0] * K for _ in range(n)]
    farg = [[0.0] * K for _ in range(n)]
    rect = [[(0, 0, -1, -1)] * K for _ in range(n)]
    for i, r in enumerate(refs):
        src[i] = r.src
        ops = r.ops
        n_ops[i] = len(ops)
        if ops:
            oi, ii, fi, ri = op[i], iarg[i], farg[i], rect[i]
            for k, (o, ia, fa, rc) in enumerate(ops):
                oi[k], ii[k], fi[k], ri[k] = o, ia, fa, rc
        w, h = r.scaled if r.scaled is not None else r.pool.size
        geo[i] = (w, h, r.pad, r.crop[0], r.crop[1])
    units = np.zeros(n, dtype=_lib.UNIT_DTYPE)
    if n == 0:
        return units
    units['src'] = src
    units['n_ops'] = n_ops
    units['op'] = op
    units['iarg'] = iarg
    units['farg'] = farg
    units['rect'] = rect
    g = np.asarray(geo, dtype=np.int64)
    units['scaled_w'], units['scaled_h'], units['pad'], units['crop_x'], units['crop_y'] = g[:, 0], g[:, 1], g[:, 2], g[:, 3], g[:, 4]
    return units


def materialize(refs, out_img=None, out_lbl=None):
    """Run the fused GPU pipeline for a list of ImageRefs (same pool, same crop, same dataset kind)."""
    first = refs[0]
    pool = first.pool
    if first.crop_size is None:
        crop_hw = (pool.size[1], pool.size[0]) if first.scaled is None else (first.scaled[1], first.scaled[0])
    else:
        crop_hw = first.crop_size
    if crop_hw[0] != crop_hw[1]:
        raise NotImplementedError("non-square output crops are not supported")
    for r in refs:
        if r.pool is not pool or r.norm != first.norm:
            raise ValueError("all images of a batch must come from one DevicePool / one dataset kind")
        if r.crop_size is not None and tuple(r.crop_size) != tuple(crop_hw):
            raise ValueError("all images of a batch must share one crop size")
    if first.norm is None:
        raise ValueError("Normalize_dg must run before collation")
    kind = _lib.DATASET_OPTIC if first.norm == 'optic' else _lib.DATASET_VESSEL
    hist = pool.histograms()
    extra = {} if hist is None else {'pool_hist': hist}
    return _lib.aug_u8_forward(pool.images, pool.masks, refs_to_units(refs), int(crop_hw[0]), kind, out_img, out_lbl, **extra)


def collect_refs(batch, nested):
    """Flattens a batch and lists its ImageRefs in output-row order: the S un-augmented images first,
    then the augmented ones at row S + s*M + j (sample s, policy j)."""
    if nested:
        batch = [item for sublist in batch for item in sublist]
    refs = [b['image'] for b in batch]
    M = len(batch[0]['aug_images']) if 'aug_images' in batch[0] else 0
    if M:
        for b in batch:
            refs.extend(b['aug_images'])
    return batch, refs, M


_ROW_SHARD = (0, 1, 'unit', False)   # (rank, world, placement law, force): which collate rows this process materialises
_PLANS = {}


def set_row_shard(rank, world_size, law='unit', force=False):
    """Multi-GPU: every rank draws the SAME batch plan (same seeds -> same draws, no communication) but
    materialises only the rows of its (domain, policy) units (aadg_amd/distributed.py: RowPlan).  `force`: a one-rank job
    takes the sharded code path too (bench.py --force_dist)."""
    global _ROW_SHARD
    _ROW_SHARD = (int(rank), int(world_size), law, bool(force))


def row_plan(D, B, M):
    from ..distributed import RowPlan
    key = (D, B, M) + _ROW_SHARD
    plan = _PLANS.get(key)
    if plan is None:
        plan = _PLANS[key] = RowPlan(D, B, M, *_ROW_SHARD)
    return plan


def _collate(batch, nested):
    D = len(batch[0]) if nested else 1
    B = len(batch)
    batch, refs, M = collect_refs(batch, nested)
    S = len(batch)
    new_batch = {'img_name': [b['img_name'] for b in batch]}
    rank, world, _, force = _ROW_SHARD
    if M and (world > 1 or force):
        # training batch of a sharded job: this rank's slice of the un-augmented rows (warm-up epochs) and the rows of its
        # (domain, policy) units, in the plan's local order
        from ..distributed import shard_rows
        plan = row_plan(D, B, M)
        lo_s, hi_s = shard_rows(S, rank, world)
        local = [refs[S + int(r)] for r in plan.rows]
    else:
        # single process -- or a test batch, which every rank scores in full (validate() then agrees on all ranks)
        plan = row_plan(D, B, M) if M else None
        lo_s, hi_s = 0, S
        local = refs[S:]
    img, lbl = materialize(refs[lo_s:hi_s] + local)
    ns = hi_s - lo_s
    new_batch['image'], new_batch['label'] = img[:ns], lbl[:ns]
    dev = img.device
    if M:
        new_batch['aug_images'], new_batch['aug_labels'] = img[ns:], lbl[ns:]
        dc = torch.cat([b['dc'] for b in batch], dim=0)
        if plan.sharded:
            dc = dc[torch.from_numpy(plan.rows)]
        new_batch['dc'] = dc.to(dev, non_blocking=True)
        new_batch['dc_image'] = torch.stack([b['dc_single'] for b in batch], dim=0)[lo_s:hi_s].to(dev, non_blocking=True)
        new_batch['plan'] = plan
        new_batch['image_rows'] = (lo_s, hi_s, S)
    else:
        new_batch['dc'] = torch.stack([b['dc'] for b in batch], dim=0).to(dev, non_blocking=True)
    if 'roi' in batch[0]:
        new_batch['roi'] = torch.stack([b['roi'] for b in batch], dim=0)
    return new_batch


# ------------------------------------------------------------------------------------------------
# Fast draw path: the standard training pipeline without per-image objects
# ------------------------------------------------------------------------------------------------
def _fast_policy_steps(policy, pool):
    """Per sub-policy: the op records Policy.__call__ would leave on an ImageRef, captured by running every op function ONCE on a
    probe ref (same argument conversions and asserts as the object path); Cutout is position dependent and stays symbolic:
    ('cutout', v_abs) with v_abs = v * width, or nothing when v <= 0 (data/basic.py:137-146)."""
    from . import basic
    if policy._compiled is None:
        policy._compiled = policy._compile()
    out = []
    for steps in policy._compiled:
        rec = []
        for fn, value in steps:
            if fn is basic.Cutout:
                assert 0.0 <= value <= 0.2
                if value > 0.:
                    rec.append(('cutout', value * pool.size[0]))
            else:
                key = (fn, value, pool.size)       # a probe on an image of this size (Cutout and friends depend on it)
                st = _STEP_CACHE.get(key)
                if st is None:                 # (op, magnitude) pairs come from a small discrete set: probe each once
                    probe, _ = fn(ImageRef(pool, 0), MaskRef(pool, 0), value)
                    st = probe.ops[-1]
                    if len(_STEP_CACHE) < 4096:
                        _STEP_CACHE[key] = st
                rec.append(st)
        out.append(rec)
    return out


_STEP_CACHE = {}


def fast_train_units(dataset, n_items):
    """Draws `n_items` training items (one image per source domain each) of the STANDARD pipeline
    [DGMultiPolicy, DGRandomScaleCrop, Normalize_dg, ToTensor] and returns the packed unit records directly:

        (units UNIT_DTYPE [S + S*M] in output-row order, dc float32 [S*M, n], dc_single float32 [S, n], names, M, kind)

    with S = n_items * n_domains.  Every random draw is made by the same function, with the same arguments and in the same
    order as the object path (synthetic.py:__getitem__ -> Policy.__call__ -> DGRandomScaleCrop -> ToTensor), so a seeded run
    produces identical records (tests/test_host_cpu.py::test_fast_draw_equals_object_path); what is skipped is the ~70 ImageRef
    objects, copies and property calls per item.  Returns None when the dataset's pipeline is not the standard one."""
    from .basic import cutout_rect
    from .policy import DGMultiPolicy
    from .synthetic import SyntheticDGSegmentation
    if type(dataset) is not SyntheticDGSegmentation:       # the draws below mirror synthetic.py:__getitem__ (pool, n_domains, per_domain)
        return None
    tfs = getattr(getattr(dataset, 'transforms', None), 'transforms', None)
    if (tfs is None or len(tfs) != 4 or type(tfs[0]) is not DGMultiPolicy or type(tfs[1]) is not DGRandomScaleCrop or
            type(tfs[2]) is not Normalize_dg or type(tfs[3]) is not ToTensor or getattr(dataset, 'phase', 'train') == 'test'):
        return None
    mp, sc, nz, tt = tfs
    pool = dataset.pool
    W0, H0 = pool.size
    D, per = dataset.n_domains, dataset.per_domain
    policies = mp.policies
    M = len(policies)
    fast = [getattr(p, '_fast', None) or _fast_policy_steps(p, pool) for p in policies]
    for p, f in zip(policies, fast):
        p._fast = f
    s0, s1 = sc.scale_range[0], sc.scale_range[1]
    K = _lib.MAX_OPS
    # python's generator without a frame per draw: uniform(a, b) = a + (b - a) * random(), randint(a, b) = a + _randbelow(b - a + 1),
    # choice(seq) = seq[_randbelow(len(seq))], and _randbelow(n) itself (CPython random.py, _randbelow_with_getrandbits):
    #     k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
    # written out at its four call sites -- the same stream, position by position (tests/test_host_cpu.py compares the records and the
    # generators' end states with the object path)
    rnd, getbits = random.random, random._inst.getrandbits
    assert random._inst._randbelow.__func__ is random.Random._randbelow_with_getrandbits
    # numpy's legacy generator: choice(n, 1)[0] draws randint(0, n) (mtrand: `idx = self.randint(0, pop_size, size=size)`), the scalar call
    # consumes the same words of the stream without building two arrays; uniform(low) = low + (1.0 - low) * random_sample()
    np_randint, np_sample = np.random.randint, np.random.random_sample
    crop_h, crop_w = sc.crop.size
    crop_pad = sc.crop.padding
    ds = s1 - s0
    S = n_items * D
    n = S + S * M
    src, n_ops, geo = [0] * n, [0] * n, [None] * n
    op = [[0] * K for _ in range(n)]
    iarg = [[0] * K for _ in range(n)]
    farg = [[0.0] * K for _ in range(n)]
    rect = [[(0, 0, -1, -1)] * K for _ in range(n)]
    names, dcs = [], []
    queues = [p.queue for p in policies]
    nsub = [len(f) for f in fast]
    ksub = [m.bit_length() for m in nsub]
    n_code = tt.n
    last_code = n_code - 1
    wlow, hlow = 1.0 - W0, 1.0 - H0
    s = 0
    for _ in range(n_items):
        for d in range(D):
            index = int(np_randint(0, per))                         # synthetic.py: np.random.choice(per, 1)[0], one random image per source domain
            pidx = d * per + index
            names.append('synth_d%d_%04d' % (d, index))
            # -- DGMultiPolicy: per policy the CutMix-queue draw, the sub-policy draw, Cutout's two numpy draws
            base = S + s * M
            for j in range(M):
                q = queues[j]
                q.append(None)
                nq = len(q)
                if nq > 10:
                    q.pop(0)
                else:                                               # the CutMix-queue draw (its value is unused: data/policy.py:17-21)
                    kq = nq.bit_length()
                    r = getbits(kq)
                    while r >= nq:
                        r = getbits(kq)
                m, kq = nsub[j], ksub[j]                            # the sub-policy draw
                r = getbits(kq)
                while r >= m:
                    r = getbits(kq)
                row = base + j
                src[row] = pidx
                k = 0
                oi, ii, fi, ri = op[row], iarg[row], farg[row], rect[row]
                for st in fast[j][r]:
                    if st[0] == 'cutout':
                        # CutoutAbs (data/basic.py:153-163): np.random.uniform(w), np.random.uniform(h), then cutout_rect
                        v = st[1]
                        x0 = int(max(0, W0 + wlow * np_sample() - v / 2.))
                        y0 = int(max(0, H0 + hlow * np_sample() - v / 2.))
                        st = (9, 0, 0.0, (x0, y0, min(int(min(W0, x0 + v)), W0 - 1), min(int(min(H0, y0 + v)), H0 - 1)))
                    if k >= K:
                        raise RuntimeError("more than %d ops per sub-policy are not supported" % K)
                    oi[k], ii[k], fi[k], ri[k] = st
                    k += 1
                n_ops[row] = k
            # -- DGRandomScaleCrop: the original first, then every augmented image (scale draw, then crop draw)
            src[s] = pidx
            row = s
            for jj in range(M + 1):
                w, h = W0, H0
                if rnd() > 0.2:
                    w = int((s0 + ds * rnd()) * W0)
                    h = int((s0 + ds * rnd()) * H0)
                # RandomCrop.draw (data/transform.py:38-53), inlined
                pad = 0
                pw, ph = w, h
                if crop_pad > 0 or w < crop_h or h < crop_w:
                    pad = int(max(crop_pad, (crop_h - w) // 2 + 5, (crop_w - h) // 2 + 5))
                    pw, ph = w + 2 * pad, h + 2 * pad
                if pw == crop_w and ph == crop_h:
                    x1 = y1 = 0
                else:
                    m = pw - crop_w + 1                             # random.randint(0, pw - crop_w)
                    kq = m.bit_length()
                    x1 = getbits(kq)
                    while x1 >= m:
                        x1 = getbits(kq)
                    m = ph - crop_h + 1
                    kq = m.bit_length()
                    y1 = getbits(kq)
                    while y1 >= m:
                        y1 = getbits(kq)
                geo[row] = (w, h, pad, x1, y1)
                row = base + jj
            # -- ToTensor: the soft domain code (SoftLable(ToMultiLabel(d, n)): the true class U[0.8, 1], the others share the rest)
            code = [0.0] * n_code
            used = 0.8 + rnd() * 0.2
            code[d] = used
            for i in range(n_code):
                if i != d:
                    if i == last_code:
                        code[i] = 1 - used
                    else:
                        t = rnd() * (1 - used)
                        code[i] = t
                        used += t
            dcs.append(code)
            s += 1
    units = np.zeros(n, dtype=_lib.UNIT_DTYPE)
    units['src'] = src
    units['n_ops'] = n_ops
    units['op'] = op
    units['iarg'] = iarg
    units['farg'] = farg
    units['rect'] = rect
    g = np.asarray(geo, dtype=np.int64)
    units['scaled_w'], units['scaled_h'], units['pad'], units['crop_x'], units['crop_y'] = g[:, 0], g[:, 1], g[:, 2], g[:, 3], g[:, 4]
    dc_single = np.array(dcs, dtype=np.float64).astype(np.float32)
    return units, np.repeat(dc_single, M, axis=0), dc_single, names, M, nz.dataset_name


def fast_train_collate(dataset, n_items):
    """train_dg_collate_fn([dataset[i] for i in ...]) through fast_train_units: same dictionary, same tensors (None when the
    pipeline is not the standard one: the caller then takes the object path)."""
    drawn = fast_train_units(dataset, n_items)
    if drawn is None:
        return None
    units, dc, dc_single, names, M, kind_name = drawn
    D = dataset.n_domains
    S = n_items * D
    pool = dataset.pool
    crop = int(dataset.transforms.transforms[1].crop.size[0])
    rank, world, _, force = _ROW_SHARD
    plan = row_plan(D, n_items, M)
    if world > 1 or force:
        from ..distributed import shard_rows
        lo_s, hi_s = shard_rows(S, rank, world)
        sel = np.concatenate([np.arange(lo_s, hi_s), S + plan.rows])
        units = units[sel]
        dc = dc[plan.rows]
    else:
        lo_s, hi_s = 0, S
    kind = _lib.DATASET_OPTIC if kind_name == 'optic' else _lib.DATASET_VESSEL
    hist = pool.histograms()
    img, lbl = _lib.aug_u8_forward(pool.images, pool.masks, units, crop, kind, **({} if hist is None else {'pool_hist': hist}))
    ns = hi_s - lo_s
    dev = img.device
    return {'img_name': names, 'image': img[:ns], 'label': lbl[:ns], 'aug_images': img[ns:], 'aug_labels': lbl[ns:],
            'dc': torch.from_numpy(dc).to(dev, non_blocking=True),
            'dc_image': torch.from_numpy(dc_single[lo_s:hi_s]).to(dev, non_blocking=True), 'plan': plan, 'image_rows': (lo_s, hi_s, S)}


def train_dg_collate_fn(batch):
    """batch = [[sample per domain] per item]; rows come out item-major, domain-minor, and the
    augmented tensors in `sample*M + policy` order, as data/transform.py:323-340."""
    return _collate(batch, nested=True)


def test_dg_collate_fn(batch):
    return _collate(batch, nested=False)
