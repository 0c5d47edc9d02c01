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
_NAMED_STEP_CACHE = {}
_UNIT_WORDS = _lib.UNIT_DTYPE.itemsize // 4                   # a unit record as int32 words: src, n_ops, op[K], iarg[K], farg[K], rect[K][4], geometry[5]


def _named_step(name, level, size, pool):
    """The op record of one (op name, level) step as Policy.__call__ would leave it on an ImageRef, as int32 words (op, iarg, bits of
    the float32 farg) -- ('cutout', v_abs) for Cutout (position dependent), None for a Cutout of size 0 -- cached by (name, level,
    image size): the controller's (op, magnitude) pairs come from a small discrete set."""
    key = (name, level, size)
    st = _NAMED_STEP_CACHE.get(key, key)
    if st is key:
        from . import basic
        if name == 'CutMix':
            raise KeyError(name)                               # not in augment_dict in the reference either (data/basic.py:253)
        fn, low, high = basic.get_augment(name)
        value = level * (high - low) + low                     # as apply_augment computes it (data/basic.py:258-260)
        if fn is basic.Cutout:
            assert 0.0 <= value <= 0.2
            st = ('cutout', value * size[0]) if value > 0. else None
        else:
            probe, _ = fn(ImageRef(pool, 0), MaskRef(pool, 0), value)     # same argument conversions and asserts as the object path
            o, i, f, r = probe.ops[-1]
            assert tuple(r) == (0, 0, -1, -1)
            st = (int(o), int(i), int(np.array([f], np.float32).view(np.int32)[0]))
        if len(_NAMED_STEP_CACHE) < 8192:
            _NAMED_STEP_CACHE[key] = st
    return st


def _policy_tables(mp, pool):
    """Per DGMultiPolicy: (sub-policies per policy, record templates int32 [M, nsub, 35] -- the unit record of every (policy,
    sub-policy) pair with src and geometry left open --, per pair the Cutout steps as (slot, v_abs) (position dependent: their
    boxes are drawn per unit), any Cutout at all)."""
    tabs = getattr(mp, '_fast_tables', None)
    if tabs is not None:
        return tabs
    from array import array
    K = _lib.MAX_OPS
    size = pool.size
    nsub = [len(p.policy) for p in mp.policies]
    ns = max(nsub)
    words, cut = array('i'), []
    any_cut = False
    tail = [0, 0, -1, -1] * K + [0] * 5                          # no Cutout box; geometry left open
    empty = [0, 0] + [0] * (3 * K) + tail
    for p in mp.policies:
        c_j = []
        for sub in p.policy:
            o, i, fb, cs, k = [0] * K, [0] * K, [0] * K, (), 0
            for name, level in sub:
                st = _named_step(name, level, size, pool)
                if st is None:
                    continue
                if k >= K:
                    raise RuntimeError("more than %d ops per sub-policy are not supported" % K)
                if st[0] == 'cutout':
                    o[k] = 9
                    cs += ((k, st[1]),)
                else:
                    o[k], i[k], fb[k] = st
                k += 1
            words.extend([0, k] + o + i + fb + tail)
            c_j.append(cs)
            any_cut = any_cut or bool(cs)
        for _ in range(ns - len(p.policy)):
            words.extend(empty)
            c_j.append(())
        cut.append(c_j)
    templ = np.frombuffer(words, dtype=np.int32).reshape(len(cut), ns, _UNIT_WORDS)
    tabs = mp._fast_tables = (tuple(nsub), templ, cut, any_cut)
    return tabs


def _draw_python_stream_py(n_items, D, M, nsub, queue_lens, sc, n_code, W0, H0):
    """(The Python statement of aadg_draw_python_stream -- csrc/host_draw.hip runs the same loop on a copy of the generator's state; tests compare
    the two draw for draw.)  Phase A of a training batch: every draw the pipeline makes from python's `random` generator, in the pipeline's order --
    per (item, domain): per policy the CutMix-queue draw and the sub-policy draw (data/policy.py:17-23), then DGRandomScaleCrop for
    the original and each augmented image (data/transform.py:104-131), then ToTensor's soft domain code (:260-274).  None of them
    depends on what the policies CONTAIN (only on how many sub-policies each has and on the length of its CutMix queue), so this
    phase can run before the controller's sample has reached the host (DeviceBatchLoader.predraw).  numpy's generator -- the image
    index and Cutout's box -- is phase B (fast_train_units): its consumption follows the chosen sub-policies."""
    # python's generator without a frame per draw: uniform(a, b) = a + (b - a) * random(), randint(a, b) = a + _randbelow(b - a + 1),
    # choice(seq) = seq[_randbelow(len(seq))], and _randbelow(n) itself (CPython random.py, _randbelow_with_getrandbits):
    #     k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
    # written out at its call sites -- the same stream, position by position (tests/test_host_cpu.py compares the records and the
    # generators' end states with the object path)
    rnd, getbits = random.random, random._inst.getrandbits
    s0, s1 = sc.scale_range[0], sc.scale_range[1]
    crop_h, crop_w = sc.crop.size
    crop_pad = sc.crop.padding
    ds = s1 - s0
    S = n_items * D
    n = S + S * M
    geo = [0] * (5 * n)
    R = [0] * (S * M)
    dcs = []
    qlen = list(queue_lens)
    ksub = [m.bit_length() for m in nsub]
    last_code = n_code - 1
    s = 0
    for _ in range(n_items):
        for d in range(D):
            base = S + s * M
            for j in range(M):
                nq = qlen[j] + 1                                    # q.append(sample)
                if nq > 10:
                    nq = 10                                         # q.pop(0): no draw
                else:                                               # the CutMix-queue draw (its value is unused: data/policy.py:17-21)
                    kq = nq.bit_length()
                    r = getbits(kq)
                    while r >= nq:
                        r = getbits(kq)
                qlen[j] = nq
                m, kq = nsub[j], ksub[j]                            # the sub-policy draw
                r = getbits(kq)
                while r >= m:
                    r = getbits(kq)
                R[s * M + j] = r
            # -- DGRandomScaleCrop: the original first, then every augmented image (scale draw, then crop draw)
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
                    m2 = ph - crop_h + 1
                    if m <= 0 or m2 <= 0:
                        raise ValueError("empty range for randrange() (0, %d, %d)" % (min(m, m2), min(m, m2)))   # what random.randint raises
                    kq = m.bit_length()
                    x1 = getbits(kq)
                    while x1 >= m:
                        x1 = getbits(kq)
                    kq = m2.bit_length()
                    y1 = getbits(kq)
                    while y1 >= m2:
                        y1 = getbits(kq)
                g5 = 5 * row
                geo[g5], geo[g5 + 1], geo[g5 + 2], geo[g5 + 3], geo[g5 + 4] = w, h, pad, x1, y1
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
    from array import array
    return {'n_items': n_items, 'D': D, 'M': M, 'nsub': tuple(nsub), 'queue_before': tuple(queue_lens), 'queue_after': tuple(qlen),
            'R': R, 'R_np': np.frombuffer(array('q', R), dtype=np.int64), 'geo': np.frombuffer(array('i', geo), dtype=np.int32).reshape(n, 5),
            'dcs': dcs}


USE_C_DRAW = True          # phase A through the library's host-side planner (csrc/host_draw.hip); False: the Python statement above


def _draw_python_stream(n_items, D, M, nsub, queue_lens, sc, n_code, W0, H0):
    """Phase A of a training batch (see _draw_python_stream_py) by the library's host planner: python's Mersenne-Twister state is copied
    out (random.getstate), advanced by aadg_draw_python_stream draw for draw as the interpreter would, and put back (random.setstate) --
    ~60 us instead of ~230 us of interpreter time per 24-sample batch.  Same record, same generator end state."""
    if not USE_C_DRAW:
        return _draw_python_stream_py(n_items, D, M, nsub, queue_lens, sc, n_code, W0, H0)
    lib = _lib.load()
    version, words, gauss = random.getstate()
    if version != 3 or len(words) != 625:
        return _draw_python_stream_py(n_items, D, M, nsub, queue_lens, sc, n_code, W0, H0)     # another interpreter's generator
    mt = np.array(words, dtype=np.uint32)
    S = n_items * D
    n = S + S * M
    nsub_a = np.asarray(nsub, dtype=np.int32)
    qlen = np.array(queue_lens, dtype=np.int32)
    sub = np.empty(S * M, dtype=np.int64)
    geo = np.empty((n, 5), dtype=np.int32)
    codes = np.empty((S, n_code), dtype=np.float64)
    crop_h, crop_w = sc.crop.size
    rc = lib.aadg_draw_python_stream(mt.ctypes.data, n_items, D, M, nsub_a.ctypes.data, qlen.ctypes.data, float(sc.scale_range[0]),
                                     float(sc.scale_range[1]), int(crop_h), int(crop_w), int(sc.crop.padding), int(n_code), int(W0), int(H0),
                                     sub.ctypes.data, geo.ctypes.data, codes.ctypes.data)
    if rc != 0:
        # an empty crop range (python's randint raises ValueError) or sizes the planner refuses: the Python statement raises / handles it,
        # from the UNTOUCHED generator state
        return _draw_python_stream_py(n_items, D, M, nsub, queue_lens, sc, n_code, W0, H0)
    random.setstate((version, tuple(mt.tolist()), gauss))
    return {'n_items': n_items, 'D': D, 'M': M, 'nsub': tuple(nsub), 'queue_before': tuple(queue_lens), 'queue_after': tuple(qlen.tolist()),
            'R': sub, 'R_np': sub, 'geo': geo, 'dcs': codes}


def _standard_pipeline(dataset):
    """(multi-policy, scale-crop, normalize, to-tensor) of the STANDARD training pipeline, or None"""
    from .policy import DGMultiPolicy
    from .synthetic import SyntheticDGSegmentation
    if type(dataset) is not SyntheticDGSegmentation:       # the draws mirror synthetic.py:__getitem__ (pool, n_domains, per_domain)
        return None
    tfs = getattr(getattr(dataset, 'transforms', None), 'transforms', None)
    if (tfs is None or len(tfs) != 4 or type(tfs[0]) is not DGMultiPolicy or type(tfs[1]) is not DGRandomScaleCrop or
            type(tfs[2]) is not Normalize_dg or type(tfs[3]) is not ToTensor or getattr(dataset, 'phase', 'train') == 'test'):
        return None
    if random._inst._randbelow.__func__ is not random.Random._randbelow_with_getrandbits:
        return None                                          # another interpreter's random.py: take the object path
    return tfs


def predraw_train_batch(dataset, n_items, fresh_policies=True):
    """Phase A of the NEXT training batch, drawn ahead of time (python's generator only: see _draw_python_stream).  `fresh_policies`:
    the batch will be drawn through NEWLY injected policies (search_dg.py:341 installs a new DGMultiPolicy per epoch: empty CutMix
    queues); False: through the objects installed now.  fast_train_units checks the assumption when it consumes the draw; a draw
    that does not fit is taken back (the generator's state is restored) and redone in place.  Drawing ahead leaves a seeded run's
    results unchanged ONLY IF nothing else draws from python's `random` between this call and the consuming batch (the test pipeline
    of validate() does: DGRandomCrop, SoftLable) -- callers with such a draw in between must not predraw (search_seg_dg_policy does
    not; bench.py's loops and the batches inside one epoch may).
    Returns False (and draws nothing) when the pipeline is not the standard one."""
    tfs = _standard_pipeline(dataset)
    if tfs is None or getattr(dataset, '_predrawn', None) is not None:
        return False
    mp, sc, nz, tt = tfs
    nsub = _policy_tables(mp, dataset.pool)[0]
    qlens = [0] * len(mp.policies) if fresh_policies else [len(p.queue) for p in mp.policies]
    W0, H0 = dataset.pool.size
    before = random.getstate()                                 # to take the draw back if it turns out not to fit (fast_train_units)
    dataset._predrawn = _draw_python_stream(n_items, dataset.n_domains, len(mp.policies), nsub, qlens, sc, tt.n, W0, H0)
    dataset._predrawn['generator_before'] = before
    dataset._predrawn['generator_after'] = random.getstate()
    return True


def fast_train_units(dataset, n_items):
    """Draws `n_items` training items (one image per source domain each) of the STANDARD pipeline
    [DGMultiPolicy, DGRandomScaleCrop, Normalize_dg, ToTensor] and returns the packed unit records directly:

        (units UNIT_DTYPE [S + S*M] in output-row order, dc float32 [S*M, n], dc_single float32 [S, n], names, M, kind)

    with S = n_items * n_domains.  Every random draw is made by the same function, with the same arguments and -- per generator --
    in the same order as the object path (synthetic.py:__getitem__ -> Policy.__call__ -> DGRandomScaleCrop -> ToTensor), so a
    seeded run produces identical records (tests/test_host_cpu.py::test_fast_draw_equals_object_path); what is skipped is the ~70
    ImageRef objects, copies and property calls per item.  Two phases (round 4): A = python's generator (policy-content
    independent, may have been drawn ahead by predraw_train_batch), B = numpy's generator (image index, Cutout boxes) + the
    records, assembled from per-policy tables with array indexing.  Returns None when the pipeline is not the standard one."""
    tfs = _standard_pipeline(dataset)
    if tfs is None:
        dataset._predrawn = None                                # a draw made ahead for a pipeline that has since become non-standard
        return None
    mp, sc, nz, tt = tfs
    pool = dataset.pool
    W0, H0 = pool.size
    D, per = dataset.n_domains, dataset.per_domain
    policies = mp.policies
    M = len(policies)
    nsub, templ, cut, any_cut = _policy_tables(mp, pool)
    qlens = tuple(len(p.queue) for p in policies)
    A = getattr(dataset, '_predrawn', None)
    dataset._predrawn = None
    if A is not None and (A['n_items'], A['D'], A['M'], A['nsub'], A['queue_before']) != (n_items, D, M, nsub, qlens):
        # drawn for another pipeline state than the one consuming it (e.g. the caller kept the policies instead of injecting new ones:
        # other CutMix-queue lengths, hence another number of draws): take the draw back -- python's generator returns to where it
        # stood before predraw_train_batch -- and draw in place, as if nothing had been drawn ahead.  Only if nobody else has drawn
        # since: rewinding would hand a third party's numbers out a second time; then the stream is kept and the batch drawn in place
        if random.getstate() == A['generator_after']:
            random.setstate(A['generator_before'])
        A = None
    if A is None:
        A = _draw_python_stream(n_items, D, M, nsub, qlens, sc, tt.n, W0, H0)
    for p, nq in zip(policies, A['queue_after']):              # the CutMix queues' state after the batch (their content is never read)
        del p.queue[:]
        p.queue.extend([None] * nq)
    S = n_items * D
    n = S + S * M
    # ---- phase B: numpy's legacy generator.  choice(n, 1)[0] draws randint(0, n) (mtrand: `idx = self.randint(0, pop_size, size=size)`);
    # the scalar call consumes the same words of the stream without building two arrays; uniform(low) = low + (1.0 - low) * random_sample()
    np_randint, np_sample = np.random.randint, np.random.random_sample
    wlow, hlow = 1.0 - W0, 1.0 - H0
    R = A['R']
    pidx = [0] * S
    names = []
    rects = []                                                  # (row, first rect word, box) of the Cutout steps
    cut_j = [j for j in range(M) if any(cut[j])] if any_cut else []
    s = 0
    for _ in range(n_items):
        for d in range(D):
            index = int(np_randint(0, per))                     # synthetic.py: np.random.choice(per, 1)[0], one random image per source domain
            pidx[s] = d * per + index
            names.append('synth_d%d_%04d' % (d, index))
            for j in cut_j:
                for k, v in cut[j][R[s * M + j]]:
                    # CutoutAbs (data/basic.py:153-163): np.random.uniform(w), np.random.uniform(h), then cutout_rect
                    x0 = int(max(0, W0 + wlow * np_sample() - v / 2.))
                    y0 = int(max(0, H0 + hlow * np_sample() - v / 2.))
                    rects.append((S + s * M + j, 2 + 3 * _lib.MAX_OPS + 4 * k,
                                  (x0, y0, min(int(min(W0, x0 + v)), W0 - 1), min(int(min(H0, y0 + v)), H0 - 1))))
            s += 1
    # the records as int32 words: the (policy, sub-policy) templates picked by the draws, then source index, geometry and Cutout boxes
    U = np.empty((n, _UNIT_WORDS), np.int32)
    U[:S] = templ[0, 0]
    U[:S, 1:2 + 3 * _lib.MAX_OPS] = 0                           # the un-augmented images: no ops
    sel = _plan_index(S, M)
    U[S:] = templ[sel, A['R_np']]
    pidx = np.asarray(pidx, dtype=np.int32)
    U[:S, 0] = pidx
    U[S:, 0] = np.repeat(pidx, M)
    U[:, _UNIT_WORDS - 5:] = A['geo']
    for row, w0, box in rects:
        U[row, w0:w0 + 4] = box
    units = U.reshape(-1).view(_lib.UNIT_DTYPE)
    dc_single = np.array(A['dcs'], dtype=np.float64).astype(np.float32)
    return units, np.repeat(dc_single, M, axis=0), dc_single, names, M, nz.dataset_name


_PLAN_INDEX = {}


def _plan_index(S, M):
    """policy index of every augmented row: 0 .. M-1 repeated S times (cached)"""
    v = _PLAN_INDEX.get((S, M))
    if v is None:
        v = _PLAN_INDEX[(S, M)] = np.tile(np.arange(M, dtype=np.intp), S)
    return v


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
