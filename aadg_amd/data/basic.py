"""Op registry of the live augmentation path -- host mirror of the reference's data/basic.py.

Same public surface (augment_list / augment_dict / get_augment / apply_augment, the op functions
with the reference's names, ranges and asserts; data/basic.py:70-120,137-167,231-260), but the
ops do not touch pixels on the host.  An op function accepts either

  * an `ImageRef` (deferred mode, used by the batched pipeline): the op and every random draw it
    makes are *recorded*; pixels are produced later by ONE fused GPU call for the whole batch
    (aadg_amd/_lib/aug.py: aug_u8_forward), or
  * a uint8 HWC torch tensor on the GPU (eager mode): the op runs immediately through the C ABI
    (aadg_op_u8).

Random draws are made on the host in exactly the reference's order (np.random.uniform twice in
CutoutAbs, data/basic.py:150-151), so a run seeded like the reference records identical parameters.
"""
import math
import random  # noqa: F401  (kept: same module-level RNG as the reference)

import numpy as np
import torch

from .. import _lib

# op ids = index in augment_list() (include/aadg_hip.h enum aadg_op)
OP_AUTOCONTRAST, OP_INVERT, OP_EQUALIZE, OP_SOLARIZE, OP_POSTERIZE, OP_CONTRAST, OP_COLOR, \
    OP_BRIGHTNESS, OP_SHARPNESS, OP_CUTOUT = range(10)


class DevicePool(object):
    """Source images resident in HBM: images uint8 [P,H,W,3], masks uint8 [P,H,W]."""
    cache_statistics = True      # A/B switch (bench.py --no_pool_stats)

    def __init__(self, images, masks):
        assert images.dtype == torch.uint8 and masks.dtype == torch.uint8
        assert images.dim() == 4 and images.shape[3] == 3 and tuple(masks.shape) == tuple(images.shape[:3])
        self.images = images.contiguous()
        self.masks = masks.contiguous()
        self._hist = None
        self._hist_version = -1

    def histograms(self):
        """Per-image statistics of the resident pool (_lib.pool_histograms), computed on first use: the policy ops see the raw
        source image, so AutoContrast / Equalize / Contrast statistics are a property of the pool image, not of the batch.
        Recomputed when `images` was written to in place (tensor version counter); invalidate() forces it."""
        if not DevicePool.cache_statistics:
            return None
        if not self.images.is_cuda:
            return None
        # in-place writes to the pool bump its version; rebinding `images` to another tensor changes its address / shape
        key = (self.images._version, self.images.data_ptr(), tuple(self.images.shape))
        if self._hist is None or self._hist_version != key:
            from .. import _lib
            self._hist = _lib.pool_histograms(self.images)
            self._hist_version = key
        return self._hist

    def invalidate(self):
        self._hist = None

    @property
    def size(self):  # PIL convention (w, h)
        return (self.images.shape[2], self.images.shape[1])

    def image(self, index):
        return ImageRef(self, int(index))

    def mask(self, index):
        return MaskRef(self, int(index))


_NO_RECT = (0, 0, -1, -1)
_F32 = {}


class ImageRef(object):
    """A not-yet-materialised RGB image: pool index + recorded ops + recorded geometry."""
    __slots__ = ("pool", "src", "ops", "scaled", "pad", "crop", "crop_size", "norm")

    def __init__(self, pool, src, ops=(), scaled=None, pad=0, crop=(0, 0), crop_size=None, norm=None):
        self.pool, self.src, self.ops = pool, src, tuple(ops)
        self.scaled, self.pad, self.crop, self.crop_size, self.norm = scaled, pad, crop, crop_size, norm

    def copy(self):
        return ImageRef(self.pool, self.src, self.ops, self.scaled, self.pad, self.crop, self.crop_size, self.norm)

    @property
    def size(self):
        """PIL-style (w, h) of the image as it currently stands."""
        if self.crop_size is not None:
            return (self.crop_size[1], self.crop_size[0])
        w, h = self.scaled if self.scaled is not None else self.pool.size
        return (w + 2 * self.pad, h + 2 * self.pad)

    width = property(lambda self: self.size[0])
    height = property(lambda self: self.size[1])

    def with_op(self, op, iarg=0, farg=0.0, rect=_NO_RECT):
        if self.scaled is not None or self.crop_size is not None:
            raise RuntimeError("policy ops must precede scale/crop (Compose slot 0, data/transform.py:284)")
        if len(self.ops) >= _lib.MAX_OPS:
            raise RuntimeError("more than %d ops per sub-policy are not supported" % _lib.MAX_OPS)
        f32 = _F32.get(farg)
        if f32 is None:                       # magnitudes come from a small discrete set: round to float32 once each
            f32 = float(np.float32(farg))
            if len(_F32) < 4096:
                _F32[farg] = f32
        if rect is not _NO_RECT:
            rect = tuple(int(v) for v in rect)
        r = self.copy()
        r.ops = self.ops + ((int(op), int(iarg), f32, rect),)
        return r


class MaskRef(ImageRef):
    """Label image (mode L).  Policy ops leave it untouched except Cutout, and the reference then
    discards the policy-modified mask anyway (data/transform.py:127-131), so ops are not recorded."""
    __slots__ = ()

    def copy(self):
        return MaskRef(self.pool, self.src, self.ops, self.scaled, self.pad, self.crop, self.crop_size, self.norm)


def _is_ref(img):
    return isinstance(img, ImageRef)


def _size(img):
    if _is_ref(img):
        return img.size
    return (img.shape[1], img.shape[0])


def _run(img, op, iarg=0, farg=0.0, rect=_NO_RECT):
    if _is_ref(img):
        return img.with_op(op, iarg, farg, rect)
    if torch.is_tensor(img):
        return _lib.op_u8(img, op, iarg, float(np.float32(farg)), rect)
    raise TypeError("expected an ImageRef or a uint8 HWC GPU tensor, got %r" % type(img))


def AutoContrast(img, mask, _):
    return _run(img, OP_AUTOCONTRAST), mask


def Invert(img, mask, _):
    return _run(img, OP_INVERT), mask


def Equalize(img, mask, _):
    return _run(img, OP_EQUALIZE), mask


def Solarize(img, mask, v):  # [0, 256]
    assert 0 <= v <= 256
    # PIL: i if i < threshold else 255 - i, with a float threshold  <=>  i < ceil(threshold)
    return _run(img, OP_SOLARIZE, iarg=int(math.ceil(v))), mask


def Posterize(img, mask, v):  # [4, 8]
    assert 4 <= v <= 8
    v = int(v)
    return _run(img, OP_POSTERIZE, iarg=v), mask


def Contrast(img, mask, v):  # [0.1,1.9]
    assert 0.1 <= v <= 1.9
    return _run(img, OP_CONTRAST, farg=v), mask


def Color(img, mask, v):  # [0.1,1.9]
    assert 0.1 <= v <= 1.9
    return _run(img, OP_COLOR, farg=v), mask


def Brightness(img, mask, v):  # [0.1,1.9]
    assert 0.1 <= v <= 1.9
    return _run(img, OP_BRIGHTNESS, farg=v), mask


def Sharpness(img, mask, v):  # [0.1,1.9]
    assert 0.1 <= v <= 1.9
    return _run(img, OP_SHARPNESS, farg=v), mask


def Cutout(img, mask, v):  # [0, 60] => percentage: [0, 0.2]
    assert 0.0 <= v <= 0.2
    if v <= 0.:
        return img, mask

    v = v * _size(img)[0]
    return CutoutAbs(img, mask, v)


def cutout_rect(w, h, v, x0, y0):
    """Inclusive, clipped pixel rectangle PIL.ImageDraw.rectangle fills for CutoutAbs' float box
    (data/basic.py:153-163): float corners are truncated, the far corner is inclusive."""
    x0 = int(max(0, x0 - v / 2.))
    y0 = int(max(0, y0 - v / 2.))
    x1 = min(w, x0 + v)
    y1 = min(h, y0 + v)
    return (x0, y0, min(int(x1), w - 1), min(int(y1), h - 1))


def CutoutAbs(img, mask, v):
    if v < 0:
        return img, mask
    w, h = _size(img)
    x0 = np.random.uniform(w)
    y0 = np.random.uniform(h)
    rect = cutout_rect(w, h, v, x0, y0)
    img = _run(img, OP_CUTOUT, rect=rect)
    if torch.is_tensor(mask):  # eager mode: label_color = 0
        mask = mask.clone()
        mask[rect[1]:rect[3] + 1, rect[0]:rect[2] + 1] = 0
    return img, mask


def augment_list(for_autoaug=False):  # same order as data/basic.py:231-243 -- the order IS the op index
    l = [
        (AutoContrast, 0, 1),
        (Invert, 0, 1),
        (Equalize, 0, 1),
        (Solarize, 0, 256),
        (Posterize, 4, 8),
        (Contrast, 0.1, 1.9),
        (Color, 0.1, 1.9),
        (Brightness, 0.1, 1.9),
        (Sharpness, 0.1, 1.9),
        (Cutout, 0, 0.2),
    ]
    if for_autoaug:
        raise NotImplementedError("for_autoaug ops (CutoutAbs/Posterize2/Translate*Abs) are not on the search path")
    return l


augment_dict = {fn.__name__: (fn, v1, v2) for fn, v1, v2 in augment_list()}


def get_augment(name):
    return augment_dict[name]


def apply_augment(img, mask, name, level):
    augment_fn, low, high = get_augment(name)
    return augment_fn(img.copy() if _is_ref(img) else img, mask.copy() if _is_ref(mask) else mask,
                      level * (high - low) + low)
