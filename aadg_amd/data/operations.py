"""nn.Module wrappers of the float tensor ops -- API mirror of the reference's data/operations.py
(`_Operation` :16-137 and its 19 subclasses :142-399, same class names, constructor arguments, magnitude
scales and flip_magnitude flags).  forward(): mask ~ RelaxedBernoulli(temperature, p) in training /
Bernoulli(p) in eval; magnitude = clamp(_magnitude, range) * scale, sign-flipped per sample when
flip_magnitude; training output = mask*op(x) + (1-mask)*x clamped; eval applies op to the masked subset."""
from typing import Callable, Optional, Tuple

import torch
from torch import nn
from torch.distributions import Bernoulli, RelaxedBernoulli

from . import functional as Fn
from .kernels import get_sharpness_kernel

__all__ = ['ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'HorizontalFlip', 'VerticalFlip', 'Rotate',
           'Invert', 'Solarize', 'Posterize', 'Gray', 'Contrast', 'AutoContrast', 'Saturate', 'Brightness',
           'Hue', 'SamplePairing', 'Equalize', 'Sharpness']


def _checked_range(rng, what):
    if rng is not None and not (0 <= rng[0] < rng[1] <= 1):
        raise AssertionError("%s must satisfy 0 <= lo < hi <= 1, got %r" % (what, (rng,)))
    return rng


class _Operation(nn.Module):
    """One stochastic float op.  State (names are the reference's state_dict keys, data/operations.py:28-70):
    `_magnitude` / `_probability` are nn.Parameters when a range is given (learnable, clamped into the range on every
    read) and buffers otherwise (frozen); `temperature` is a buffer.  Ops without a magnitude keep `_magnitude = None`."""

    def __init__(self, operation: Optional[Callable], initial_magnitude: Optional[float] = None,
                 initial_probability: float = 0.5, magnitude_range: Optional[Tuple[float, float]] = None,
                 probability_range: Optional[Tuple[float, float]] = None, temperature: float = 0.1,
                 flip_magnitude: bool = False, magnitude_scale: float = 1, debug: bool = False):
        super().__init__()
        if not temperature > 0 or not magnitude_scale > 0:
            raise AssertionError("temperature and magnitude_scale must be positive")
        if probability_range is not None and not 0 <= initial_probability <= 1:
            raise AssertionError("initial_probability outside [0, 1]")
        has_mag = initial_magnitude is not None
        self.operation = operation
        self.magnitude_scale = magnitude_scale
        self.flip_magnitude = bool(flip_magnitude and has_mag)
        # a range only means something for a magnitude that exists
        self.magnitude_range = _checked_range(magnitude_range, "magnitude_range") if has_mag else None
        self.probability_range = _checked_range(probability_range, "probability_range")
        # registration order = the reference's parameter / state_dict order: magnitude, probability, temperature
        if has_mag:
            self._scalar("_magnitude", initial_magnitude, learnable=self.magnitude_range is not None)
        else:
            self._magnitude = None
        self._scalar("_probability", initial_probability, learnable=self.probability_range is not None)
        self._scalar("temperature", temperature, learnable=False)

    def _scalar(self, name, value, learnable):
        t = torch.full((1,), float(value))
        if learnable:
            setattr(self, name, nn.Parameter(t))
        else:
            self.register_buffer(name, t)

    # -- what forward() reads ------------------------------------------------------------------------------------
    @property
    def probability(self) -> torch.Tensor:
        p = self._probability
        return p if self.probability_range is None else p.clamp(self.probability_range[0], self.probability_range[1])

    @property
    def magnitude(self) -> Optional[torch.Tensor]:
        m = self._magnitude
        if m is None:
            return None
        if self.magnitude_range is not None:
            m = m.clamp(self.magnitude_range[0], self.magnitude_range[1])
        return m * self.magnitude_scale

    def get_mask(self, batch_size=None) -> torch.Tensor:
        """[B,1,1,1] gate per sample: relaxed (differentiable w.r.t. the probability) in training, hard 0/1 in eval"""
        dist = RelaxedBernoulli(self.temperature, self.probability) if self.training else Bernoulli(self.probability)
        shape = (batch_size, 1, 1)
        return dist.rsample(shape) if self.training else dist.sample(shape)

    def _signed_magnitude(self, n, device):
        mag = self.magnitude
        if not self.flip_magnitude:
            return mag
        # one fair coin per sample: 0 / 1 -> -1 / +1
        coins = torch.randint(2, (n,), dtype=torch.float32, device=device)
        return (2 * coins - 1) * mag

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        n = input.size(0)
        gate = self.get_mask(n)                       # draw order of the reference: gate first, then the signs
        mag = self._signed_magnitude(n, input.device)
        if self.training:
            changed = self.operation(input, mag)
            return (gate * changed + (1 - gate) * input).clamp_(0, 1)
        # eval: the op overwrites the gated samples of `input` itself
        rows = gate.reshape(-1).eq(1).nonzero().reshape(-1)
        if rows.numel():
            if torch.is_tensor(mag):
                mag = mag.repeat(rows.numel()) if mag.numel() == 1 else mag.index_select(0, rows)
            input.index_copy_(0, rows, self.operation(input.index_select(0, rows), mag))
        return input.clamp(0, 1)

    def extra_repr(self) -> str:
        def state(t, rng):
            return "%.3f (%s)" % (float(t.detach()), "frozen" if rng is None else "learnable")
        parts = ["probability=" + state(self.probability, self.probability_range)]
        if self._magnitude is not None:
            parts.append("magnitude=" + state(self.magnitude, self.magnitude_range))
        parts.append("temperature=%.3f" % float(self.temperature))
        return ", ".join(parts)


def _make(name, fn, has_mag=True, flip=False, scale=1.0):
    """Class factory: the 18 plain ops differ only in (function, has magnitude, flip_magnitude, scale)."""
    if has_mag:
        def __init__(self, initial_magnitude: float = 0.5, initial_probability: float = 0.5,
                     magnitude_range: Optional[Tuple[float, float]] = (0, 1),
                     probability_range: Optional[Tuple[float, float]] = (0, 1), temperature: float = 0.1,
                     magnitude_scale: float = scale, debug: bool = False):
            _Operation.__init__(self, fn, initial_magnitude, initial_probability, magnitude_range, probability_range,
                                temperature, flip_magnitude=flip, magnitude_scale=magnitude_scale, debug=debug)
    else:
        def __init__(self, initial_probability: float = 0.5, probability_range: Optional[Tuple[float, float]] = (0, 1),
                     temperature: float = 0.1, debug: bool = False):
            _Operation.__init__(self, fn, None, initial_probability, None, probability_range, temperature, debug=debug)
    return type(name, (_Operation,), {"__init__": __init__})


# geometric (magnitude scales 0.3 / 0.45 / 30 and sign flips as data/operations.py:142-231)
ShearX = _make('ShearX', Fn.shear_x, flip=True, scale=0.3)
ShearY = _make('ShearY', Fn.shear_y, flip=True, scale=0.3)
TranslateX = _make('TranslateX', Fn.translate_x, flip=True, scale=0.45)
TranslateY = _make('TranslateY', Fn.translate_y, flip=True, scale=0.45)
HorizontalFlip = _make('HorizontalFlip', Fn.hflip, has_mag=False)
VerticalFlip = _make('VerticalFlip', Fn.vflip, has_mag=False)
Rotate = _make('Rotate', Fn.rotate, flip=True, scale=30)
# colour (data/operations.py:237-352)
Invert = _make('Invert', Fn.invert, has_mag=False)
Solarize = _make('Solarize', Fn.solarize)
Posterize = _make('Posterize', Fn.posterize)
Gray = _make('Gray', Fn.gray, has_mag=False)
Contrast = _make('Contrast', Fn.contrast, flip=True)
AutoContrast = _make('AutoContrast', Fn.auto_contrast, has_mag=False)
Saturate = _make('Saturate', Fn.saturate, flip=True)
Brightness = _make('Brightness', Fn.brightness, flip=True)
Hue = _make('Hue', Fn.hue, scale=2)
SamplePairing = _make('SamplePairing', Fn.sample_pairing)
Equalize = _make('Equalize', Fn.equalize, has_mag=False)


class _KernelOperation(_Operation):
    def __init__(self, operation, kernel: torch.Tensor, initial_magnitude: float = 0.5, initial_probability: float = 0.5,
                 magnitude_range=(0, 1), probability_range=(0, 1), temperature: float = 0.1, flip_magnitude: bool = False,
                 magnitude_scale: float = 1, debug: bool = False):
        super().__init__(None, initial_magnitude, initial_probability, magnitude_range, probability_range, temperature,
                         flip_magnitude=flip_magnitude, magnitude_scale=magnitude_scale, debug=debug)
        self.register_buffer('kernel', kernel)
        self._original_operation = operation
        self.operation = self._operation

    def _operation(self, img: torch.Tensor, mag: torch.Tensor) -> torch.Tensor:
        return self._original_operation(img, mag, self.kernel)


class Sharpness(_KernelOperation):
    def __init__(self, initial_magnitude: float = 0.5, initial_probability: float = 0.5, magnitude_range=(0, 1),
                 probability_range=(0, 1), temperature: float = 0.1, debug: bool = False):
        super().__init__(Fn.sharpness, get_sharpness_kernel(), initial_magnitude, initial_probability, magnitude_range,
                         probability_range, temperature, flip_magnitude=True, debug=debug)
