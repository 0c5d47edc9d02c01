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


class _Operation(nn.Module):
    def __init__(self, operation: Optional[Callable], initial_magnitude: Optional[float] = None,
                 initial_probability: float = 0.5, magnitude_range: Optional[Tuple[float, float]] = None,
                 probability_range: Optional[Tuple[float, float]] = None, temperature: float = 0.1,
                 flip_magnitude: bool = False, magnitude_scale: float = 1, debug: bool = False):
        super().__init__()
        self.operation = operation
        self.magnitude_range = None
        if initial_magnitude is None:
            self._magnitude = None
        elif magnitude_range is None:
            self.register_buffer("_magnitude", torch.empty(1).fill_(initial_magnitude))
        else:
            self._magnitude = nn.Parameter(torch.empty(1).fill_(initial_magnitude))
            assert 0 <= magnitude_range[0] < magnitude_range[1] <= 1
            self.magnitude_range = magnitude_range
        self.probability_range = probability_range
        if probability_range is None:
            self.register_buffer("_probability", torch.empty(1).fill_(initial_probability))
        else:
            assert 0 <= initial_probability <= 1
            assert 0 <= probability_range[0] < probability_range[1] <= 1
            self._probability = nn.Parameter(torch.empty(1).fill_(initial_probability))
        assert 0 < temperature
        self.register_buffer("temperature", torch.empty(1).fill_(temperature))
        self.flip_magnitude = flip_magnitude and (self._magnitude is not None)
        assert 0 < magnitude_scale
        self.magnitude_scale = magnitude_scale
        self.debug = debug
        self._py_magnitude = initial_magnitude
        self._py_probability = initial_probability

    @property
    def magnitude(self) -> Optional[torch.Tensor]:
        if self._magnitude is None:
            return None
        mag = self._magnitude
        if self.magnitude_range is not None:
            mag = mag.clamp(*self.magnitude_range)
        m = mag * self.magnitude_scale
        self._py_magnitude = m.item()
        return m

    @property
    def probability(self) -> torch.Tensor:
        if self.probability_range is None:
            return self._probability
        p = self._probability.clamp(*self.probability_range)
        self._py_probability = p.item()
        return p

    def get_mask(self, batch_size=None) -> torch.Tensor:
        size = (batch_size, 1, 1)
        if self.training:
            return RelaxedBernoulli(self.temperature, self.probability).rsample(size)
        return Bernoulli(self.probability).sample(size)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        mask = self.get_mask(input.size(0))
        mag = self.magnitude
        if self.flip_magnitude:
            sign = torch.randint(2, (input.size(0),), dtype=torch.float32, device=input.device).mul_(2).sub_(1)
            mag = sign * mag
        if self.training:
            return (mask * self.operation(input, mag) + (1 - mask) * input).clamp_(0, 1)
        keep = mask.reshape(-1) == 1
        output = input
        if torch.is_tensor(mag):
            mag = mag.repeat(int(keep.sum())) if mag.size(0) == 1 else mag[keep]
        if bool(keep.any()):
            output[keep, ...] = self.operation(output[keep, ...], mag)
        return output.clamp(0, 1)

    def __repr__(self) -> str:
        s = self.__class__.__name__
        s += f"(probability={self._py_probability:.3f} ({'frozen' if self.probability_range is None else 'learnable'}), "
        if self.magnitude is not None:
            s += f"magnitude={self._py_magnitude:.3f} ({'frozen' if self.magnitude_range is None else 'learnable'}), "
        return s + f"temperature={self.temperature.item():.3f})"


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
