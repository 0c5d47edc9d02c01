"""3x3 / 5x5 kernels for the float sharpen / blur ops -- API mirror of data/kernels.py:9-35."""
from typing import Optional

import torch


def get_sharpness_kernel(device: Optional[torch.device] = None) -> torch.Tensor:
    """ones(3,3) with centre 5, divided by 13 (PIL's SMOOTH filter)."""
    kernel = torch.ones(3, 3)
    kernel[1, 1] = 5
    return (kernel / 13).to(device=device)


def _gaussian(sigma: torch.Tensor, kernel_size: int, device: Optional[torch.device] = None) -> torch.Tensor:
    """exp(-(i^2+j^2)/(2 v)) normalised to sum 1, where v := mean(sigma)^2 (the reference squares the mean
    sigma and then uses it as the variance directly, data/kernels.py:21-24)."""
    radius = kernel_size // 2
    var = sigma.mean().pow(2)
    sq = torch.arange(-radius, radius + 1, dtype=torch.float32).pow(2).view(-1, 1)
    grid = sq + sq.t()
    k = (-grid / (2 * var.cpu())).exp()
    return (k / k.sum()).to(device=device)


def get_gaussian_3x3kernel(sigma: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    return _gaussian(sigma, 3, device)


def get_gaussian_5x5kernel(sigma: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    return _gaussian(sigma, 5, device)
