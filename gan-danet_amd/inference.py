"""Forward-only consumer of the generator: the device side of ``test.ipynb``'s ``predict_and_plot`` loop body
(c1:123-171) -- SURVEY.md row f3.  File formats (HDF5 / NetCDF), masks, scalers and plots stay in the caller's
Python; what runs per batch on the GPU is here:

    xin  = cat([lr_grace_025, aux], 1)                                   c1:154   (torch.cat: a copy, plumbing)
    yhat = model(xin)                       eval-mode generator          c1:157   (180 x 88 tiles: PAM over 15 840 tokens)
    yhat = F.interpolate(yhat, scale_factor=1.25, mode='bicubic')        c1:158   -> ``bicubic_resize``
    yhat = apply_mild_histogram_matching(yhat, lr_grace_025, weight)     c1:161   -> ``mild_histogram_matching``
    hr_grace = F.interpolate(lr_grace_025, scale_factor=4, 'bicubic')    c1:164   -> ``bicubic_resize``
    yhat = smooth_blend(yhat, hr_grace, region)                          c1:165   -> ``smooth_blend``
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np
import torch

from . import kern as K
from . import ops


def bicubic_resize(x: torch.Tensor, scale_factor: float) -> torch.Tensor:
    """``F.interpolate(x, scale_factor=s, mode='bicubic', align_corners=False)``: output size floor(in * s), source
    coordinate (dst + 0.5) / s - 0.5, A = -0.75, border-clamped taps (differentiable: ``ops.BicubicFn``)"""
    Ho, Wo = int(math.floor(x.shape[2] * scale_factor)), int(math.floor(x.shape[3] * scale_factor))
    return ops.BicubicFn.apply(x, Ho, Wo, 1.0 / scale_factor, 1.0 / scale_factor)


def blend_mask(region: Sequence[int], sigma: int = 5) -> np.ndarray:
    """the feathered window of ``smooth_blend`` (c1:92-97): ramps of width ``sigma`` on the four edges, then a Gaussian
    filter -- a host-side constant of the region's size, built with the calls the notebook makes"""
    from scipy.ndimage import gaussian_filter
    sr, er, sc, ec = region
    mask = np.ones((er - sr, ec - sc), dtype=float)
    mask[0:sigma, :] = np.linspace(0, 1, sigma)[:, None]
    mask[-sigma:, :] = np.linspace(1, 0, sigma)[:, None]
    mask[:, 0:sigma] = np.maximum(mask[:, 0:sigma], np.linspace(0, 1, sigma)[None, :])
    mask[:, -sigma:] = np.maximum(mask[:, -sigma:], np.linspace(1, 0, sigma)[None, :])
    return gaussian_filter(mask, sigma=sigma)


@torch.no_grad()
def smooth_blend(hr_generated: torch.Tensor, hr_grace: torch.Tensor, region: Sequence[int], sigma: int = 5) -> torch.Tensor:
    """c1:87-101: feather ``hr_grace`` into ``hr_generated`` over ``region`` = (row0, row1, col0, col1), IN PLACE on
    ``hr_generated`` as the notebook does; one launch (``gd_blend_region``)"""
    mask = torch.from_numpy(blend_mask(region, sigma)).to(dtype=torch.float32, device=hr_generated.device).contiguous()
    return K.blend_region(hr_generated, hr_grace.contiguous(), mask, region)


def mild_histogram_matching(hr_generated: torch.Tensor, lr_grace_025: torch.Tensor, weight: float = 0.0) -> torch.Tensor:
    """c1:69-85 ``apply_mild_histogram_matching``: per sample, ``(1 - weight) * source + weight * matched`` where
    ``matched`` maps the sample's empirical CDF onto the reference field's (np.unique / np.interp semantics).  The
    notebook calls it with ``weight = 0.0`` (c1:161), where the result IS the source: that case returns the input
    untouched.  Otherwise one launch sequence per sample (``gd_hist_match``: radix sorts + a quantile-interpolation
    kernel); the result is float64, as the notebook's numpy code returns."""
    if weight == 0.0:
        return hr_generated
    return K.hist_match(hr_generated.contiguous(), lr_grace_025.contiguous(), weight)


@torch.no_grad()
def predict_batch(model: torch.nn.Module, lr_grace_025: torch.Tensor, aux: torch.Tensor,
                  region: Tuple[int, int, int, int] = (0, 90, 0, 44), upscale: float = 1.25, hist_weight: float = 0.0,
                  blend_with: float = 4.0) -> torch.Tensor:
    """one iteration of the loader loop of ``predict_and_plot`` (c1:149-167) on the device.  As in the notebook the blend
    target is the x4 bicubic of ``lr_grace_025`` and the blend happens on the x1.25 image, so ``region`` must lie inside
    both."""
    xin = torch.cat([lr_grace_025, aux], dim=1)                          # layout copy
    yhat = bicubic_resize(model(xin), upscale)
    yhat = mild_histogram_matching(yhat, lr_grace_025, hist_weight)
    if yhat.dtype != torch.float32:          # the notebook continues in float64 from here; the blend kernel is fp32
        yhat = yhat.float()
    hr_grace = bicubic_resize(lr_grace_025, blend_with)
    sr, er, sc, ec = region
    if yhat.shape == hr_grace.shape:
        return smooth_blend(yhat, hr_grace, region)
    # the notebook slices both images with the same indices (c1:90-91): images of different size share the corner
    mask = torch.from_numpy(blend_mask(region)).to(dtype=torch.float32, device=yhat.device).contiguous()
    patch = hr_grace[:, :, sr:er, sc:ec].contiguous()
    canvas = yhat[:, :, sr:er, sc:ec].contiguous()
    K.blend_region(canvas, patch, mask, (0, er - sr, 0, ec - sc))
    yhat[:, :, sr:er, sc:ec] = canvas                                     # slice copy (plumbing)
    return yhat
