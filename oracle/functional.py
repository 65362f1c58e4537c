"""Explicit arithmetic of every op on the GAN-DANet hot path (CPU, torch).

Test infrastructure only (see ``oracle/__init__.py``).  Every function works
on plain tensors, is differentiable through torch autograd (so backward
references come for free) and runs in the dtype it is given (fp32 for parity,
fp64 for gradient checks).  Citations are into ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# convolution / linear: the reference calls ATen directly, so does the oracle
# --------------------------------------------------------------------------
def conv2d(x: Tensor, w: Tensor, b: Optional[Tensor] = None, stride: int = 1, padding: int = 0,
           groups: int = 1) -> Tensor:
    """nn.Conv2d forward (generator.py:20,34,63,108-110,148,188,214,218,222,228;
    discriminator.py:62-65).  Cross-correlation, zero padding."""
    return F.conv2d(x, w, b, stride=stride, padding=padding, groups=groups)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Linear / nn.LazyLinear forward (discriminator.py:66-67,76-77): x @ w.T + b."""
    y = x @ w.t()
    return y if b is None else y + b


def leaky_relu(x: Tensor, slope: float = 0.2) -> Tensor:
    """nn.LeakyReLU(0.2) (discriminator.py:68)."""
    return torch.where(x >= 0, x, x * slope)


def relu(x: Tensor) -> Tensor:
    return torch.clamp_min(x, 0)


# --------------------------------------------------------------------------
# batch norm (training and eval), written out
# --------------------------------------------------------------------------
def batch_norm_train(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor],
                     running_var: Optional[Tensor], momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm2d in train mode (generator.py:32,61,149,189,219,223).

    Normalises with the biased batch variance; running_var is updated with the
    unbiased one (n/(n-1)), running stats blended with ``momentum``.
    Updates the running buffers in place (no grad)."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    if running_mean is not None:
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(momentum * mean.detach())
            running_var.mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
    inv = torch.rsqrt(var + eps)
    return (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


def batch_norm_eval(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor,
                    eps: float = 1e-5) -> Tensor:
    inv = torch.rsqrt(running_var + eps)
    return (x - running_mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


# --------------------------------------------------------------------------
# dual attention
# --------------------------------------------------------------------------
def pam_attention(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """Core of PAMModule.forward (generator.py:115-121).

    q, k: (B, r, N); v: (B, C, N).  energy[i, j] = sum_d q[d, i] k[d, j] (NO
    1/sqrt(d) scale), attention = softmax over j, out[c, i] = sum_j v[c, j]
    attention[i, j].  Returns (B, C, N)."""
    energy = torch.einsum("bdi,bdj->bij", q, k)
    energy = energy - energy.amax(dim=-1, keepdim=True)
    p = torch.exp(energy)
    p = p / p.sum(dim=-1, keepdim=True)
    return torch.einsum("bcj,bij->bci", v, p)


def pam(x: Tensor, wq: Tensor, bq: Tensor, wk: Tensor, bk: Tensor, wv: Tensor, bv: Tensor,
        gamma: Tensor) -> Tensor:
    """PAMModule.forward (generator.py:113-122): gamma * attention(x) + x."""
    b, c, h, w = x.shape
    q = conv2d(x, wq, bq).reshape(b, -1, h * w)
    k = conv2d(x, wk, bk).reshape(b, -1, h * w)
    v = conv2d(x, wv, bv).reshape(b, -1, h * w)
    out = pam_attention(q, k, v).reshape(b, c, h, w)
    return gamma * out + x


def cam(x: Tensor, gamma: Tensor) -> Tensor:
    """CAMModule.forward (generator.py:130-139).

    energy = X X^T (C x C, reduction over N); attention = softmax(rowmax(energy)
    - energy); out = attention X; gamma * out + x.  Follows the reference
    literally (including the max subtraction that autograd sees)."""
    b, c, h, w = x.shape
    xf = x.reshape(b, c, h * w)
    energy = torch.einsum("bcn,bdn->bcd", xf, xf)
    energy_new = energy.amax(dim=-1, keepdim=True) - energy
    energy_new = energy_new - energy_new.amax(dim=-1, keepdim=True)
    p = torch.exp(energy_new)
    p = p / p.sum(dim=-1, keepdim=True)
    out = torch.einsum("bcd,bdn->bcn", p, xf).reshape(b, c, h, w)
    return gamma * out + x


# --------------------------------------------------------------------------
# resampling (ATen semantics; written out so the kernels have a formula to follow)
# --------------------------------------------------------------------------
def _cubic_weights(t: Tensor, a: float = -0.75) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Keys cubic-convolution coefficients used by ATen upsample_bicubic2d (A = -0.75)."""
    def c1(x):  # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1

    def c2(x):  # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a

    return c2(t + 1), c1(t), c1(1 - t), c2(2 - t)


def bicubic_resize(x: Tensor, out_h: int, out_w: int, scale_h: Optional[float] = None,
                   scale_w: Optional[float] = None) -> Tensor:
    """F.interpolate(mode='bicubic', align_corners=False) written out
    (nn.Upsample at generator.py:221,225; GAN_DANet_train.ipynb:L226,L231).

    src = (dst + 0.5) * s - 0.5 with s = 1/scale_factor when a scale factor was
    given (``scale_h/scale_w`` = that 1/scale_factor) else in/out; taps at
    floor(src)-1..+2, indices clamped to the border, no clamp of src."""
    b, c, h, w = x.shape
    sh = (h / out_h) if scale_h is None else scale_h
    sw = (w / out_w) if scale_w is None else scale_w

    def axis(n_in, n_out, s):
        dst = torch.arange(n_out, dtype=x.dtype)
        src = (dst + 0.5) * s - 0.5
        i0 = torch.floor(src)
        t = src - i0
        i0 = i0.long()
        idx = torch.stack([(i0 + d).clamp(0, n_in - 1) for d in (-1, 0, 1, 2)], 0)  # (4, n_out)
        wts = torch.stack(_cubic_weights(t), 0)  # (4, n_out)
        return idx, wts

    iy, wy = axis(h, out_h, sh)
    ix, wx = axis(w, out_w, sw)
    rows = sum(x[:, :, iy[d], :] * wy[d][None, None, :, None] for d in range(4))  # (B,C,out_h,W)
    return sum(rows[:, :, :, ix[d]] * wx[d][None, None, None, :] for d in range(4))


def bilinear_resize(x: Tensor, out_h: int, out_w: int) -> Tensor:
    """F.interpolate(size=..., mode='bilinear', align_corners=False) written out
    (generator.py:244).  src = max((dst+0.5)*in/out - 0.5, 0)."""
    b, c, h, w = x.shape

    def axis(n_in, n_out):
        dst = torch.arange(n_out, dtype=x.dtype)
        src = ((dst + 0.5) * (n_in / n_out) - 0.5).clamp_min(0)
        i0 = torch.floor(src).long().clamp_max(n_in - 1)
        i1 = (i0 + 1).clamp_max(n_in - 1)
        lam = src - i0.to(x.dtype)
        return i0, i1, lam

    y0, y1, ly = axis(h, out_h)
    x0, x1, lx = axis(w, out_w)
    rows = x[:, :, y0, :] * (1 - ly)[None, None, :, None] + x[:, :, y1, :] * ly[None, None, :, None]
    return rows[:, :, :, x0] * (1 - lx)[None, None, None, :] + rows[:, :, :, x1] * lx[None, None, None, :]


def max_pool2(x: Tensor) -> Tensor:
    """nn.MaxPool2d(2, 2) of the VGG19 feature stack."""
    return F.max_pool2d(x, 2, 2)


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def bce_with_logits(z: Tensor, target: Tensor) -> Tensor:
    """torch.nn.BCEWithLogitsLoss() mean reduction (GAN_DANet_train.ipynb:L190,L252-253,L261)."""
    return (torch.clamp_min(z, 0) - z * target + torch.log1p(torch.exp(-z.abs()))).mean()


def mse(a: Tensor, b: Tensor) -> Tensor:
    """torch.nn.MSELoss() (GAN_DANet_train.ipynb:L191,L262)."""
    return ((a - b) ** 2).mean()


def l1(a: Tensor, b: Tensor) -> Tensor:
    """F.l1_loss mean (losses.py:72)."""
    return (a - b).abs().mean()


def tv_loss(x: Tensor, weight: float = 1.0) -> Tensor:
    """TVLoss.forward (losses.py:81-87).  Note the counts include batch and
    channel and the result is divided by batch size again."""
    bsz = x.shape[0]
    h_tv = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum()
    count_h = x[:, :, 1:, :].numel()
    count_w = x[:, :, :, 1:].numel()
    return weight * 2 * (h_tv / count_h + w_tv / count_w) / bsz


def gaussian_window(window_size: int = 11, sigma: float = 1.5, dtype=torch.float32) -> Tensor:
    """SSIM._gaussian/_create_window (losses.py:98-107): normalised 1-D Gaussian, outer product."""
    coords = torch.arange(window_size, dtype=torch.float32)
    g = torch.exp(-((coords - window_size // 2) ** 2) / (2 * sigma ** 2))
    g = (g / g.sum()).unsqueeze(1)
    return (g @ g.t()).to(dtype)


def ssim(img1: Tensor, img2: Tensor, window_size: int = 11, size_average: bool = True) -> Tensor:
    """SSIM._ssim (losses.py:109-136): five depthwise Gaussian convs with zero
    padding, C1 = 1e-4, C2 = 9e-4."""
    ch = img1.shape[1]
    win = gaussian_window(window_size, 1.5, img1.dtype).expand(ch, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    mu1 = F.conv2d(img1, win, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, win, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(img1 * img1, win, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, win, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, win, padding=pad, groups=ch) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


# VGG19 "E" configuration, features[:21] (torchvision.models.vgg19; restated
# from the published architecture -- torchvision is absent here).
# entries: ("conv", cin, cout) | ("relu",) | ("pool",)
VGG19_FEATURES_21 = (
    ("conv", 3, 64), ("relu",), ("conv", 64, 64), ("relu",), ("pool",),
    ("conv", 64, 128), ("relu",), ("conv", 128, 128), ("relu",), ("pool",),
    ("conv", 128, 256), ("relu",), ("conv", 256, 256), ("relu",), ("conv", 256, 256), ("relu",),
    ("conv", 256, 256), ("relu",), ("pool",),
    ("conv", 256, 512), ("relu",),
)


def perceptual(x: Tensor, y: Tensor, vgg_params: Sequence[Tuple[Tensor, Tensor]],
               feature_layers: Sequence[int] = (1, 6, 11, 20)) -> Tensor:
    """PerceptualLoss.forward (losses.py:63-73): 1-channel inputs repeated to 3
    channels, both walked through VGG19.features[: max+1], L1 summed at the
    listed indices.  ``vgg_params`` = (weight, bias) of the conv layers in
    order.  ReLU is in-place in torchvision, so index 0's output is already
    rectified when index 1 is tapped -- same numbers either way."""
    xf = x if x.shape[1] == 3 else x.repeat(1, 3, 1, 1)
    yf = y if y.shape[1] == 3 else y.repeat(1, 3, 1, 1)
    taps = set(feature_layers)
    loss = torch.zeros((), dtype=x.dtype)
    ci = 0
    for idx, ent in enumerate(VGG19_FEATURES_21[: max(taps) + 1]):
        if ent[0] == "conv":
            w, b = vgg_params[ci]
            ci += 1
            xf, yf = conv2d(xf, w, b, padding=1), conv2d(yf, w, b, padding=1)
        elif ent[0] == "relu":
            xf, yf = relu(xf), relu(yf)
        else:
            xf, yf = max_pool2(xf), max_pool2(yf)
        if idx in taps:
            loss = loss + l1(xf, yf)
    return loss


# --------------------------------------------------------------------------
# optimiser / schedule
# --------------------------------------------------------------------------
@torch.no_grad()
def adamw_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.5,
                 beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-4) -> None:
    """torch.optim.AdamW single-tensor update (GAN_DANet_train.ipynb:L182-183):
    decoupled weight decay, bias correction, denom = sqrt(v)/sqrt(1-b2^t) + eps.
    ``step`` is the 1-based step count.  In place on p, m, v."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def cosine_warm_restarts_lr(base_lr: float, epoch: int, t0: int = 10, t_mult: int = 2, eta_min: float = 1e-6) -> float:
    """CosineAnnealingWarmRestarts(T_0=10, T_mult=2, eta_min=1e-6) value after
    ``epoch`` calls of ``scheduler.step()`` (GAN_DANet_train.ipynb:L186-187,L294-295)."""
    t_i, t_cur = t0, epoch
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= t_mult
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2
