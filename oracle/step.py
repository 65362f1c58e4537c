"""The G+D update of ``ModelTrainer.train`` as a plain function (CPU oracle).

Test infrastructure only (see ``oracle/__init__.py``).
Follows GAN_DANet_train.ipynb:L225-272 (cell 0 lines 217-264):

    hr = G(x)                                               L243
    D step:  zero_grad; real = D(target); fake = D(hr.detach());
             loss_D = (BCE(real,1) + BCE(fake,0)) / 2; backward; optD.step   L246-256
    G step:  zero_grad; fake = D(hr) (D already updated);
             adv = BCE(fake,1); pix = MSE(hr,target); ssim = 1-SSIM (computed,
             NOT in the loss); tv = TV(hr); perc = Perceptual(hr,target);
             w = epoch/epochs; loss_G = (1-w) pix + w adv + tv + perc;
             backward; optG.step                                             L259-269

The optimiser is the explicit AdamW of ``functional.adamw_update`` (lr 4e-4 D,
2e-4 G, betas (0.5, 0.999), weight decay 1e-4: L182-183).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
from torch import nn

from . import functional as OF


@dataclass
class AdamWState:
    lr: float
    betas: tuple = (0.5, 0.999)
    eps: float = 1e-8
    weight_decay: float = 1e-4
    step: int = 0
    m: Dict[int, torch.Tensor] = field(default_factory=dict)
    v: Dict[int, torch.Tensor] = field(default_factory=dict)

    def apply(self, params: List[nn.Parameter]) -> None:
        self.step += 1
        for i, p in enumerate(params):
            if p.grad is None:
                continue
            if i not in self.m:
                self.m[i] = torch.zeros_like(p)
                self.v[i] = torch.zeros_like(p)
            OF.adamw_update(p.data, p.grad, self.m[i], self.v[i], self.step, self.lr, self.betas[0],
                            self.betas[1], self.eps, self.weight_decay)


@dataclass
class StepResult:
    loss_d: float
    loss_g: float
    parts: Dict[str, float]
    hr: torch.Tensor


def combine_inputs(lr_grace_05: torch.Tensor, hr_aux: torch.Tensor) -> torch.Tensor:
    """the loader-loop preamble, GAN_DANet_train.ipynb:L218 (bicubic x0.5 of lr_grace_05), L223 (bicubic x0.25 of
    hr_aux), L224 (cat on channels); F.interpolate with a scale_factor uses 1/scale as the coordinate ratio and
    floor(in * scale) as the output size, no antialiasing."""
    h, w = lr_grace_05.shape[2] // 2, lr_grace_05.shape[3] // 2
    lr = OF.bicubic_resize(lr_grace_05, h, w, 2.0, 2.0)
    down = OF.bicubic_resize(hr_aux, hr_aux.shape[2] // 4, hr_aux.shape[3] // 4, 4.0, 4.0)
    return torch.cat([lr, down], dim=1)


def train_step(G: nn.Module, D: nn.Module, opt_g: AdamWState, opt_d: AdamWState, x: torch.Tensor,
               target: torch.Tensor, loss_weight: float, tv_weight: float = 1e-5,
               perceptual: Optional[nn.Module] = None, compute_ssim: bool = True,
               grad_hook=None) -> StepResult:
    """One G+D update.  ``x`` is the already combined generator input
    (B, 1+C_aux, H, W) of L232, ``target`` the 4H x 4W image (``lr_grace_025``).
    ``grad_hook(params)`` (optional) runs after each backward, before the
    optimiser -- where a data-parallel all-reduce goes."""
    g_params = [p for p in G.parameters()]
    d_params = [p for p in D.parameters()]
    hr = G(x)

    for p in d_params:
        p.grad = None
    real = D(target)
    fake = D(hr.detach())
    loss_d = (OF.bce_with_logits(real, torch.ones_like(real)) + OF.bce_with_logits(fake, torch.zeros_like(fake))) / 2
    loss_d.backward()
    if grad_hook is not None:
        grad_hook(d_params)
    opt_d.apply(d_params)

    for p in g_params:
        p.grad = None
    fake = D(hr)
    adv = OF.bce_with_logits(fake, torch.ones_like(fake))
    pix = OF.mse(hr, target)
    ssim_term = (1 - OF.ssim(hr, target)) if compute_ssim else torch.zeros(())
    tv = OF.tv_loss(hr, tv_weight)
    perc = perceptual(hr, target) if perceptual is not None else torch.zeros((), dtype=hr.dtype)
    loss_g = (1 - loss_weight) * pix + loss_weight * adv + tv + perc
    loss_g.backward()
    if grad_hook is not None:
        grad_hook(g_params)
    opt_g.apply(g_params)

    return StepResult(float(loss_d.detach()), float(loss_g.detach()),
                      {"adv": float(adv), "pix": float(pix), "ssim": float(ssim_term), "tv": float(tv),
                       "perc": float(perc)}, hr.detach())
