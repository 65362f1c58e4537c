"""AdamW on a HIP kernel, as a torch.optim.Optimizer so the reference's
``CosineAnnealingWarmRestarts`` schedulers drive it unchanged (GAN_DANet_train.ipynb:L182-187)."""
from __future__ import annotations

import torch

from . import kern as K


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt-bias term).
    ``grad_scale`` multiplies every gradient as it is read (1/world_size after a summing all-reduce)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale: float = 1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = grad_scale

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                K.adamw(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], b1, b2, group["eps"],
                        group["weight_decay"], self.grad_scale)
        return loss
