"""AdamW on a HIP kernel, as a torch.optim.Optimizer so the reference's
``CosineAnnealingWarmRestarts`` schedulers drive it unchanged (GAN_DANet_train.ipynb:L182-187)."""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch

from . import kern as K


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt-bias term).
    ``grad_scale`` multiplies every gradient as it is read (1/world_size after a summing all-reduce).

    ``sharded``: ``parallel.ShardedParam`` objects for parameters whose update is split over the ranks: this rank
    holds optimiser state for its slice only, updates that slice from its slice of the summed gradient
    (``ShardedParam.wait_grad``) and starts the all-gather of the updated weight.  ``state_dict`` /
    ``load_state_dict`` exchange FULL state tensors (collective: call them on every rank), so checkpoints do not
    depend on the world size.
    ``update_fn``: the element-wise update kernel (default: the HIP ``gd_adamw``; the CPU tests of the sharding logic
    pass the oracle's)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale: float = 1.0,
                 sharded: Iterable = (), update_fn: Optional[Callable] = None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = grad_scale
        self.sharded = {id(sp.p): sp for sp in sharded}
        self.update_fn = update_fn or K.adamw

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
                sp = self.sharded.get(id(p))
                st = self.state[p]
                if sp is not None:
                    g, target = sp.wait_grad(), sp.param_shard()
                else:
                    g, target = (p.grad if p.grad.is_contiguous() else p.grad.contiguous()), p.data
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(target, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(target, memory_format=torch.contiguous_format)
                st["step"] += 1
                self.update_fn(target, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"], b1, b2, group["eps"],
                               group["weight_decay"], self.grad_scale)
                if sp is not None:
                    sp.launch_all_gather()
        return loss

    # ---- checkpoints hold full tensors whatever the sharding ----
    def state_dict(self):
        if not self.sharded:
            return super().state_dict()
        saved = {}
        for p, st in self.state.items():
            sp = self.sharded.get(id(p))
            if sp is not None and st:
                saved[p] = (st["exp_avg"], st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = sp.gather_state(st["exp_avg"]), sp.gather_state(st["exp_avg_sq"])
        try:
            sd = super().state_dict()
            # the packed per-parameter dicts ARE self.state's dicts: copy them before the slices go back in
            sd["state"] = {k: dict(v) for k, v in sd["state"].items()}
            return sd
        finally:
            for p, (m, v) in saved.items():
                self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"] = m, v

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            sp = self.sharded.get(id(p))
            if sp is not None and st and st["exp_avg"].numel() == p.numel():
                st["exp_avg"], st["exp_avg_sq"] = sp.slice_state(st["exp_avg"]), sp.slice_state(st["exp_avg_sq"])
