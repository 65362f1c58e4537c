"""Deterministic parameter fill shared by the fixture generator (applied to the
*reference* modules in the build container) and by the tests (applied to the
oracle / product modules), so a fixture needs no weight blob.  Tensor k of the
``state_dict`` (in key order) is drawn from a CPU ``torch.Generator`` seeded
with 1000 + k: Kaiming-scaled normal conv/linear weights, BN gamma ~ 1 +- 0.1,
small biases / running stats.  (A sin() closed form was tried first: it makes
near-degenerate channels that BatchNorm amplifies until the reference's own
fp32 run differs from its fp64 run by 5e-4 in y and 10 % in gradients --
useless as a parity pin.  With this fill fp32-vs-fp64 is 1e-6 / 1e-5.)
"""
from __future__ import annotations

import math

import torch


def _normal(n: int, k: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + k)
    return torch.randn(n, generator=g, dtype=torch.float32).double()


@torch.no_grad()
def fill_module(module: torch.nn.Module, pam_gamma: float = 0.7, cam_gamma: float = 0.3) -> None:
    sd = module.state_dict()
    for k, (name, t) in enumerate(sd.items()):
        if t.numel() == 0 or name.endswith("num_batches_tracked") or name == "window":
            continue
        n = t.numel()
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "gamma":
            val = torch.full((n,), pam_gamma if "position" in name else cam_gamma, dtype=torch.float64)
        elif leaf == "running_mean":
            val = 0.05 * _normal(n, k)
        elif leaf == "running_var":
            val = 1.0 + 0.2 * _normal(n, k) ** 2
        elif t.dim() == 1 and leaf == "weight":      # BN gamma
            val = 1.0 + 0.1 * _normal(n, k)
        elif t.dim() == 1:                           # biases (BN beta, conv/linear bias)
            val = 0.1 * _normal(n, k)
        else:                                        # conv / linear weights
            fan_in = n // t.shape[0]
            val = math.sqrt(2.0 / fan_in) * _normal(n, k)
            if ".query." in "." + name or ".key." in "." + name:
                # PAM has no 1/sqrt(d) scale: Kaiming-sized q/k give |energy| ~ 50 and an argmax-like
                # softmax whose near-ties flip under fp32 rounding (reference fp32 vs fp64: 1e-2 in
                # gradients).  Quarter-size q/k keep the softmax smooth (1e-4).
                val = 0.25 * val
        t.copy_(val.reshape(t.shape).to(t.dtype))


def seeded(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale
