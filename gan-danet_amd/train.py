"""The G+D update of ``ModelTrainer.train`` (GAN_DANet_train.ipynb:L225-272) driven through the HIP-backed
modules, sharded over GPUs with ``parallel.GradReducer``.

Order kept from the reference: ONE generator forward per step, reused by both updates; D is updated BEFORE
the generator loss is evaluated; ``loss_G = (1-w) MSE + w BCE(D(hr), 1) + TV + Perceptual`` with
``w = epoch/epochs``; SSIM is evaluated and not used.  Deliberate, observable-state-preserving differences:

* during the G step D's parameters do not require grad, so its weight gradients (which the reference computes
  and then discards at the next ``optimizer_D.zero_grad()``, L246) are never formed -- for Discriminator1.fc1
  that is 2.1e9 elements per step at 256x256 tiles;
* loss scalars stay on the device; ``.item()`` (the reference's two host syncs per step, L271-272) is left to
  the caller.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import nn

from . import kern as K
from . import ops
from .optim import AdamW
from .parallel import GradReducer, shard_big_params, world_size


@dataclass
class StepOutput:
    loss_d: torch.Tensor
    loss_g: torch.Tensor
    parts: Dict[str, torch.Tensor]
    hr: torch.Tensor


class _NoReduce:
    def reduce(self) -> None:
        return None


class GanTrainer:
    def __init__(self, G: nn.Module, D: nn.Module, perceptual: Optional[nn.Module] = None, lr_g: float = 2e-4,
                 lr_d: float = 4e-4, betas=(0.5, 0.999), weight_decay: float = 1e-4, tv_weight: float = 1e-5,
                 compute_ssim: bool = True, tv_global_batch_semantics: bool = False,
                 batch_real_fake: bool = True, input_attention: Optional[nn.Module] = None,
                 reduce_gradients: bool = True, external_world: Optional[int] = None,
                 shard_bytes: int = 256 << 20, sync_bn: Optional[bool] = None) -> None:
        self.G, self.D, self.perceptual = G, D, perceptual
        # sync_bn: BatchNorm statistics reduced over all ranks (config.sync_bn; process-wide).  With
        # tv_global_batch_semantics=True a world of W ranks on B/W samples each then takes the step one device takes on B
        if sync_bn is not None:
            from .config import set_sync_bn
            set_sync_bn(sync_bn)
        # optional gate on the combined input (the notebook's attention_module / senet_module, L145-171,
        # L229-232: SqueezeExcitation or CBAMBlock); its parameters join the generator's optimiser (L165-175)
        self.input_attention = input_attention
        g_params = list(G.parameters()) + (list(input_attention.parameters()) if input_attention is not None else [])
        # reduce_gradients=False: this process trains an INDEPENDENT replica (an ensemble member, checkpoint.py) even
        # when torch.distributed is initialised -- no gradient exchange, no 1/world scaling
        # external_world = W: the caller sums this trainer's gradients with W - 1 other shards itself between
        # d_backward / g_backward and the optimiser steps (no torch.distributed): same 1/W scale as W ranks
        ws = world_size() if reduce_gradients else 1
        if external_world is not None:
            reduce_gradients, ws = False, int(external_world)
        self._reduce = reduce_gradients
        # discriminator tensors of >= shard_bytes (Discriminator1.fc1: 8.6 GB at 256x256 tiles) take the
        # reduce-scatter -> 1/world AdamW -> all-gather route under data parallelism (parallel.ShardedParam)
        self.sharded = shard_big_params(D, shard_bytes) if (reduce_gradients and shard_bytes > 0) else []
        self.opt_d = AdamW(D.parameters(), lr=lr_d, betas=betas, weight_decay=weight_decay, grad_scale=1.0 / ws,
                           sharded=self.sharded)
        self.opt_g = AdamW(g_params, lr=lr_g, betas=betas, weight_decay=weight_decay, grad_scale=1.0 / ws)
        self.red_d = GradReducer(D.parameters(), sharded=self.sharded) if reduce_gradients else _NoReduce()
        self.red_g = GradReducer(g_params) if reduce_gradients else _NoReduce()
        # TVLoss divides by the batch size twice (losses.py:82-87): per-shard TV averaged over ranks is `world`
        # times the single-device global-batch value.  Default = plain DDP semantics (per-shard loss, as the
        # per-shard oracle computes it); set tv_global_batch_semantics to scale it back by 1/world.
        self.tv_weight = tv_weight / ws if tv_global_batch_semantics else tv_weight
        self.compute_ssim = compute_ssim
        # D(cat[real, fake]) == (D(real), D(fake)) only for a discriminator WITHOUT batch statistics: with BatchNorm
        # (SRGAND has ten) the mixed batch would change the statistics and update the running stats once instead
        # of twice (the reference calls D separately, L247-248) -- such a D always takes the two-pass form
        self._d_has_bn = any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in D.modules())
        self.batch_real_fake = batch_real_fake and not self._d_has_bn
        self._world = ws                       # the world this trainer's 1/world gradient scale was built for

    def step_from_batch(self, lr_grace_05: torch.Tensor, lr_grace_025: torch.Tensor, hr_aux: torch.Tensor,
                        loss_weight: float) -> StepOutput:
        """one iteration of the notebook's loader loop (L217-272): the input preamble -- bicubic x0.5 of
        ``lr_grace_05``, bicubic x0.25 of ``hr_aux``, channel cat (L218-224) -- as one fused launch, then ``step``
        with ``lr_grace_025`` as the discriminator's real sample / the pixel target"""
        return self.step(K.combine_inputs(lr_grace_05, hr_aux, 0.5, 0.25), lr_grace_025, loss_weight)

    def step(self, x: torch.Tensor, target: torch.Tensor, loss_weight: float) -> StepOutput:
        """one G+D update (L243-269) = d_backward -> reduce -> D.step -> g_backward -> reduce -> G.step.  The two
        backward phases are separate methods so that an EXTERNAL exchange of gradients (tests emulating N ranks in
        one process, other communication layers) can sit where ``GradReducer.reduce`` does."""
        if self._reduce and world_size() != self._world:
            raise RuntimeError(f"GanTrainer was built for world size {self._world} but torch.distributed now reports "
                               f"{world_size()}: construct the trainer AFTER init_process_group (its AdamW gradient "
                               "scale and all-reduce hooks are fixed at construction)")
        st = self.d_backward(x, target)
        self.red_d.reduce()
        self.opt_d.step()
        self.g_backward(st, target, loss_weight)
        self.red_g.reduce()
        self.opt_g.step()
        return self.finish(st)

    def sync_params(self) -> None:
        """wait for the all-gathers of sharded weights still in flight (before reading D's parameters outside a
        forward: checkpoints, tests)"""
        for sp in self.sharded:
            sp.wait_param()

    def d_backward(self, x: torch.Tensor, target: torch.Tensor) -> dict:
        """generator forward (ONE per step, reused by both updates, L243) and the discriminator's loss + backward
        (L246-255); gradients are left un-reduced in ``D.parameters()``"""
        G, D = self.G, self.D
        hr = G(x if self.input_attention is None else self.input_attention(x))

        # ---- discriminator update (L246-256) ----
        self.opt_d.zero_grad(set_to_none=True)
        if self.batch_real_fake and target.shape == hr.shape:
            # D has no BatchNorm (checked in __init__), so D(cat[real, fake]) == (D(real), D(fake)) exactly; one pass
            # streams fc1's weights once per forward / data-gradient / weight-gradient instead of twice
            nb = target.shape[0]
            both = torch.empty((2 * nb,) + tuple(target.shape[1:]), device=target.device, dtype=torch.float32)
            K.copy_slab(target if target.is_contiguous() else target.contiguous(), both[:nb])
            K.copy_slab(hr.detach(), both[nb:])
            logits = D(both)
            real, fake = logits[:nb], logits[nb:]
        else:
            real = D(target)
            fake = D(hr.detach())
        loss_d = ops.weighted_sum([0.5, 0.5], [ops.bce_with_logits(real, 1.0), ops.bce_with_logits(fake, 0.0)])
        loss_d.backward()
        return {"hr": hr, "loss_d": loss_d.detach()}

    def g_backward(self, st: dict, target: torch.Tensor, loss_weight: float) -> None:
        """generator loss against the ALREADY UPDATED discriminator and its backward (L259-268); gradients are left
        un-reduced in the generator's parameters"""
        D, hr = self.D, st["hr"]
        # ---- generator update (L259-269) ----
        self.opt_g.zero_grad(set_to_none=True)
        d_flags = [p.requires_grad for p in D.parameters()]
        for p in D.parameters():
            p.requires_grad_(False)
        try:
            fake = D(hr)
            adv = ops.bce_with_logits(fake, 1.0)
            pix = ops.mse_loss(hr, target)
            ssim_term = ops.ssim_value(hr.detach(), target) if self.compute_ssim else None
            tv = ops.tv_loss(hr, self.tv_weight)
            terms, coefs = [pix, adv, tv], [1.0 - loss_weight, loss_weight, 1.0]
            perc = None
            if self.perceptual is not None:
                perc = self.perceptual(hr, target)
                terms.append(perc)
                coefs.append(1.0)
            loss_g = ops.weighted_sum(coefs, terms)
            loss_g.backward()
        finally:
            for p, f in zip(D.parameters(), d_flags):
                p.requires_grad_(f)
        st.update(loss_g=loss_g.detach(), adv=adv.detach(), pix=pix.detach(), tv=tv.detach(),
                  perc=None if perc is None else perc.detach(), ssim=ssim_term)

    @staticmethod
    def finish(st: dict) -> StepOutput:
        parts = {"adv": st["adv"], "pix": st["pix"], "tv": st["tv"]}
        if st["perc"] is not None:
            parts["perc"] = st["perc"]
        if st["ssim"] is not None:
            parts["ssim"] = st["ssim"]   # SSIM value; the reference logs 1 - SSIM and never uses it
        return StepOutput(st["loss_d"], st["loss_g"], parts, st["hr"].detach())
