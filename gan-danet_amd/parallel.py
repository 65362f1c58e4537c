"""Data parallelism for the G+D step: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI; "gloo" for the CPU tests), the minibatch sharded by rank, full replicas of G and D, and a
summing all-reduce of the discriminator gradients after ``loss_D.backward()`` and of the generator
gradients after ``loss_G.backward()`` (the two points where the single-device loop of
GAN_DANet_train.ipynb:L255 and L268 hands gradients to the optimiser).  The 1/world factor is applied
by the optimiser as it reads the gradient (``AdamW.grad_scale``), so no extra pass over the 2-billion
element fc1 gradient.  BatchNorm uses per-replica batch statistics (DDP semantics).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large messages beat many small ones, so
small gradients are packed into flat buckets of ``bucket_bytes`` and big tensors (Discriminator1.fc1)
are reduced in place.  A big tensor's all-reduce is launched THE MOMENT autograd has finished its gradient
(post-accumulate hook): fc1 sits at the end of the discriminator, so its 8.6 GB gradient is ready first and
travels under the backward of the four conv layers; everything is waited on together in ``reduce()``.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


@torch.no_grad()
def broadcast_module(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """make every replica start from rank ``src``'s parameters and buffers"""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if isinstance(t, torch.nn.parameter.UninitializedParameter):
            raise RuntimeError("materialise lazy parameters (one dummy forward) before broadcast_module")
        dist.broadcast(t.data, src=src, group=group)


def shard_batch(global_batch: int, world: Optional[int] = None, rk: Optional[int] = None) -> slice:
    """contiguous shard of the global minibatch owned by this rank (remainder spread over the first ranks)"""
    world = world_size() if world is None else world
    rk = rank() if rk is None else rk
    base, rem = divmod(global_batch, world)
    lo = rk * base + min(rk, rem)
    return slice(lo, lo + base + (1 if rk < rem else 0))


class GradReducer:
    """Sum-all-reduce of a parameter list's ``.grad`` tensors with bucketing of the small ones."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params]
        self.bucket_bytes = bucket_bytes
        self.group = group
        self._flat = {}
        self._early = {}            # id(param) -> Work of an all-reduce already in flight for this backward
        self._hooks = []
        self._world = world_size()      # hooks are registered (or not) for THIS world: reduce() checks it still holds
        if is_distributed():
            for p in self.params:
                if isinstance(p, torch.nn.parameter.UninitializedParameter):
                    continue
                if p.numel() * p.element_size() >= self.bucket_bytes and hasattr(p, "register_post_accumulate_grad_hook"):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._launch_early))

    def close(self) -> None:
        """remove the gradient hooks (a parameter must be watched by ONE reducer at a time)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._early.clear()

    def _launch_early(self, p: torch.nn.Parameter) -> None:
        if p.grad is not None and id(p) not in self._early:
            self._early[id(p)] = dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @torch.no_grad()
    def reduce(self) -> None:
        if world_size() != self._world:
            raise RuntimeError(f"GradReducer was built for world size {self._world}, torch.distributed now reports "
                               f"{world_size()}: build it after init_process_group")
        if not is_distributed():
            return
        early_ids = set(self._early.keys())           # big tensors whose all-reduce started during the backward
        works = [self._early.pop(i) for i in early_ids]
        grads = [p.grad for p in self.params if p.grad is not None and id(p) not in early_ids]
        buckets = []
        cur, cur_bytes = [], 0
        for g in grads:
            nbytes = g.numel() * g.element_size()
            if nbytes >= self.bucket_bytes:
                works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                continue
            if cur and cur_bytes + nbytes > self.bucket_bytes:
                buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(g)
            cur_bytes += nbytes
        if cur:
            buckets.append(cur)
        flats = []
        for i, bucket in enumerate(buckets):
            n = sum(g.numel() for g in bucket)
            flat = self._flat.get((i, n))
            if flat is None or flat.device != bucket[0].device:
                flat = torch.empty(n, device=bucket[0].device, dtype=bucket[0].dtype)
                self._flat[(i, n)] = flat
            off = 0
            for g in bucket:   # pack (device-to-device copies; plumbing)
                flat[off:off + g.numel()].copy_(g.reshape(-1))
                off += g.numel()
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            flats.append((flat, bucket))
        for w in works:
            w.wait()
        for flat, bucket in flats:
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
