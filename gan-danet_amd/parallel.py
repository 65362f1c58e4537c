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

``ShardedParam`` (ZeRO-1 for the one tensor that matters): Discriminator1.fc1 is 2.1e9 parameters at 256x256 tiles --
all-reducing its gradient moves 8.6 GB per rank per step and every rank then streams the same 34 GB of AdamW state.
Sharded, a rank receives only ITS 1/world slice of the summed gradient (reduce-scatter, launched from the same
gradient hook), updates that slice with 1/world of the optimiser state and traffic, and the updated slices are
all-gathered back into the full weight, waited for by a forward pre-hook of the discriminator (the gather runs under
whatever the step does between the D update and the next D forward).  Replicas stay identical: the arithmetic per
element is unchanged.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

# GD_FORCE_COLLECTIVES=1: run every collective of the data-parallel path even in a ONE-rank world.  A one-GPU box cannot
# host two RCCL ranks (RCCL refuses two ranks on one device), so this is how the "nccl" code paths -- reduce_scatter_tensor,
# all_gather_into_tensor, the bucketed all-reduces -- execute on RCCL at all outside an 8-GPU node (tests/test_gpu_ddp.py).
FORCE_COLLECTIVES = os.environ.get("GD_FORCE_COLLECTIVES", "0") == "1"


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


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


def _backend_has_reduce_scatter(group=None) -> bool:
    return dist.get_backend(group) != "gloo"       # gloo: no reduce_scatter / all_gather_into_tensor on all builds


class ShardedParam:
    """One big parameter whose gradient sum, optimiser state and update are split 1/world per rank (module docstring).
    ``numel`` must be a multiple of the world size (Discriminator1.fc1: 1024 x 32 H W)."""

    def __init__(self, p: torch.nn.Parameter, group=None) -> None:
        self.p, self.group = p, group
        self.world, self.rank = world_size(), rank()
        if p.numel() % self.world:
            raise ValueError(f"ShardedParam: {p.numel()} elements do not split over {self.world} ranks")
        self.n = p.numel() // self.world
        self.lo = self.rank * self.n
        self.gshard: Optional[torch.Tensor] = None
        self._rs: list = []
        self._ag: list = []
        self._tmp = None

    # ---- gradient: reduce-scatter (launched from the post-accumulate hook) ----
    def launch_reduce_scatter(self) -> None:
        g = self.p.grad.reshape(-1)
        if _backend_has_reduce_scatter(self.group):
            if self.gshard is None:
                self.gshard = torch.empty(self.n, device=g.device, dtype=g.dtype)
            self._rs = [dist.reduce_scatter_tensor(self.gshard, g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        else:       # gloo (CPU tests / one-GPU rehearsal): one reduce per destination rank, in place on the gradient
            self._rs = [dist.reduce(g[r * self.n:(r + 1) * self.n], dst=r, op=dist.ReduceOp.SUM, group=self.group,
                                    async_op=True) for r in range(self.world)]
            self.gshard = g[self.lo:self.lo + self.n]

    def wait_grad(self) -> torch.Tensor:
        for w in self._rs:
            w.wait()
        self._rs = []
        return self.gshard

    def param_shard(self) -> torch.Tensor:
        return self.p.data.view(-1)[self.lo:self.lo + self.n]

    # ---- parameter: all-gather of the updated slices (waited for by the owning layer's forward pre-hook) ----
    @torch.no_grad()
    def launch_all_gather(self) -> None:
        full = self.p.data.view(-1)
        self._tmp = self.param_shard().clone()          # send buffer (1/world of the tensor)
        if _backend_has_reduce_scatter(self.group):
            self._ag = [dist.all_gather_into_tensor(full, self._tmp, group=self.group, async_op=True)]
        else:
            self._ag = [dist.all_gather([full[r * self.n:(r + 1) * self.n] for r in range(self.world)], self._tmp,
                                        group=self.group, async_op=True)]

    def wait_param(self) -> None:
        for w in self._ag:
            w.wait()
        self._ag, self._tmp = [], None

    # ---- optimiser state of the whole tensor <-> this rank's slice (checkpoints hold FULL tensors) ----
    @torch.no_grad()
    def gather_state(self, shard: torch.Tensor) -> torch.Tensor:
        full = torch.empty(self.n * self.world, device=shard.device, dtype=shard.dtype)
        dist.all_gather([full[r * self.n:(r + 1) * self.n] for r in range(self.world)], shard.contiguous(), group=self.group)
        return full.view_as(self.p)

    def slice_state(self, full: torch.Tensor) -> torch.Tensor:
        return full.reshape(-1)[self.lo:self.lo + self.n].clone()


def shard_big_params(module: torch.nn.Module, min_bytes: int, group=None) -> List[ShardedParam]:
    """ShardedParam for every parameter of ``module`` with at least ``min_bytes`` whose size splits over the ranks.
    A forward pre-hook on ``module`` ITSELF waits for the all-gathers of the updated weights: correct for any forward
    (a hook on the owning sub-module would let the gather run under the layers in front of it, but a forward that reads
    ``sub.weight`` functionally never fires it and would silently compute with stale slices)."""
    out: List[ShardedParam] = []
    if not is_distributed():
        return out
    for p in module.parameters():
        if isinstance(p, torch.nn.parameter.UninitializedParameter):
            continue
        if p.numel() * p.element_size() >= min_bytes and p.numel() % world_size() == 0:
            out.append(ShardedParam(p, group))
    if out:
        module.register_forward_pre_hook(lambda mod, args, sps=tuple(out): [sp.wait_param() for sp in sps] and None)
    return out


class GradReducer:
    """Sum-all-reduce of a parameter list's ``.grad`` tensors with bucketing of the small ones; parameters given as
    ``sharded`` get a reduce-scatter instead (their owner only receives its slice of the sum)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None,
                 sharded: Iterable[ShardedParam] = ()) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params]
        self.bucket_bytes = bucket_bytes
        self.group = group
        self._flat = {}
        self._early = {}            # id(param) -> Work of an all-reduce already in flight for this backward
        self._hooks = []
        self._world = world_size()      # hooks are registered (or not) for THIS world: reduce() checks it still holds
        self._sharded = {id(sp.p): sp for sp in sharded}
        self._rs_launched = set()
        if is_distributed():
            for p in self.params:
                if isinstance(p, torch.nn.parameter.UninitializedParameter):
                    continue
                if id(p) in self._sharded:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._launch_rs))
                elif p.numel() * p.element_size() >= self.bucket_bytes and hasattr(p, "register_post_accumulate_grad_hook"):
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

    def _launch_rs(self, p: torch.nn.Parameter) -> None:
        if p.grad is not None and id(p) not in self._rs_launched:
            self._sharded[id(p)].launch_reduce_scatter()
            self._rs_launched.add(id(p))

    @torch.no_grad()
    def reduce(self) -> None:
        if world_size() != self._world:
            raise RuntimeError(f"GradReducer was built for world size {self._world}, torch.distributed now reports "
                               f"{world_size()}: build it after init_process_group")
        if not is_distributed():
            return
        early_ids = set(self._early.keys())           # big tensors whose all-reduce started during the backward
        works = [self._early.pop(i) for i in early_ids]
        for p in self.params:                         # sharded tensors: reduce-scatter (from the hook, or here)
            if id(p) in self._sharded and p.grad is not None:
                if id(p) not in self._rs_launched:
                    self._sharded[id(p)].launch_reduce_scatter()
                self._sharded[id(p)].wait_grad()
        self._rs_launched.clear()
        grads = [p.grad for p in self.params
                 if p.grad is not None and id(p) not in early_ids and id(p) not in self._sharded]
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
