"""Device-resident counterpart of ``CustomDataset`` + ``DataLoader`` (datasets.py:156-208,
GAN_DANet_train.ipynb:L130-134): the three arrays live in HBM, a batch is a slice (the notebook's loaders do not
shuffle), the optional augmentation of a batch is one HIP gather per array, and under data parallelism every
rank takes its contiguous share of each global batch.
"""
from __future__ import annotations

import random
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import kern as K
from .parallel import rank as _rank
from .parallel import world_size as _world

OP_HFLIP, OP_VFLIP, OP_NOISE = 1, 2, 16


def draw_augmentation(rng=random) -> int:
    """the random decisions of ``apply_augmentation`` for ONE sample, in the reference's order of draws
    (datasets.py:183-205), as an op word (include/gandanet.h, gd_augment_d4)"""
    op = 0
    if rng.random() > 0.5:
        op |= OP_HFLIP
    if rng.random() > 0.5:
        op |= OP_VFLIP
    if rng.random() > 0.5:
        op |= (rng.choice([90, 180, 270]) // 90) << 2
    if rng.random() > 0.5:
        op |= OP_NOISE
    return op


class DeviceTileDataset:
    """``CustomDataset(lr_grace_05, lr_grace_025, hr_aux, augment)`` with the tensors on ``device``.

    ``lr_grace_05`` (N, h, w), ``lr_grace_025`` (N, H, W), ``hr_aux`` (N, H, W, C) as in the reference (numpy or
    tensors); stored as (N, 1, h, w), (N, 1, H, W), (N, C, H, W) fp32 (datasets.py:158-160).
    ``noise``: "reference" draws the Gaussian noise with torch's CPU generator exactly where the reference does
    (bit-identical stream, host-bound); "device" draws it on the GPU (same distribution, different stream)."""

    def __init__(self, lr_grace_05, lr_grace_025, hr_aux, augment: bool = False, device=None, noise: str = "device"):
        device = torch.device("cuda") if device is None else torch.device(device)
        if device.type != "cuda":
            raise K.L.GandanetError("DeviceTileDataset: tensors live on the GPU (there is no CPU path)")
        as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        self.lr_grace_05 = as_t(lr_grace_05).float().unsqueeze(1).contiguous().to(device)
        self.lr_grace_025 = as_t(lr_grace_025).float().unsqueeze(1).contiguous().to(device)
        self.hr_aux = as_t(hr_aux).float().to(device).permute(0, 3, 1, 2).contiguous()
        if not (len(self.lr_grace_05) == len(self.lr_grace_025) == len(self.hr_aux)):
            raise ValueError("the three arrays must hold the same number of samples")
        if noise not in ("device", "reference"):
            raise ValueError("noise must be 'device' or 'reference'")
        self.augment, self.noise, self.device = augment, noise, device

    def __len__(self) -> int:
        return len(self.lr_grace_05)

    def _augment(self, a, b, c, ops_list: List[int]):
        ops = torch.tensor(ops_list, dtype=torch.int32, device=self.device)
        if any((op >> 2) & 1 for op in ops_list):
            for t in (a, b, c):
                if t.shape[2] != t.shape[3]:
                    raise K.L.GandanetError("a quarter turn needs square tiles")
        na = nb = None
        if any(op & OP_NOISE for op in ops_list):
            if self.noise == "reference":      # per sample, lr_grace_05 first then lr_grace_025 (datasets.py:201-204)
                na, nb = torch.zeros(a.shape), torch.zeros(b.shape)
                # rot90 by an odd number of quarter turns returns a transposed-stride tensor, and randn_like on a
                # non-contiguous tensor consumes the generator differently (strided scalar path): draw on a
                # tensor of the same strides
                draw = lambda t, odd: (torch.randn_like(torch.empty(t.shape[1], t.shape[3], t.shape[2]).transpose(1, 2))
                                       if odd else torch.randn(t.shape[1:]))
                for i, op in enumerate(ops_list):
                    if op & OP_NOISE:
                        na[i] = draw(a, (op >> 2) & 1)
                        nb[i] = draw(b, (op >> 2) & 1)
                na, nb = na.to(self.device), nb.to(self.device)
            else:
                na = torch.randn(a.shape, device=self.device)
                nb = torch.randn(b.shape, device=self.device)
        return K.augment_d4(a, ops, na, 0.05), K.augment_d4(b, ops, nb, 0.05), K.augment_d4(c, ops)

    def get(self, lo: int, hi: int, ops_list: Optional[List[int]] = None):
        """samples [lo, hi) as one batch; ``ops_list`` overrides the random draws (tests)"""
        a, b, c = self.lr_grace_05[lo:hi], self.lr_grace_025[lo:hi], self.hr_aux[lo:hi]
        if self.augment or ops_list is not None:
            if ops_list is None:
                ops_list = [draw_augmentation() for _ in range(hi - lo)]
            a, b, c = self._augment(a, b, c, ops_list)
        return a, b, c

    def __getitem__(self, idx: int):
        a, b, c = self.get(idx, idx + 1)
        return a[0], b[0], c[0]

    def batches(self, batch_size: int, rank: Optional[int] = None, world: Optional[int] = None
                ) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """``DataLoader(dataset, batch_size=batch_size)`` (no shuffle, last batch kept, L133); with ``world`` > 1
        ``batch_size`` is the GLOBAL batch and each rank gets its contiguous slice of it.

        Under ``world`` > 1 every rank must run the SAME number of steps (each step issues collectives) and hold
        shards of EQUAL size (per-shard means are averaged over ranks): a global batch is cut into ``world`` equal
        shards and the ``len % world`` samples of a ragged last batch that do not fill one more round are dropped
        (a last batch smaller than ``world`` is dropped whole).  ``batch_plan`` returns the same slices."""
        rank = _rank() if rank is None else rank
        world = _world() if world is None else world
        for lo, hi in self.batch_plan(batch_size, rank, world):
            yield self.get(lo, hi)

    def batch_plan(self, batch_size: int, rank: int, world: int) -> List[Tuple[int, int]]:
        """[lo, hi) sample ranges of ``batches`` for one rank: one entry per step, equal length on every rank"""
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"rank {rank} / world {world}")
        if world > 1 and batch_size % world:
            raise ValueError(f"global batch {batch_size} is not a multiple of the world size {world}")
        plan, n = [], len(self)
        for g0 in range(0, n, batch_size):
            g1 = min(n, g0 + batch_size)
            per = (g1 - g0) // world            # world == 1: the whole (possibly ragged) batch, as the reference
            if per > 0:
                plan.append((g0 + rank * per, g0 + (rank + 1) * per))
        return plan
