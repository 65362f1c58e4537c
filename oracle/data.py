"""CPU restatement of ``CustomDataset`` (datasets.py:156-208) and of the notebook's un-shuffled ``DataLoader``
batching (GAN_DANet_train.ipynb:L130-134).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pinned against the reference itself: ``tests/golden/customdataset_6x8x8.npz`` holds what ``datasets.CustomDataset`` and
``DataLoader`` returned in the build container (``tests/golden/make_golden_data.py``: ``datasets.py`` loaded by path with
stand-ins for its two unused, absent imports), for plain items, batches, and twelve augmented items drawn from seeded
``random`` / ``torch`` streams; ``tests/test_oracle_golden.py`` checks this class against it bit for bit.
"""
from __future__ import annotations

import random

import numpy as np
import torch


class CustomDataset:
    def __init__(self, lr_grace_05, lr_grace_025, hr_aux, augment: bool = False) -> None:   # datasets.py:157-161
        self.lr_grace_05 = torch.from_numpy(np.asarray(lr_grace_05)).float().unsqueeze(1)
        self.lr_grace_025 = torch.from_numpy(np.asarray(lr_grace_025)).float().unsqueeze(1)
        self.hr_aux = torch.from_numpy(np.asarray(hr_aux)).float().permute(0, 3, 1, 2)
        self.augment = augment

    def __len__(self) -> int:
        return len(self.lr_grace_05)

    def __getitem__(self, idx):                                                             # datasets.py:166-174
        a, b, c = self.lr_grace_05[idx], self.lr_grace_025[idx], self.hr_aux[idx]
        if self.augment:
            a, b, c = self.apply_augmentation(a, b, c)
        return a, b, c

    @staticmethod
    def apply_augmentation(a, b, c):                                                        # datasets.py:181-208
        if random.random() > 0.5:
            a, b, c = torch.flip(a, [2]), torch.flip(b, [2]), torch.flip(c, [2])
        if random.random() > 0.5:
            a, b, c = torch.flip(a, [1]), torch.flip(b, [1]), torch.flip(c, [1])
        if random.random() > 0.5:
            k = random.choice([90, 180, 270]) // 90
            a, b, c = torch.rot90(a, k=k, dims=[1, 2]), torch.rot90(b, k=k, dims=[1, 2]), torch.rot90(c, k=k, dims=[1, 2])
        if random.random() > 0.5:
            a = a + torch.randn_like(a) * 0.05
            b = b + torch.randn_like(b) * 0.05
        return a, b, c


def batches(ds: CustomDataset, batch_size: int):
    """DataLoader(ds, batch_size=batch_size): sequential, last partial batch kept, default collate (stack)"""
    for lo in range(0, len(ds), batch_size):
        items = [ds[i] for i in range(lo, min(len(ds), lo + batch_size))]
        yield tuple(torch.stack(col) for col in zip(*items))
