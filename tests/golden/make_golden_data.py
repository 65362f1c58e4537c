"""Golden fixture for SURVEY row f2, generated FROM THE REFERENCE's ``CustomDataset`` (datasets.py:156-208):

    python tests/golden/make_golden_data.py            (build container only: needs /root/reference)

``datasets.py`` is loaded by file path.  Two of its module-level imports are absent from this image and unused by
``CustomDataset`` -- ``NC_READ`` (the reference's NetCDF reader, needs netCDF4) and ``statsmodels.tsa.seasonal.STL`` --
so empty stand-in modules are registered for those NAMES before the import (the recipe SURVEY Appendix A uses for
torchvision; an ordinary ModuleNotFoundError, not a permission denial).  Nothing of the reference is copied: the
fixture holds inputs, the seeds, and the tensors ``CustomDataset.__getitem__`` / ``DataLoader`` returned.
"""
from __future__ import annotations

import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference_datasets():
    for name in ("NC_READ", "statsmodels", "statsmodels.tsa", "statsmodels.tsa.seasonal"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["statsmodels.tsa.seasonal"].STL = None
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("ref_datasets", os.path.join(REF, "datasets.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ds_mod = load_reference_datasets()
    rs = np.random.RandomState(2024)
    n, h, w, c = 6, 8, 8, 3                       # square tiles: apply_augmentation may rotate by 90 degrees
    lr05 = rs.randn(n, h, w).astype(np.float32)
    lr025 = rs.randn(n, 2 * h, 2 * w).astype(np.float32)
    aux = rs.randn(n, 2 * h, 2 * w, c).astype(np.float32)
    out = {"lr_grace_05": lr05, "lr_grace_025": lr025, "hr_aux": aux, "seed_random": np.int64(1234), "seed_torch": np.int64(4321)}

    # ---- augment=False: plain items and the notebook's DataLoader(batch_size=4) batches (L138-142, no shuffle) ----
    ds = ds_mod.CustomDataset(lr05, lr025, aux, augment=False)
    a, b, cc = ds[2]
    out.update(plain_a=a.numpy(), plain_b=b.numpy(), plain_c=cc.numpy(), length=np.int64(len(ds)))
    for i, (ba, bb, bc) in enumerate(torch.utils.data.DataLoader(ds, batch_size=4)):
        out[f"batch{i}_a"], out[f"batch{i}_b"], out[f"batch{i}_c"] = ba.numpy(), bb.numpy(), bc.numpy()

    # ---- augment=True: items 0..n-1 twice over, with python `random` and torch seeded once up front ----
    dsa = ds_mod.CustomDataset(lr05, lr025, aux, augment=True)
    random.seed(1234)
    torch.manual_seed(4321)
    for rep in range(2):
        for i in range(n):
            a, b, cc = dsa[i]
            k = f"aug{rep}_{i}"
            out[k + "_a"], out[k + "_b"], out[k + "_c"] = a.contiguous().numpy(), b.contiguous().numpy(), cc.contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, "customdataset_6x8x8.npz"), **out)
    print("wrote customdataset_6x8x8.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
