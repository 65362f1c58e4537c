"""f2: device-resident dataset + augmentation kernel against the CPU restatement of
CustomDataset / DataLoader (oracle/data.py).  Geometry is a pure gather and the noise uses the same CPU torch
stream in "reference" mode, so everything is bit-exact."""
import os
import random

import numpy as np
import pytest
import torch

from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _arrays(n=11, h=16, c=7, seed=0):
    rs = np.random.RandomState(seed)
    return (rs.randn(n, h, h).astype(np.float32), rs.randn(n, 2 * h, 2 * h).astype(np.float32),
            rs.randn(n, 4 * h, 4 * h, c).astype(np.float32))


def test_every_d4_op_word_is_bit_exact():
    from gan_danet_amd import kern as K
    x = torch.randn(3, 5, 12, 12)
    for op in range(16):
        h, v, k = op & 1, (op >> 1) & 1, (op >> 2) & 3
        ref = x
        if h:
            ref = torch.flip(ref, [3])
        if v:
            ref = torch.flip(ref, [2])
        if k:
            ref = torch.rot90(ref, k=k, dims=[2, 3])
        out = K.augment_d4(x.to(DEV), torch.full((3,), op, dtype=torch.int32, device=DEV))
        assert torch.equal(out.cpu(), ref), f"op word {op}"
    # rectangular tiles: flips and the half turn only
    xr = torch.randn(2, 3, 6, 10)
    for op in (0, 1, 2, 3, 8, 9, 10, 11):
        ref = xr
        if op & 1:
            ref = torch.flip(ref, [3])
        if op & 2:
            ref = torch.flip(ref, [2])
        if op >> 2:
            ref = torch.rot90(ref, k=2, dims=[2, 3])
        assert torch.equal(K.augment_d4(xr.to(DEV), torch.full((2,), op, dtype=torch.int32, device=DEV)).cpu(), ref)


@pytest.mark.parametrize("augment", [False, True])
def test_dataset_batches_match_reference_restatement(augment):
    from gan_danet_amd.data import DeviceTileDataset
    from oracle import data as OD
    a, b, c = _arrays()
    ref = OD.CustomDataset(a, b, c, augment=augment)
    random.seed(5)
    torch.manual_seed(5)
    want = list(OD.batches(ref, 4))
    ds = DeviceTileDataset(a, b, c, augment=augment, device=DEV, noise="reference")
    random.seed(5)
    torch.manual_seed(5)
    got = list(ds.batches(4, rank=0, world=1))
    assert len(ds) == len(ref) == 11 and len(got) == len(want) == 3
    for g, w in zip(got, want):
        for tg, tw in zip(g, w):
            assert tuple(tg.shape) == tuple(tw.shape)
            assert torch.equal(tg.cpu(), tw)


def test_rank_shards_partition_each_global_batch():
    """every rank runs the same number of steps with shards of equal size (each step issues collectives, per-shard
    means are averaged over ranks); the ranks' shards of a step are consecutive slices of the global batch"""
    from gan_danet_amd.data import DeviceTileDataset
    a, b, c = _arrays(n=10)
    ds = DeviceTileDataset(a, b, c, device=DEV)
    full = list(ds.batches(8, rank=0, world=1))
    assert [len(fb[0]) for fb in full] == [8, 2]            # world 1: the reference's loader, ragged tail kept
    parts = [list(ds.batches(8, rank=r, world=4)) for r in range(4)]
    assert len({len(p) for p in parts}) == 1                # same step count on every rank
    assert [len(pb[0]) for pb in parts[0]] == [2]           # the 2-sample tail cannot fill 4 ranks: dropped
    for i in range(len(parts[0])):
        for col in range(3):
            got = torch.cat([p[i][col] for p in parts])
            assert torch.equal(got, full[i][col][:len(got)])
    parts2 = [list(ds.batches(8, rank=r, world=2)) for r in range(2)]
    assert [[len(pb[0]) for pb in p] for p in parts2] == [[4, 1], [4, 1]]


def test_loader_batch_feeds_the_trainer_step():
    """dataset batch -> preamble -> one G+D step (the notebook's loop body, L217-272), finite losses"""
    import gan_danet_amd as gd
    from gan_danet_amd.data import DeviceTileDataset
    rs = np.random.RandomState(1)
    ds = DeviceTileDataset(rs.randn(2, 32, 32).astype(np.float32), rs.randn(2, 64, 64).astype(np.float32),
                           rs.randn(2, 64, 64, 7).astype(np.float32), augment=True, device=DEV)
    G, D = gd.FlexibleUpsamplingModule(input_channels=8).to(DEV), gd.Discriminator1().to(DEV)
    with torch.no_grad():
        D(torch.zeros(1, 1, 64, 64, device=DEV))
    G.apply(gd.weights_init_normal), D.apply(gd.weights_init_normal)
    tr = gd.GanTrainer(G, D, None)
    for lr05, lr025, aux in ds.batches(2):
        out = tr.step_from_batch(lr05, lr025, aux, 0.5)
        assert torch.isfinite(out.loss_d).all() and torch.isfinite(out.loss_g).all()


def test_device_dataset_vs_reference_fixture(golden_dir):
    """f2 pin on the PRODUCT path: DeviceTileDataset (gd_augment_d4 gather + reference-stream noise) against what the
    reference's CustomDataset returned for the same seeds (tests/golden/customdataset_6x8x8.npz) -- bit for bit:
    plain items, un-shuffled batches, and twelve augmented items in the reference's order of random draws"""
    import random
    from gan_danet_amd.data import DeviceTileDataset
    fx = np.load(os.path.join(golden_dir, "customdataset_6x8x8.npz"))
    ds = DeviceTileDataset(fx["lr_grace_05"], fx["lr_grace_025"], fx["hr_aux"], augment=False, device=DEV)
    assert len(ds) == int(fx["length"])
    a, b, c = ds[2]
    assert np.array_equal(a.cpu().numpy(), fx["plain_a"]) and np.array_equal(c.cpu().numpy(), fx["plain_c"])
    for i, (ba, bb, bc) in enumerate(ds.batches(4, rank=0, world=1)):
        assert np.array_equal(ba.cpu().numpy(), fx[f"batch{i}_a"]) and np.array_equal(bb.cpu().numpy(), fx[f"batch{i}_b"])
        assert np.array_equal(bc.cpu().numpy(), fx[f"batch{i}_c"])
    dsa = DeviceTileDataset(fx["lr_grace_05"], fx["lr_grace_025"], fx["hr_aux"], augment=True, device=DEV, noise="reference")
    random.seed(int(fx["seed_random"]))
    torch.manual_seed(int(fx["seed_torch"]))
    for rep in range(2):
        for i in range(len(dsa)):
            a, b, c = dsa[i]
            k = f"aug{rep}_{i}"
            assert np.array_equal(a.cpu().numpy(), fx[k + "_a"]), k
            assert np.array_equal(b.cpu().numpy(), fx[k + "_b"]), k
            assert np.array_equal(c.cpu().numpy(), fx[k + "_c"]), k
