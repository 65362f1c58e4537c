"""helpers shared by the GPU parity tests"""
import os

import numpy as np
import torch

DEV = torch.device("cuda")


def load_golden(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rell2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def assert_close(a, b, tol, what="", metric=relmax):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a.detach().float().cpu()).all(), f"{what}: non-finite values"
    e = metric(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def copy_params(src: torch.nn.Module, dst: torch.nn.Module):
    """state_dict transfer oracle -> product (same keys by construction)"""
    missing, unexpected = dst.load_state_dict(src.state_dict(), strict=True)
    assert not missing and not unexpected
