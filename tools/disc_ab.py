"""Discriminator1 trunk A/B: pixel-major bf16 node (ops.Disc1TrunkFn) vs the fp32-NCHW layer chain.
Errors of both against the reference fixture (tests/golden/disc1_64x64.npz), then fwd+bwd time at the bench shape."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import gan_danet_amd as gd
from gan_danet_amd import ops
from fill import fill_module

dev = torch.device("cuda")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(root, "tests/golden/disc1_64x64.npz")).items()}


def rl2(a, b):
    a, b = a.detach().double().cpu(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


for nhwc in (True, False):
    ops.DISC_NHWC = nhwc
    m = gd.Discriminator1().to(dev)
    x = fx["x"].to(dev).requires_grad_(True)
    with gd.precision("bf16"):
        with torch.no_grad():
            m(x)
        fill_module(m)
        y = m(x)
        y.backward(fx["go"].to(dev))
    errs = {"y": rl2(y, fx["y"]), "dx": rl2(x.grad, fx["gx"])}
    for k, p in m.named_parameters():
        key = "grad__" + k.replace(".", "__")
        if key in fx:
            errs[k] = rl2(p.grad, fx[key])
    print("nhwc" if nhwc else "nchw", {k: f"{v:.2e}" for k, v in errs.items()}, flush=True)

B, H = int(os.environ.get("AB_B", 64)), int(os.environ.get("AB_H", 1024))
m = gd.Discriminator1().to(dev)
with torch.no_grad(), gd.precision("bf16"):
    m(torch.zeros(1, 1, H, H, device=dev))
for need_dx in (False, True):
    for nhwc in (True, False, True, False):
        ops.DISC_NHWC = nhwc
        x = torch.randn(B, 1, H, H, device=dev).requires_grad_(need_dx)
        for p in m.parameters():
            p.requires_grad_(not need_dx)
        ts = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with gd.precision("bf16"):
                y = m(x)
                y.sum().backward()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"B={B} {H}x{H} {'G-step (dx only)' if need_dx else 'D-step (param grads)'} {'nhwc' if nhwc else 'nchw'}: "
              f"{min(ts):.2f} ms", flush=True)
