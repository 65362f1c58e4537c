"""Uninitialised-read hunt: run a module forward + backward twice, the second time after the caching allocator's free blocks
were filled with a poison value; any kernel that reads memory it did not write (padding it assumes zero, a tail it
over-reads) then changes the result.  Differences beyond atomic-order noise are reported per tensor.
    python tools/poison_check.py [fp32|bf16] [poison value]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch

import gan_danet_amd as gd
from gan_danet_amd.generator import DANetAttention
from fill import fill_module

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
poison = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0e4
dev = torch.device("cuda")
g = torch.Generator().manual_seed(3)
x0 = torch.randn(1, 64, 128, 128, generator=g)
go = torch.randn(1, 64, 128, 128, generator=g)
m = DANetAttention(64)
fill_module(m)
m.to(dev).train()


def run():
    for p in m.parameters():
        p.grad = None
    x = x0.to(dev).requires_grad_(True)
    with gd.precision(prec):
        y = m(x)
        y.backward(go.to(dev))
    torch.cuda.synchronize()
    out = {"y": y.detach().clone(), "dx": x.grad.clone()}
    out.update({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    return out


def dirty():
    torch.cuda.synchronize()
    blocks = []
    try:
        for mb in (2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1):
            for _ in range(6):
                blocks.append(torch.full((mb * 262144,), poison, device=dev))
    except RuntimeError:
        pass
    del blocks
    torch.cuda.synchronize()


a = run()
for rnd in range(3):
    dirty()
    b = run()
    worst = max(((b[k] - a[k]).norm() / (a[k].norm() + 1e-30)).item() for k in a)
    bad = {k: f"{((b[k] - a[k]).norm() / (a[k].norm() + 1e-30)).item():.2e}" for k in a
           if not torch.isfinite(b[k]).all() or ((b[k] - a[k]).norm() / (a[k].norm() + 1e-30)).item() > 1e-5}
    print(f"{prec} round {rnd}: worst rel diff {worst:.2e}; beyond 1e-5: {bad}", flush=True)
