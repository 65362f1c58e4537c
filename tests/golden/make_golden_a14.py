"""Golden fixtures for the SURVEY section-8 rows a14 (SqueezeExcitation, CBAMBlock, SRGAND,
OriginalRelationshipLearner: exported by the reference, optional or unused in the train loop) and f1 (the input
preamble of the step, GAN_DANet_train.ipynb:L218-224), generated FROM THE REFERENCE like ``make_golden.py``:

    python tests/golden/make_golden_a14.py

Only runs in the build container (needs /root/reference).  Data only: inputs, outputs, gradients.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from fill import fill_module, seeded  # noqa: E402
from make_golden import _grads, _np, _save, load_reference  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gen, disc, _, _ = load_reference()

    x = seeded((2, 32, 8, 8), 101).requires_grad_(True)
    go = seeded((2, 32, 8, 8), 102)
    m = gen.SqueezeExcitation(32, reduction_ratio=4)
    fill_module(m)
    y = m(x)
    y.backward(go)
    _save("se_c32_8x8", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad),
          **_grads(m, ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]))

    x = seeded((2, 32, 8, 8), 103).requires_grad_(True)
    m = gen.CBAMBlock(32, reduction_ratio=4)
    fill_module(m)
    y = m(x)
    y.backward(go)
    _save("cbam_c32_8x8", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad),
          **_grads(m, ["spatial_attention.0.weight", "channel_attention.fc1.weight", "channel_attention.fc2.bias"]))

    x = seeded((3, 1, 64, 64), 104).requires_grad_(True)
    m = disc.SRGAND(dim=8)
    fill_module(m)
    m.train()
    y = m(x)
    go1 = seeded((3, 1), 105)
    y.backward(go1)
    _save("srgand_d8_64x64", x=_np(x), go=_np(go1), y=_np(y), gx=_np(x.grad),
          rm1=_np(m.bn1.running_mean), rv1=_np(m.bn1.running_var),
          **_grads(m, ["conv1.weight", "conv6.weight", "bn8.weight", "conv11.weight", "fc.weight"]))

    x = seeded((1, 8, 8, 8), 106).requires_grad_(True)
    m = gen.OriginalRelationshipLearner(8)
    fill_module(m)
    y = m(x)
    go2 = seeded((1, 1024, 8, 8), 107)
    y.backward(go2)
    _save("orl_8ch_8x8", x=_np(x), y_head=_np(y[:, :64]), y_sum=_np(y.sum(dim=1)), gx=_np(x.grad),
          **{"grad__net__0__weight": _np(m.net[0].weight.grad), "grad__net__8__bias": _np(m.net[8].bias.grad)})

    # ---- f1: the loader-loop preamble, exactly the notebook's three lines (L218, L223, L224) ----
    lr05 = seeded((2, 1, 32, 32), 111)
    aux = seeded((2, 7, 64, 64), 112)
    lr = F.interpolate(lr05, scale_factor=0.5, mode="bicubic", align_corners=False)
    down = F.interpolate(aux, scale_factor=0.25, mode="bicubic", align_corners=False)
    _save("preamble_16x16", lr05=_np(lr05), aux=_np(aux), combined=_np(torch.cat([lr, down], dim=1)))


if __name__ == "__main__":
    main()
