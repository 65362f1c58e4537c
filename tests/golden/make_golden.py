"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; the GPU box has
neither it nor this need).  The reference modules are imported by file path
(``models/__init__`` cannot be imported: torchvision is absent), filled with
the closed-form weights of ``fill.py``, run on seeded inputs on CPU in fp32,
and inputs + outputs + gradients are written as ``*.npz`` (data only).

    python tests/golden/make_golden.py

``PerceptualLoss`` cannot be built from the reference here (needs
``torchvision.models.vgg19``) -- no fixture for it ("parity unpinned").
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from fill import fill_module, seeded  # noqa: E402

REF = "/root/reference/models"


def _load(name):
    spec = importlib.util.spec_from_file_location(f"ref_{name}", os.path.join(REF, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    gen, disc, utils = _load("generator"), _load("discriminator"), _load("utils")
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tv.models)
    losses = _load("losses")
    return gen, disc, utils, losses


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def _save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KB")


def _grads(mod, names):
    sd = dict(mod.named_parameters())
    return {"grad__" + n.replace(".", "__"): _np(sd[n].grad) for n in names}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gen, disc, utils, losses = load_reference()

    # ---- PAM / CAM / DANetAttention ------------------------------------------------
    for tag, c, hw, b in (("c32_8x8", 32, 8, 2), ("c160_16x16", 160, 16, 1)):
        x = seeded((b, c, hw, hw), 11).requires_grad_(True)
        go = seeded((b, c, hw, hw), 12)
        m = gen.PAMModule(c)
        fill_module(m)
        with torch.no_grad():
            m.gamma.fill_(0.7)
        y = m(x)
        y.backward(go)
        _save(f"pam_{tag}", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad),
              **_grads(m, ["query.weight", "key.bias", "value.weight", "gamma"]))
        # CAM logits scale with N*|x|^2: scale the input so the softmax is not one-hot
        x = seeded((b, c, hw, hw), 13, 0.25).requires_grad_(True)
        m = gen.CAMModule(c)
        with torch.no_grad():
            m.gamma.fill_(0.3)
        y = m(x)
        y.backward(go)
        _save(f"cam_{tag}", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad), ggamma=_np(m.gamma.grad))

    x = seeded((2, 64, 16, 16), 21, 0.5).requires_grad_(True)
    go = seeded((2, 64, 16, 16), 22)
    m = gen.DANetAttention(64)
    fill_module(m)
    m.train()
    y = m(x)
    y.backward(go)
    _save("danet_c64_16x16", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad),
          rm=_np(m.fuse[1].running_mean), rv=_np(m.fuse[1].running_var),
          **_grads(m, ["fuse.0.weight", "fuse.1.weight", "position_attention.gamma", "channel_attention.gamma"]))

    # ---- DenseBlock -----------------------------------------------------------------
    x = seeded((2, 64, 8, 8), 31).requires_grad_(True)
    go = seeded((2, 160, 8, 8), 32)
    m = gen.DenseBlock(4, 64, 24)
    fill_module(m)
    m.train()
    y = m(x)
    y.backward(go)
    _save("denseblock_64_8x8", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad),
          rm3=_np(m.layers[3].bn.running_mean), rv3=_np(m.layers[3].bn.running_var),
          **_grads(m, ["layers.0.conv.weight", "layers.3.bn.weight", "layers.3.bn.bias", "layers.2.conv.bias"]))

    # ---- Discriminator1 ---------------------------------------------------------------
    x = seeded((2, 1, 64, 64), 41).requires_grad_(True)
    m = disc.Discriminator1()
    with torch.no_grad():
        m(x)  # materialise LazyLinear
    fill_module(m)
    y = m(x)
    go = seeded((2, 1), 42)
    y.backward(go)
    _save("disc1_64x64", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad),
          **_grads(m, ["conv1.weight", "conv4.bias", "fc2.weight"]),
          grad__fc1__weight_head=_np(m.fc1.weight.grad[:8, :64]))

    # ---- losses --------------------------------------------------------------------
    a = seeded((2, 1, 32, 32), 51).requires_grad_(True)
    bimg = seeded((2, 1, 32, 32), 52)
    tv = losses.TVLoss(weight=1e-5)(a)
    (ga_tv,) = torch.autograd.grad(tv, a)
    ss = losses.SSIM(11, True)(a, bimg)
    (ga_ss,) = torch.autograd.grad(ss, a)
    z = seeded((4, 1), 53)
    bce1 = torch.nn.BCEWithLogitsLoss()(z, torch.ones_like(z))
    bce0 = torch.nn.BCEWithLogitsLoss()(z, torch.zeros_like(z))
    ms = torch.nn.MSELoss()(a, bimg)
    _save("losses_32x32", a=_np(a), b=_np(bimg), z=_np(z), tv=_np(tv), gtv=_np(ga_tv), ssim=_np(ss),
          gssim=_np(ga_ss), bce1=_np(bce1), bce0=_np(bce0), mse=_np(ms))

    # ---- full generator ---------------------------------------------------------------
    x = seeded((2, 8, 16, 16), 61).requires_grad_(True)
    go = seeded((2, 1, 64, 64), 62)
    G = gen.FlexibleUpsamplingModule(input_channels=8, attention_type="danet")
    fill_module(G)
    G.train()
    y = G(x)
    y.backward(go)
    names = ["initial.0.weight", "dense_blocks.1.layers.2.conv.weight",
             "attention_modules.0.position_attention.query.weight", "attention_modules.2.channel_attention.gamma",
             "attention_modules.1.position_attention.gamma", "channel_adjust.0.weight", "upsample.4.weight",
             "final.bias", "transition_layers.0.layer.0.weight"]
    _save("generator_8ch_16x16", x=_np(x), go=_np(go), y=_np(y), gx=_np(x.grad),
          rm_up1=_np(G.upsample[1].running_mean), rv_up1=_np(G.upsample[1].running_var), **_grads(G, names))
    G.eval()
    with torch.no_grad():
        ye = G(x)
    _save("generator_8ch_16x16_eval", y=_np(ye))

    # ---- 3-step G+D trajectory (no perceptual term: not constructible here) ---------
    G = gen.FlexibleUpsamplingModule(input_channels=8, attention_type="danet")
    D = disc.Discriminator1()
    tgt = seeded((2, 1, 64, 64), 72)
    with torch.no_grad():
        D(tgt)
    fill_module(G)
    fill_module(D)
    G.train()
    D.train()
    optD = torch.optim.AdamW(D.parameters(), lr=4e-4, betas=(0.5, 0.999), weight_decay=1e-4)
    optG = torch.optim.AdamW(G.parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-4)
    bce, mse_ = torch.nn.BCEWithLogitsLoss(), torch.nn.MSELoss()
    tvl, ssim_ = losses.TVLoss(1e-5), losses.SSIM(11, True)
    xin = seeded((2, 8, 16, 16), 71)
    w = 0.5
    rec = {"loss_d": [], "loss_g": [], "g_norm": [], "d_norm": [], "ssim": []}
    for _ in range(3):
        hr = G(xin)
        optD.zero_grad()
        real, fake = D(tgt), D(hr.detach())
        loss_d = (bce(real, torch.ones_like(real)) + bce(fake, torch.zeros_like(fake))) / 2
        loss_d.backward()
        optD.step()
        optG.zero_grad()
        fake = D(hr)
        adv = bce(fake, torch.ones_like(fake))
        pix = mse_(hr, tgt)
        ss = 1 - ssim_(hr, tgt)
        loss_g = (1 - w) * pix + w * adv + tvl(hr)
        loss_g.backward()
        optG.step()
        rec["loss_d"].append(float(loss_d))
        rec["loss_g"].append(float(loss_g))
        rec["ssim"].append(float(ss))
        rec["g_norm"].append(float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in G.parameters()))))
        rec["d_norm"].append(float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in D.parameters()))))
    _save("trajectory_3steps", x=_np(xin), target=_np(tgt), hr_last=_np(hr),
          final_w=_np(G.final.weight), **{k: np.asarray(v, np.float64) for k, v in rec.items()})

    # ---- structural pin: state_dict keys/shapes ---------------------------------------
    G46 = gen.FlexibleUpsamplingModule(input_channels=46)
    with open(os.path.join(HERE, "generator_state_dict_keys.txt"), "w") as f:
        for k, v in G46.state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")
    D1 = disc.Discriminator1()
    with torch.no_grad():
        D1(torch.zeros(1, 1, 64, 64))
    with open(os.path.join(HERE, "discriminator1_state_dict_keys.txt"), "w") as f:
        for k, v in D1.state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")
    S = disc.SRGAND()
    with open(os.path.join(HERE, "srgand_state_dict_keys.txt"), "w") as f:
        for k, v in S.state_dict().items():
            f.write(f"{k} {tuple(v.shape)}\n")

    # ---- weights_init_normal statistics (RNG-order dependent -> only moments) --------
    torch.manual_seed(123)
    G = gen.FlexibleUpsamplingModule(input_channels=8)
    G.apply(utils.weights_init_normal)
    st = {n: (float(p.mean()), float(p.std())) for n, p in G.named_parameters()
          if n in ("initial.0.weight", "attention_modules.0.fuse.0.weight", "final.weight")}
    with open(os.path.join(HERE, "init_stats.txt"), "w") as f:
        for n, (mu, sd) in st.items():
            f.write(f"{n} {mu:.6e} {sd:.6e}\n")


if __name__ == "__main__":
    main()
