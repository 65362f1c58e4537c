"""Measured parity of the product path in each operand mode (fp32 / bf16 / fp16), written as JSON.
    python tools/parity_report.py > profiles/rNN_parity_report.json       (GPU box; oracle = CPU checker)

* fixtures generated from the REFERENCE modules (tests/golden/*.npz): generator, DANetAttention, PAM, CAM;
* the fp64 oracle on seeded inputs at BASELINE config 2 (DANetAttention(64) on 128 x 128, N = 16 384) and on the
  bench-style initialisation (weights_init_normal, gamma = 0.1) of the whole generator;
* the 3-step trajectory (hr after three G+D updates).
Every entry is a relative L2 error unless named *_max.  tests/ assert bounds derived from this report."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import gan_danet_amd as gd  # noqa: E402
from fill import fill_module  # noqa: E402
from gpu_util import DEV, load_golden, rell2, relmax  # noqa: E402
from oracle import modules as OM  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
rep = {}


def grads_vs_fixture(mod, fx):
    out = {}
    params = dict(mod.named_parameters())
    for k, v in fx.items():
        if k.startswith("grad__") and not k.endswith("_head"):
            name = k[6:].replace("__", ".")
            if not name.endswith("key.bias"):
                out[name] = rell2(params[name].grad, v)
    return out


def run_fixture(name, build, precs=("fp32", "bf16", "fp16"), gamma=None):
    fx = load_golden(GOLD, name)
    res = {}
    for prec in precs:
        m = build()
        fill_module(m)
        if gamma is not None:      # stand-alone PAM / CAM fixtures: gamma as make_golden.py set it
            with torch.no_grad():
                m.gamma.fill_(gamma)
        m.to(DEV).train()
        x = fx["x"].to(DEV).requires_grad_(True)
        with gd.precision(prec):
            y = m(x)
            y.backward(fx["go"].to(DEV))
        res[prec] = {"y": rell2(y, fx["y"]), "y_max": relmax(y, fx["y"]), "dx": rell2(x.grad, fx["gx"]),
                     "param_grads": grads_vs_fixture(m, fx)}
    rep[name] = res


from gan_danet_amd.generator import CAMModule, DANetAttention, PAMModule  # noqa: E402

run_fixture("generator_8ch_16x16", lambda: gd.FlexibleUpsamplingModule(input_channels=8))
run_fixture("danet_c64_16x16", lambda: DANetAttention(64))
run_fixture("pam_c160_16x16", lambda: PAMModule(160), gamma=0.7)
run_fixture("cam_c160_16x16", lambda: CAMModule(160), gamma=0.3)


def vs_oracle(tag, make_prod, make_orac, shape, init, precs=("fp32", "bf16", "fp16"), seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    mo = make_orac().double()
    init(mo)
    xo = x.double().requires_grad_(True)
    yo = mo.train()(xo)
    go = torch.randn(yo.shape, generator=g)
    yo.backward(go.double())
    res = {}
    for prec in precs:
        mp = make_prod()
        mp.load_state_dict({k: v.float() for k, v in mo.state_dict().items()})
        mp.to(DEV).train()
        xd = x.to(DEV).requires_grad_(True)
        with gd.precision(prec):
            y = mp(xd)
            y.backward(go.to(DEV))
        pg = {}
        po = dict(mo.named_parameters())
        for n, p in mp.named_parameters():
            if p.grad is not None and po[n].grad is not None and not n.endswith("key.bias") and po[n].grad.norm() > 0:
                pg[n] = rell2(p.grad, po[n].grad.float())
        worst = max(pg.items(), key=lambda kv: kv[1]) if pg else (None, 0.0)
        res[prec] = {"y": rell2(y, yo.float()), "y_max": relmax(y, yo.float()), "dx": rell2(xd.grad, xo.grad.float()),
                     "param_grad_worst": {"name": worst[0], "err": worst[1]},
                     "param_grad_median": sorted(pg.values())[len(pg) // 2] if pg else 0.0}
    rep[tag] = res


def bench_init(m):
    torch.manual_seed(11)
    m.apply(OM.weights_init_normal)
    for n, p in m.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.1)


def fill_init(m):
    fill_module(m)


# BASELINE config 2: DANetAttention(64) on a 128 x 128 x 64 feature map (PAM N = 16 384, CAM Gram over 16 384 pixels)
vs_oracle("config2_danet64_128x128_fill", lambda: DANetAttention(64), lambda: OM.DANetAttention(64), (1, 64, 128, 128), fill_init)
vs_oracle("config2_cam64_128x128", lambda: CAMModule(64), lambda: OM.CAMModule(64), (1, 64, 128, 128),
          lambda m: m.gamma.data.fill_(0.3))
# whole generator at the bench's initialisation (weights_init_normal, gamma 0.1), 8 channels, 32 x 32 tiles
vs_oracle("generator_bench_init_8ch_32x32", lambda: gd.FlexibleUpsamplingModule(input_channels=8),
          lambda: OM.FlexibleUpsamplingModule(input_channels=8), (2, 8, 32, 32), bench_init)

# 3-step trajectory
fx = load_golden(GOLD, "trajectory_3steps")
traj = {}
for prec in ("fp32", "bf16"):
    G = gd.FlexibleUpsamplingModule(input_channels=8)
    D = gd.Discriminator1()
    fill_module(G)
    G.to(DEV).train()
    D.to(DEV).train()
    x, tgt = fx["x"].to(DEV), fx["target"].to(DEV)
    with gd.precision(prec):
        with torch.no_grad():
            D(tgt)
        fill_module(D)
        tr = gd.GanTrainer(G, D, perceptual=None)
        rows = []
        for i in range(3):
            out = tr.step(x, tgt, 0.5)
            rows.append({"loss_d_rel": abs(out.loss_d.item() - fx["loss_d"][i].item()) / abs(fx["loss_d"][i].item()),
                         "loss_g_rel": abs(out.loss_g.item() - fx["loss_g"][i].item()) / abs(fx["loss_g"][i].item())})
        traj[prec] = {"steps": rows, "hr_last": rell2(out.hr, fx["hr_last"]), "final_w": rell2(G.final.weight, fx["final_w"])}
rep["trajectory_3steps"] = traj
print(json.dumps(rep, indent=1))
