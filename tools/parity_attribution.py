"""Which layer class carries the 16-bit modes' error?  
    python tools/parity_attribution.py > profiles/rNN_parity_attribution.json       (GPU box; oracle = CPU checker)

On the two generator cases of tools/parity_report.py -- the REFERENCE fixture (tests/golden/generator_8ch_16x16.npz) and
the bench-style initialisation against the fp64 oracle (8ch, 32 x 32, B = 2) -- the generator runs
  * in every base mode (fp32 / mixed / bf16 / fp16),
  * in bf16 with ONE layer class switched to exact f32 MFMA ("bf16_but_exact:<class>": what fixing that class buys),
  * in fp32 with ONE layer class switched to 16-bit operands ("fp32_but_16bit:<class>": that class's own contribution).
Every entry: relative L2 error of the output y, the input gradient dx, and the median / worst parameter-gradient error."""
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
CLASSES = ("stem", "dense3x3", "conv1x1", "pam", "cam_apply", "fuse3x3", "decoder")


def summarise(y, y_ref, dx, dx_ref, pg):
    vals = sorted(pg.values())
    worst = max(pg.items(), key=lambda kv: kv[1]) if pg else (None, 0.0)
    return {"y": rell2(y, y_ref), "y_max": relmax(y, y_ref), "dx": rell2(dx, dx_ref),
            "param_grad_median": vals[len(vals) // 2] if vals else 0.0,
            "param_grad_worst": {"name": worst[0], "err": worst[1]}}


def case_fixture():
    fx = load_golden(GOLD, "generator_8ch_16x16")

    def run():
        m = gd.FlexibleUpsamplingModule(input_channels=8)
        fill_module(m)
        m.to(DEV).train()
        x = fx["x"].to(DEV).requires_grad_(True)
        y = m(x)
        y.backward(fx["go"].to(DEV))
        params = dict(m.named_parameters())
        pg = {}
        for k, v in fx.items():
            if k.startswith("grad__") and not k.endswith("_head"):
                name = k[6:].replace("__", ".")
                if not name.endswith("key.bias"):
                    pg[name] = rell2(params[name].grad, v)
        return summarise(y, fx["y"], x.grad, fx["gx"], pg)
    return run


def case_bench_init():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 32, 32, generator=g)
    mo = OM.FlexibleUpsamplingModule(input_channels=8).double()
    torch.manual_seed(11)
    mo.apply(OM.weights_init_normal)
    for n, p in mo.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.1)
    xo = x.double().requires_grad_(True)
    yo = mo.train()(xo)
    go = torch.randn(yo.shape, generator=g)
    yo.backward(go.double())
    sd = {k: v.float() for k, v in mo.state_dict().items()}
    po = dict(mo.named_parameters())

    def run():
        mp = gd.FlexibleUpsamplingModule(input_channels=8)
        mp.load_state_dict(sd)
        mp.to(DEV).train()
        xd = x.to(DEV).requires_grad_(True)
        y = mp(xd)
        y.backward(go.to(DEV))
        pg = {}
        for n, p in mp.named_parameters():
            if p.grad is not None and po[n].grad is not None and not n.endswith("key.bias") and po[n].grad.norm() > 0:
                pg[n] = rell2(p.grad, po[n].grad.float())
        return summarise(y, yo.float(), xd.grad, xo.grad.float(), pg)
    return run


rep = {}
for tag, make in (("generator_8ch_16x16_reference_fixture", case_fixture), ("generator_bench_init_8ch_32x32_vs_fp64_oracle", case_bench_init)):
    run = make()
    res = {}
    for prec in ("fp32", "mixed", "bf16", "fp16"):
        with gd.precision(prec):
            res[prec] = run()
    for c in CLASSES:
        with gd.precision("bf16"), gd.layer_override(**{c: "exact"}):
            res[f"bf16_but_exact:{c}"] = run()
    for c in CLASSES:
        with gd.precision("fp32"), gd.layer_override(**{c: "16"}):
            res[f"fp32_but_16bit:{c}"] = run()
    rep[tag] = res
print(json.dumps(rep, indent=1))
