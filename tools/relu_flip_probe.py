"""Locate the ReLU-mask flips behind the run-to-run variation of the config-2 parity case.
    python tools/relu_flip_probe.py [runs] > profiles/rNN_relu_flip_probe.json      (GPU box; oracle = CPU checker)

DANetAttention(64) on 128 x 128 (fill weights, seed 3), fp32 mode, R runs in the default (atomic split-K) mode plus one
deterministic run.  The module's output is the fuse conv's ReLU output, so the mask of a run is y > 0.  For every run
that differs from run 0 in dx by more than 1e-4, the flipped mask elements are listed with the fp64 oracle's
pre-activation z at those positions (relative to mean |z|), and the dx difference is split into the part supported on
the flipped elements' receptive field -- the 3x3 fuse conv makes that a 3x3 pixel window per flipped element for the CAM
half of dx; PAM's value/query paths spread it further, so the check is made one level up: the difference of the two
runs' gradients w.r.t. the fuse conv's INPUT (dfeats) must vanish outside the 3x3 windows."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import gan_danet_amd as gd  # noqa: E402
from gan_danet_amd.generator import DANetAttention  # noqa: E402
from fill import fill_module  # noqa: E402
from oracle import modules as OM  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
g = torch.Generator().manual_seed(3)
x0 = torch.randn(1, 64, 128, 128, generator=g)
go = torch.randn(1, 64, 128, 128, generator=g)
m = DANetAttention(64)
fill_module(m)
mo = OM.DANetAttention(64).double()
mo.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
m.to(dev).train()

# fp64 oracle pre-activation of the fuse conv
with torch.no_grad():
    xo = x0.double()
    feats = torch.cat([mo.position_attention(xo), mo.channel_attention(xo)], 1)
    z = OM._bn(mo.fuse[1].train(), OM._conv(mo.fuse[0], feats))[0]          # (64, 128, 128)
zscale = z.abs().mean().item()


def run():
    for p in m.parameters():
        p.grad = None
    x = x0.to(dev).requires_grad_(True)
    keep = {}

    def hook(mod, inp):
        inp[0].register_hook(lambda gr: keep.__setitem__("dfeats", gr.detach().clone()))
    h = m.fuse.register_forward_pre_hook(hook)
    with gd.precision("fp32"):
        y = m(x)
        y.backward(go.to(dev))
    h.remove()
    torch.cuda.synchronize()
    return {"y": y.detach()[0].cpu(), "dx": x.grad[0].cpu(), "dfeats": keep["dfeats"][0].cpu()}


def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


base = run()
rep = {"case": "DANetAttention(64) on 128x128, fp32 mode, fill weights, seed 3", "runs": runs,
       "oracle_mean_abs_preactivation": zscale, "outliers": [], "dx_rel_diff_vs_run0": []}
events = []
for r in range(1, runs + 1):
    det = r == runs
    if det:
        gd.set_deterministic(True)
    cur = run()
    if det:
        gd.set_deterministic(False)
    d = rel(cur["dx"], base["dx"])
    rep["dx_rel_diff_vs_run0"].append(round(d, 9))
    flips = (cur["y"] > 0) != (base["y"] > 0)
    if d > 1e-4 or flips.any():
        idx = flips.nonzero().tolist()
        # 3x3 windows around the flipped (h, w) positions
        win = torch.zeros(128, 128, dtype=torch.bool)
        for _, hh, ww in idx:
            win[max(0, hh - 1):hh + 2, max(0, ww - 1):ww + 2] = True
        dd = cur["dfeats"] - base["dfeats"]
        inside = dd[:, win].norm().item()
        outside = dd[:, ~win].norm().item()
        events.append({"run": r, "deterministic": det, "dx_rel_diff": d, "flipped_elements": len(idx),
                       "flipped": [{"c_h_w": i, "oracle_z_over_mean_abs_z": z[i[0], i[1], i[2]].item() / zscale,
                                    "run0_y": base["y"][i[0], i[1], i[2]].item(), "this_y": cur["y"][i[0], i[1], i[2]].item()}
                                   for i in idx[:16]],
                       "dfeats_diff_norm_inside_3x3_windows": inside, "dfeats_diff_norm_outside": outside,
                       "dfeats_norm": base["dfeats"].norm().item()})
rep["outliers"] = events
vals = sorted(rep["dx_rel_diff_vs_run0"][:-1])
rep["summary"] = {"median_dx_rel_diff": vals[len(vals) // 2], "max_dx_rel_diff": vals[-1],
                  "runs_with_a_flip": sum(1 for e in events if e["flipped_elements"] and not e["deterministic"]),
                  "deterministic_run_dx_rel_diff": rep["dx_rel_diff_vs_run0"][-1]}
print(json.dumps(rep, indent=1))
