"""Prints, for each PAM block of the bench generator's forward, the Cauchy-Schwarz logit bound the max-free forward tests
(|q_i| max_j |k_j| in log2 units; threshold 60 for bf16 operands)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gan_danet_amd as gd
from gan_danet_amd import kern as K

dev = torch.device("cuda")
torch.manual_seed(0)
B, T = int(os.environ.get("PB_B", 4)), int(os.environ.get("PB_T", 256))
G = gd.FlexibleUpsamplingModule(input_channels=8).to(dev)
G.apply(gd.weights_init_normal)
orig = K.pam_flash_fwd


def probe(qt, kt, v, B_, N, Npad, Cn, Cp, gamma, x, out, o_attn, lse, **kw):
    q2 = qt.float().pow(2).sum(2)                        # (B, Npad), pre-scaled by log2 e
    ks = kw.get("k_sqmax")
    kmax = ks.sqrt() if ks is not None else kt.float()[..., :31].pow(2).sum(2).max(1).values.sqrt()
    bound = q2.sqrt().max(1).values * kmax
    # softmax over the keys is invariant under k_j -> k_j - kbar (a per-query shift q_i . kbar): the bound after
    # centring the keys on their mean over the image (round 3: gd_pack_16 row_shift)
    kf = kt.float()[:, :N, :31]
    kc = kf - kf.mean(1, keepdim=True)
    kcmax = kc.pow(2).sum(2).max(1).values.sqrt()
    bound_c = q2.sqrt().max(1).values * kcmax
    # per 256-query workgroup: the share of workgroups whose own bound passes (the kernel votes per workgroup)
    qn = q2.sqrt()[:, :N].reshape(B_, -1, 256).max(2).values * kcmax[:, None]
    print(f"C={Cn}: max|q|(log2 units) {q2.sqrt().max().item():.2f}  max|k| {kmax.max().item():.2f}  bound {bound.max().item():.1f}"
          f"   centred keys: max|k - kbar| {kcmax.max().item():.2f}  bound {bound_c.max().item():.1f}  "
          f"workgroups under 60: {(qn <= 60).float().mean().item():.3f}", flush=True)
    return orig(qt, kt, v, B_, N, Npad, Cn, Cp, gamma, x, out, o_attn, lse, **kw)


K.pam_flash_fwd = probe
with torch.no_grad(), gd.precision("bf16"):
    G(torch.randn(B, 8, T, T, device=dev))
