"""Interleaved A/B timing of the K64 backward schedule variants in ONE process (guide rule 24).
    python tools/pam_ab.py --variants 0:1,0:2,1:2,3:2 --rounds 5        (order:vreg pairs)"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gan_danet_amd import kern as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--tile", type=int, default=256)
ap.add_argument("--channels", type=int, default=184)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--variants", default="0:1,0:2")
ap.add_argument("--form", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda")
B, C, N = a.batch, a.channels, a.tile * a.tile
r = max(1, C // 8)
Np, Cp = (N + 255) // 256 * 256, (C + 31) // 32 * 32
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(B, r, N, device=dev, generator=g) * 0.5
k = torch.randn(B, r, N, device=dev, generator=g) * 0.5
v = torch.randn(B, C, N, device=dev, generator=g)
x = torch.randn(B, C, N, device=dev, generator=g)
do = torch.randn(B, C, N, device=dev, generator=g)
gamma = torch.full((1,), 0.1, device=dev)
ones = Cp - 1 if C < Cp else -1
_, qt = K.pack_bf16(q, r, N, scale_imm=K.LOG2E, t_shape=(Np, 32))
kn, kt = K.pack_bf16(k, r, N, plain_shape=(32, Np), t_shape=(Np, 32), perm16=True, ones_row=31)
vn, vt = K.pack_bf16(v, C, N, plain_shape=(Cp, Np), t_shape=(Np, Cp), perm16=True, ones_row=ones)
out, o = torch.empty_like(x), torch.empty_like(x)
lse = torch.empty(B, N, device=dev)
_, dot_ = K.pack_bf16(do, C, N, scale=gamma, t_shape=(Np, Cp))
dqn = torch.empty(B, 32, Np, device=dev)
dkn = torch.empty(B, 32, Np, device=dev)
dv = torch.empty(B, Cp, Np, device=dev)
K.pam_flash_fwd(qt, kt, vn, B, N, Np, C, Cp, gamma, x, out, o, lse, r_alg=r, v_ones=ones >= 0)
_, delta = K.chan_dot(do, o, gamma)
variants = [tuple(int(t) for t in s.split(":")) for s in a.variants.split(",")]
times = {vv: [] for vv in variants}
ref = None
for rnd in range(a.rounds + 1):
    for vv in variants:
        K.lib().gd_pam_k64_variant(vv[0], vv[1])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dv, r_alg=r, c_alg=C, form=a.form)
        e1.record()
        torch.cuda.synchronize()
        if rnd == 0:     # warm-up round doubles as a cross-variant numerics check
            cur = torch.cat([dqn[:, :r, :N].flatten(), dkn[:, :r, :N].flatten(), dv[:, :C, :N].flatten()])
            if ref is None:
                ref = cur.clone()
            else:
                print(f"variant {vv}: rel diff vs first {((cur - ref).norm() / ref.norm()).item():.2e}", flush=True)
        else:
            times[vv].append(e0.elapsed_time(e1) / a.iters)
K.lib().gd_pam_k64_variant(0, 0)
fl = 4.0 * N * N * (r + C) * B
for vv in variants:
    t = times[vv]
    print(f"order {vv[0]} vreg {vv[1]}: median {statistics.median(t):7.3f} ms  min {min(t):7.3f}  "
          f"{fl / statistics.median(t) / 1e9:7.1f} TF", flush=True)
