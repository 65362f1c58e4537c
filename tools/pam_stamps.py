"""Cycle anatomy of the K64 PAM backward's tile loop from in-kernel s_memtime stamps (diagnostic variants 5 / 6 of
gd_pam_k64_variant; the production kernel carries no stamp).
    python tools/pam_stamps.py [--batch 4] > profiles/rNN_k64_stamps.txt
Per wave the kernel sums, over its sweep of the query tiles, the cycles between the segment seams:
  wait+barrier | head (row constants, Q rows, S MFMAs issued) | S + dP phase | dS arithmetic | dV^T steps | dK^T + dQ^T steps + exchange write
Reported: mean cycles per 32-query tile and wave, per segment, for the production schedule (variant 5) and the hand-placed
second half (variant 6), beside the MFMA cycles each segment holds (32 per v_mfma_f32_32x32x16)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gan_danet_amd import kern as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--tile", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda")
B, C, N = a.batch, 184, a.tile * a.tile
r = C // 8
Np, Cp = (N + 255) // 256 * 256, 192
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(B, r, N, device=dev, generator=g) * 0.5
k = torch.randn(B, r, N, device=dev, generator=g) * 0.5
v = torch.randn(B, C, N, device=dev, generator=g)
x = torch.randn(B, C, N, device=dev, generator=g)
do = torch.randn(B, C, N, device=dev, generator=g)
gamma = torch.full((1,), 0.1, device=dev)
_, qt = K.pack_bf16(q, r, N, scale_imm=K.LOG2E, t_shape=(Np, 32))
kn, kt = K.pack_bf16(k, r, N, plain_shape=(32, Np), t_shape=(Np, 32), perm16=True, ones_row=31)
vn, vt = K.pack_bf16(v, C, N, plain_shape=(Cp, Np), t_shape=(Np, Cp), perm16=True, ones_row=Cp - 1)
out, o = torch.empty_like(x), torch.empty_like(x)
lse = torch.empty(B, N, device=dev)
_, dot_ = K.pack_bf16(do, C, N, scale=gamma, t_shape=(Np, Cp))
dqn = torch.empty(B, 32, Np, device=dev)
dkn = torch.empty(B, 32, Np, device=dev)
dv = torch.empty(B, Cp, Np, device=dev)
K.pam_flash_fwd(qt, kt, vn, B, N, Np, C, Cp, gamma, x, out, o, lse, r_alg=r, v_ones=True)
_, delta = K.chan_dot(do, o, gamma)
names = ["wait+barrier", "head", "S+dP", "dS", "dV^T", "dK^T+dQ^T"]
mfma = {5: [0, 4, 24, 0, 24, 8], 6: [0, 4, 24, 0, 24, 8], 10: [0, 4, 28, 0, 24, 4]}
for variant in [int(v) for v in os.environ.get("STAMP_VARIANTS", "5,6").split(",")]:
    dbg = torch.zeros(B * (Np // 256) * 4 * 12, device=dev, dtype=torch.int32)
    K.lib().gd_pam_k64_debug(dbg.data_ptr())
    K.lib().gd_pam_k64_variant(variant, 2)
    for _ in range(2):
        K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dv, r_alg=r, c_alg=C, form=0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dv, r_alg=r, c_alg=C, form=0)
    e1.record()
    torch.cuda.synchronize()
    K.lib().gd_pam_k64_variant(0, 0)
    K.lib().gd_pam_k64_debug(None)
    d = dbg.view(-1, 12).double().cpu()
    tiles = d[:, 7]
    per = d[:, :6] / tiles[:, None]
    mean = per.mean(0)
    tot = mean.sum().item()
    print(f"variant {variant} ({ {5: 'production schedule', 6: 'hand-placed second half', 10: 'DQ16: dQ sub-tiles, no exchange'}[variant] } + stamps): "
          f"{e0.elapsed_time(e1):.2f} ms, {tot:.0f} cycles per tile and wave (MFMA: 60 x 32 = 1920)")
    for n, c, m in zip(names, mean.tolist(), mfma[variant]):
        print(f"    {n:14s} {c:8.1f} cycles  ({100 * c / tot:5.1f} %)   MFMA cycles issued in it: {32 * m}")
    w = per.view(-1, 4, 6).mean(0)
    print("    by wave (wait+barrier): " + "  ".join(f"{w[i, 0].item():.0f}" for i in range(4)))
    clk = (d[:, 8] / d[:, 9].clamp(min=1)).median().item() * 100.0          # s_memrealtime ticks at 100 MHz
    print(f"    in-kernel clock over the sweep (median wave): {clk / 1000:.3f} GHz")
    wq = (d[:, 6] / tiles).view(-1, 4).mean(0)
    print("      of which loop back + counted vmcnt / lgkmcnt waits (before the barrier), by wave: " + "  ".join(f"{v.item():.0f}" for v in wq))
