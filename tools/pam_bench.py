"""Micro-benchmark of the fused PAM kernels (forward, backward dK/dV, backward dQ) through the C ABI.
    python tools/pam_bench.py [--batch 4 --tile 256 --channels 184 --iters 3]
Prints per-kernel ms and algorithmic TFLOP/s (2 N^2 (r + C) per image forward; backward 2x)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gan_danet_amd as gd  # noqa: E402
from gan_danet_amd import kern as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--tile", type=int, default=256)
ap.add_argument("--channels", type=int, default=184)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda")
B, C, N = a.batch, a.channels, a.tile * a.tile
r = max(1, C // 8)
Np, Cp = (N + 255) // 256 * 256, (C + 31) // 32 * 32
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(B, r, N, device=dev, generator=g) * float(os.environ.get("PB_QK_SCALE", "0.5"))
k = torch.randn(B, r, N, device=dev, generator=g) * float(os.environ.get("PB_QK_SCALE", "0.5"))
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


def timeit(fn, name, flops):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name:16s} {ms:9.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s (algorithmic)", flush=True)


fl = 2.0 * N * N * (r + C) * B
from gan_danet_amd import _lib as L  # noqa: E402

timeit(lambda: K.pam_flash_fwd(qt, kt, vn, B, N, Np, C, Cp, gamma, x, out, o, lse, r_alg=r, v_ones=ones >= 0), "fwd", fl)
ksq = K.pam_key_sqnorm_max(kt, N)
timeit(lambda: K.pam_flash_fwd(qt, kt, vn, B, N, Np, C, Cp, gamma, x, out, o, lse, r_alg=r, v_ones=ones >= 0, k_sqmax=ksq),
       "fwd, max-free", fl)
d_raw, delta = K.chan_dot(do, o, gamma)
forms = [int(f) for f in os.environ.get("PAM_BENCH_FORMS", "0,1,2,3").split(",")]
names = {0: "bwd k64 atomic", 1: "bwd k64 parts", 2: "bwd k32 parts", 3: "bwd 2 kernels"}
ref = None
for form in forms:
    timeit(lambda: K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dv, r_alg=r, c_alg=C, form=form),
           names[form], 2 * fl)
    cur = (dqn[:, :r, :N].clone(), dkn[:, :r, :N].clone(), dv[:, :C, :N].clone())
    if ref is None:
        ref = cur
    else:
        print("    vs first form: " + "  ".join(
            f"{n} {((a - b).norm() / b.norm()).item():.2e}" for n, a, b in zip(("dq", "dk", "dv"), cur, ref)), flush=True)
if os.environ.get("PAM_BENCH_F16", "1") != "0":
    # fp16 operand mode (BASELINE config 5): same inputs packed as IEEE fp16
    _, qt = K.pack_bf16(q, r, N, scale_imm=K.LOG2E, t_shape=(Np, 32), f16=True)
    kn, kt = K.pack_bf16(k, r, N, plain_shape=(32, Np), t_shape=(Np, 32), perm16=True, ones_row=31, f16=True)
    vn, vt = K.pack_bf16(v, C, N, plain_shape=(Cp, Np), t_shape=(Np, Cp), perm16=True, ones_row=ones, f16=True)
    _, dot_ = K.pack_bf16(do, C, N, scale=gamma, t_shape=(Np, Cp), f16=True)
    timeit(lambda: K.pam_flash_fwd(qt, kt, vn, B, N, Np, C, Cp, gamma, x, out, o, lse, r_alg=r, v_ones=ones >= 0, f16=True),
           "fwd fp16", fl)
    d_raw, delta = K.chan_dot(do, o, gamma)
    timeit(lambda: K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dv, r_alg=r, c_alg=C, f16=True,
                                   form=0), "bwd fp16 k64", 2 * fl)
    cur = (dqn[:, :r, :N].clone(), dkn[:, :r, :N].clone(), dv[:, :C, :N].clone())
    print("    fp16 vs bf16 first form: " + "  ".join(
        f"{n} {((a - b).norm() / b.norm()).item():.2e}" for n, a, b in zip(("dq", "dk", "dv"), cur, ref)), flush=True)
