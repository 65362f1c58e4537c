"""K64 backward, DQ16 / DV16 variants (gd_pam_k64_variant 9, 11, 12) against the production schedule (variant 4) on one small
case: products left on the 32x32x16 shape must be bit-identical, the others equal to fp32 round-off (other summation order).
python tools/k64_dq16_check.py [variants, default 9,11,12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_danet_amd import kern as K
dev = torch.device("cuda")
B, C, N = 1, 184, 256*8
r = C // 8
Np, Cp = N, 192
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(B, r, N, device=dev, generator=g) * 0.5
k = torch.randn(B, r, N, device=dev, generator=g) * 0.5
v = torch.randn(B, C, N, device=dev, generator=g)
x = torch.randn(B, C, N, device=dev, generator=g)
do = torch.randn(B, C, N, device=dev, generator=g)
gamma = torch.full((1,), 0.1, device=dev)
_, qt = K.pack_bf16(q, r, N, scale_imm=K.LOG2E, t_shape=(Np, 32))
kn, kt = K.pack_bf16(k, r, N, plain_shape=(32, Np), t_shape=(Np, 32), perm16=True, ones_row=31)
vn, vt = K.pack_bf16(v, C, N, plain_shape=(Cp, Np), t_shape=(Np, Cp), perm16=True, ones_row=Cp-1)
out, o = torch.empty_like(x), torch.empty_like(x)
lse = torch.empty(B, N, device=dev)
_, dot_ = K.pack_bf16(do, C, N, scale=gamma, t_shape=(Np, Cp))
K.pam_flash_fwd(qt, kt, vn, B, N, Np, C, Cp, gamma, x, out, o, lse, r_alg=r, v_ones=True)
_, delta = K.chan_dot(do, o, gamma)
res = {}
VARS = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else '9,11,12').split(',')]
for var in [4] + VARS:
    dqn = torch.zeros(B, 32, Np, device=dev); dkn = torch.zeros(B, 32, Np, device=dev); dv = torch.zeros(B, Cp, Np, device=dev)
    K.lib().gd_pam_k64_variant(var, 2)
    K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dv, r_alg=r, c_alg=C, form=0)
    torch.cuda.synchronize()
    res[var] = (dqn.clone(), dkn.clone(), dv.clone())
K.lib().gd_pam_k64_variant(0, 0)
for var in VARS:
    rel = [((res[4][i] - res[var][i]).norm() / res[4][i].norm()).item() for i in range(3)]
    print(f"variant {var}: dq rel {rel[0]:.3e}  dk rel {rel[1]:.3e}  dv rel {rel[2]:.3e}", flush=True)
    if max(rel) > 1e-5:
        a, b_ = res[4][2][0], res[var][2][0]
        print("  dv rows", [(c, a[c, :4].tolist(), b_[c, :4].tolist()) for c in (0, 5, 16, 40)])
        a, b_ = res[4][1][0], res[var][1][0]
        print("  dk rows", [(c, a[c, :4].tolist(), b_[c, :4].tolist()) for c in (0, 5, 16)])
if os.environ.get("K64_ERRMAP"):
    var = int(os.environ["K64_ERRMAP"])
    ek = (res[4][1][0] - res[var][1][0]).abs()[:r]          # (d, key)
    eq = (res[4][0][0] - res[var][0][0]).abs()[:r]          # (d, query)
    print("dk err by key % 64 (sum over d, blocks):", [round(x, 4) for x in ek.sum(0).view(-1, 64).sum(0).tolist()])
    print("dk err by key // 64 :", [round(x, 4) for x in ek.sum(0).view(-1, 64).sum(1).tolist()])
    print("dk err by d:", [round(x, 4) for x in ek.sum(1).tolist()])
    print("dq err by query % 32:", [round(x, 4) for x in eq.sum(0).view(-1, 32).sum(0).tolist()])
    print("dq err by query // 32:", [round(x, 4) for x in eq.sum(0).view(-1, 32).sum(1).tolist()])
