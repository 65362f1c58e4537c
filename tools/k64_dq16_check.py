"""K64 backward, DQ16 variant (gd_pam_k64_variant 9) against the production schedule (variant 4) on one small case: dK / dV must be
bit-identical, dQ equal to fp32 round-off (different summation order).  python tools/k64_dq16_check.py"""
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
for var in (4, 9):
    dqn = torch.zeros(B, 32, Np, device=dev); dkn = torch.zeros(B, 32, Np, device=dev); dv = torch.zeros(B, Cp, Np, device=dev)
    K.lib().gd_pam_k64_variant(var, 2)
    K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dv, r_alg=r, c_alg=C, form=0)
    torch.cuda.synchronize()
    res[var] = (dqn.clone(), dkn.clone(), dv.clone())
K.lib().gd_pam_k64_variant(0, 0)
a, b_ = res[4][0][0], res[9][0][0]     # (32 d, N)
print("dk rel", ((res[4][1]-res[9][1]).norm()/res[4][1].norm()).item(), "dv rel", ((res[4][2]-res[9][2]).norm()/res[4][2].norm()).item())
print("dq rel", ((a-b_).norm()/a.norm()).item(), "norms", a.norm().item(), b_.norm().item())
for d in (0, 1, 15, 16, 17, 22, 31):
    print("d", d, "ref", a[d, :6].tolist(), "new", b_[d, :6].tolist())
# per-query pattern
err = (a-b_).abs().sum(0).view(-1, 32)[:4]
print("err by query within tile:", err[0].tolist())
ratio = (b_[:r].flatten() @ a[:r].flatten() / (a[:r].flatten() @ a[:r].flatten())).item()
print("projection ratio", ratio)
