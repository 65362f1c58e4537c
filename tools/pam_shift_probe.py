import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_danet_amd as gd
from gan_danet_amd import kern as K
dev = torch.device("cuda")
torch.manual_seed(1234)
B, T = 8, 256
G = gd.FlexibleUpsamplingModule(input_channels=8).to(dev)
G.apply(gd.weights_init_normal)
for n, p in G.named_parameters():
    if n.endswith("gamma"):
        p.data.fill_(0.1)
orig = K.pam_flash_fwd_shift
def probe(qt, kt, v, B_, N, Npad, Cn, Cp, gamma, x, out, o_attn, lse, **kw):
    for ns in (128, 256, 512):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fl = orig(qt, kt, v, B_, N, Npad, Cn, Cp, gamma, x, out, o_attn, lse, r_alg=kw.get('r_alg', 32), v_ones=kw.get('v_ones', False), nsample=ns, return_flags=True)
        e1.record(); torch.cuda.synchronize()
        print(f"C={Cn} nsample {ns}: {e0.elapsed_time(e1):.2f} ms, flagged workgroups {fl.float().mean().item():.4f}", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.pam_flash_fwd(qt, kt, v, B_, N, Npad, Cn, Cp, gamma, x, out, o_attn, lse, r_alg=kw.get('r_alg', 32), v_ones=kw.get('v_ones', False))
    e1.record(); torch.cuda.synchronize()
    print(f"C={Cn} running max: {e0.elapsed_time(e1):.2f} ms", flush=True)
K.pam_flash_fwd_shift = probe
with torch.no_grad(), gd.precision("bf16"):
    G(torch.randn(B, 8, T, T, device=dev))
