"""Bitwise run-to-run equality of the pixel-major conv kernels (no atomics in them: any difference is a race)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_danet_amd import kern as K
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return torch.randn(*s, device=dev, generator=g)
for (B, C, H, W, M, stride) in ((2, 64, 32, 32, 128, 2), (2, 64, 64, 64, 128, 2), (4, 128, 16, 16, 256, 2), (2, 64, 32, 32, 64, 1), (2, 256, 8, 8, 512, 2)):
    x = rnd(B, H, W, C).to(torch.bfloat16)
    w = rnd(M, C, 3, 3) * 0.05
    b = rnd(M) * 0.1
    wp = K.conv3x3_nhwc_pack(w, 0)
    ref = None
    bad = 0
    for it in range(40):
        y = K.conv3x3_nhwc_s2(x, wp, b, M, 2, 0.2) if stride == 2 else K.conv3x3_nhwc(x, wp, b, M, relu=True)
        if ref is None: ref = y.clone()
        elif not torch.equal(y.view(torch.int16), ref.view(torch.int16)):
            bad += 1
            d = (y.float() - ref.float()).abs()
            idx = d.nonzero()
            print("  mismatch it", it, "count", idx.shape[0], "first", idx[:4].tolist(), "max", d.max().item())
    print(f"conv s{stride} {C}->{M} @{H}x{W} B{B}: {bad} of 39 repeats differ", flush=True)
    if stride == 2:
        dy = rnd(B, (H + 1) // 2, (W + 1) // 2, M).to(torch.bfloat16)
        ref = None; bad = 0
        K.lib().gd_set_deterministic(1)
        for it in range(40):
            dw, db = K.conv3x3_wgrad_nhwc(dy, x, 2, True)
            if ref is None: ref = dw.clone()
            elif not torch.equal(dw, ref):
                bad += 1
                print("  wgrad mismatch it", it, (dw - ref).abs().max().item(), ref.abs().max().item())
        K.lib().gd_set_deterministic(0)
        print(f"wgrad s2 {C}->{M}: {bad} of 39 repeats differ (deterministic mode: one split)", flush=True)
