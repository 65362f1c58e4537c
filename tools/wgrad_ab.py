import os, sys, torch
sys.path.insert(0, "/root/repo")
from gan_danet_amd import kern as K, _lib as L
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
B, Cin, Cout, H = 32, 368, 184, 256
x = torch.randn(B, Cin, H, H, device=dev, generator=g)
dy = torch.randn(B, Cout, H, H, device=dev, generator=g)
res = {}
for mode in (True, False, True, False):
    K.WGRAD_X_NHWC = mode
    dw = K.conv2d_wgrad(dy, x, 3, 1, 1, L.PREC_BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dw = K.conv2d_wgrad(dy, x, 3, 1, 1, L.PREC_BF16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res[mode] = dw.clone()
    print(f"x_nhwc={mode}: {ms:.3f} ms  {2*9*Cin*Cout*H*H*B/ms/1e9:.0f} TF", flush=True)
print("rel diff", ((res[True]-res[False]).norm()/res[False].norm()).item())
