"""A few launches of the 184 -> 184 1x1 convolution (PAM value projection) for counter collection:
    rocprofv3 -i tools/pmc_r02.txt --kernel-trace --output-format csv -d <dir> -- python3 tools/conv1x1_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gan_danet_amd import kern as K
from gan_danet_amd import _lib as L

dev = torch.device("cuda")
B, C, H = 8, 184, 256
x = torch.randn(B, C, H, H, device=dev)
w = torch.randn(C, C, 1, 1, device=dev) * 0.05
for _ in range(3):
    y = K.conv2d_fwd(x, w, None, 1, 0, L.PREC_BF16)
torch.cuda.synchronize()
