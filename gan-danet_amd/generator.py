"""GAN-DANet generator on HIP kernels: the reference's module tree (class names, constructor signatures,
attribute names == ``state_dict`` keys of models/generator.py) with MI355X-first forwards:

* DenseBlock        one autograd node over a pre-allocated channel slab (no torch.cat); each layer's
                    BatchNorm+ReLU is folded into its conv's operand load           (generator.py:29-54)
* DANetAttention    PAM (fused flash attention, bf16 MFMA) and CAM (split-K Gram + row softmax + apply)
                    write the two halves of one 2C buffer, then conv3x3+BN+ReLU     (generator.py:104-157)
* TransitionLayer   BN+ReLU folded into the 1x1 conv                              (generator.py:57-67)
* skip connections  the bias-free 1x1 ``channel_adjust`` convs commute with the bilinear resize, so they
                    run at H x W instead of 4H x 4W and the three skips share one upsample-add
                                                                                    (generator.py:243-245)
"""
from __future__ import annotations

import warnings
from typing import List, Optional

import torch
from torch import nn

from . import ops
from .layers import ACT_RELU, BatchNorm2d, Conv2d, ReLU, Sigmoid, UpsampleBicubic2x


class _ConvBnRelu(nn.Sequential):
    """[Conv2d, BatchNorm2d, ReLU] with the reference's indices; BN+ReLU run as one kernel."""

    def __init__(self, cin: int, cout: int, k: int, pad: int, layer_class: str = "other") -> None:
        super().__init__(Conv2d(cin, cout, kernel_size=k, padding=pad, bias=False), BatchNorm2d(cout), ReLU())
        self[0].layer_class = layer_class

    def forward(self, x):
        return self[1](self[0](x), ACT_RELU)


class DenseLayer(nn.Module):
    def __init__(self, in_channels: int, growth_rate: int) -> None:
        super().__init__()
        self.bn = BatchNorm2d(in_channels)
        self.relu = ReLU()
        self.conv = Conv2d(in_channels, growth_rate, kernel_size=3, padding=1)

    def forward(self, x):  # stand-alone use; inside DenseBlock the slab path below is taken
        return DenseBlock.run([self], x)


class DenseBlock(nn.Module):
    def __init__(self, num_layers: int, in_channels: int, growth_rate: int) -> None:
        super().__init__()
        self.layers = nn.ModuleList(
            DenseLayer(in_channels + i * growth_rate, growth_rate) for i in range(num_layers))

    @staticmethod
    def run(layers, x):
        flat = []
        training = momentum = eps = None
        for lyr in layers:
            g, b, rm, rv, training, momentum, eps = lyr.bn.fold_args()
            flat += [g, b, rm, rv, lyr.conv.weight, lyr.conv.bias]
        return ops.DenseBlockFn.apply(x, training, momentum, eps, *flat)

    def forward(self, x):
        return DenseBlock.run(list(self.layers), x)


class TransitionLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.layer = nn.Sequential(BatchNorm2d(in_channels), ReLU(), Conv2d(in_channels, out_channels, kernel_size=1))

    def forward(self, x):
        g, b, rm, rv, training, momentum, eps = self.layer[0].fold_args()
        conv = self.layer[2]
        return ops.BnReluConvFn.apply(x, g, b, rm, rv, conv.weight, conv.bias, training, momentum, eps, 0)


class PAMModule(nn.Module):
    """Position attention (generator.py:104-122).  """

    def __init__(self, channels: int) -> None:
        super().__init__()
        r = max(1, channels // 8)
        self.query = Conv2d(channels, r, kernel_size=1)
        self.key = Conv2d(channels, r, kernel_size=1)
        self.value = Conv2d(channels, channels, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return ops.PamFn.apply(x, self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                               self.value.weight, self.value.bias, self.gamma)


class CAMModule(nn.Module):
    """Channel attention (generator.py:125-139); ``channels`` is unused there as well."""

    def __init__(self, channels: int) -> None:
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return ops.CamFn.apply(x, self.gamma)


class DANetAttention(nn.Module):
    def __init__(self, channels: int) -> None:
        super().__init__()
        self.position_attention = PAMModule(channels)
        self.channel_attention = CAMModule(channels)
        self.fuse = _ConvBnRelu(2 * channels, channels, 3, 1, "fuse3x3")

    def forward(self, x):
        pa = self.position_attention
        feats = ops.DualAttentionFn.apply(x, pa.query.weight, pa.query.bias, pa.key.weight, pa.key.bias,
                                          pa.value.weight, pa.value.bias, pa.gamma, self.channel_attention.gamma)
        return self.fuse(feats)


def _build_attention(attention_type: Optional[str], channels: int) -> Optional[nn.Module]:
    """generator.py:160-172.  'senet'/'cbam' alias to 'danet' (the reference raises NameError there because
    ``warnings`` is never imported; the aliasing it intends is kept)."""
    if attention_type is None or attention_type.lower() == "none":
        return None
    kind = attention_type.lower()
    if kind in ("senet", "cbam"):
        warnings.warn(f"Attention type '{attention_type}' currently aliases to 'danet'.", RuntimeWarning)
    elif kind != "danet":
        raise ValueError(f"Unsupported attention type: {attention_type}")
    return DANetAttention(channels)


class FlexibleUpsamplingModule(nn.Module):
    """Super-resolution generator (x4) of GAN-DANet (generator.py:175-247)."""

    def __init__(self, input_channels: int = 40, growth_rate: int = 24, num_blocks: int = 3,
                 num_layers_per_block: int = 4, attention_type: Optional[str] = "danet") -> None:
        super().__init__()
        self.initial = _ConvBnRelu(input_channels, 64, 3, 1, "stem")
        self.dense_blocks = nn.ModuleList()
        self.transition_layers = nn.ModuleList()
        self.attention_modules = nn.ModuleList()
        self.feature_channels: List[int] = []
        width = 64
        for i in range(num_blocks):
            self.dense_blocks.append(DenseBlock(num_layers_per_block, width, growth_rate))
            width += num_layers_per_block * growth_rate
            self.attention_modules.append(_build_attention(attention_type, width))
            self.feature_channels.append(width)
            if i + 1 < num_blocks:
                self.transition_layers.append(TransitionLayer(width, width // 2))
                width //= 2
        self.channel_adjust = nn.ModuleList(
            Conv2d(ch, 64, kernel_size=1, bias=False) for ch in self.feature_channels[::-1])
        self.upsample = nn.Sequential(
            Conv2d(width, 64, kernel_size=3, padding=1, bias=False), BatchNorm2d(64), ReLU(), UpsampleBicubic2x(),
            Conv2d(64, 64, kernel_size=3, padding=1, bias=False), BatchNorm2d(64), ReLU(), UpsampleBicubic2x())
        self.final = Conv2d(64, 1, kernel_size=3, padding=1)
        for m in (self.upsample[0], self.upsample[4], self.final):
            m.layer_class = "decoder"
        for m in self.channel_adjust:
            m.layer_class = "conv1x1"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.initial(x)
        skips = []
        for i, block in enumerate(self.dense_blocks):
            x = block(x)
            att = self.attention_modules[i]
            if att is not None:
                x = att(x)
            skips.append(x)
            if i < len(self.transition_layers):
                x = self.transition_layers[i](x)
        up = self.upsample
        x = up[3](up[1](up[0](x), ACT_RELU))
        c = up[5](up[4](x), ACT_RELU)                      # (B, 64, 2H, 2W)
        fw = self.final.weight
        if fw.shape[0] != 1 or tuple(fw.shape[2:]) != (3, 3):
            x = ops.SkipFuseFn.apply(up[7](c), *[adj.weight for adj in self.channel_adjust], *skips[::-1])
            return self.final(x)
        # Tail as written (generator.py:242-247): final(up2(c) + sum_k adjust_k(resize4(f_k))).  Every map after the
        # last ReLU is linear and the resizes act per channel, so the channel contraction of `final` is pulled in
        # front of them: nine "tap planes" t_tap = sum_ch w[ch][tap] c_ch (a 1x1 conv 64 -> 9 at 2H x 2W) and
        # sum_ch (w[.][tap] A_k)[ch] f_k,ch (1x1 convs C_k -> 9 at H x W, final and channel_adjust weights composed)
        # are resized and summed with their tap shifts.  The 4H x 4W stage carries 9 planes instead of 64 channels
        # (7x less HBM traffic there, no 64-channel 4H x 4W tensor kept for backward).  Same parameters, same
        # result up to fp32 re-association.
        w9 = fw.reshape(fw.shape[1], 9).t()                # (9, 64): w9[tap][ch]
        t = ops.conv2d(c, w9.reshape(9, -1, 1, 1).contiguous(), layer="conv1x1")   # .t()/.reshape(): views + one layout copy
        u = up[7](t)                                       # (B, 9, 4H, 4W)
        # composed operators (9 x C_k) = w9 (9 x 64) . A_k (64 x C_k): a 1x1 conv over A_k read as a 64-channel,
        # C_k-pixel image (exact-fp32 MFMA kernel whatever the configured precision: these are weights)
        w9c = w9.reshape(9, -1, 1, 1).contiguous()
        wk = [ops.conv2d(adj.weight.reshape(1, adj.weight.shape[0], 1, -1), w9c, prec=ops.L.PREC_FP32).reshape(9, -1, 1, 1)
              for adj in self.channel_adjust]
        u = ops.SkipFuseFn.apply(u, *wk, *skips[::-1])
        return ops.ShiftSum9Fn.apply(u, self.final.bias)


# ---- exported by the reference but not used by the train loop: compositions of the same kernels --------
class OriginalRelationshipLearner(nn.Module):
    """generator.py:11-26: five conv3x3+ReLU."""

    def __init__(self, input_channels: int) -> None:
        super().__init__()
        mods, cin = [], input_channels
        for cout in (64, 128, 256, 512, 1024):
            mods += [Conv2d(cin, cout, kernel_size=3, padding=1), ReLU()]
            cin = cout
        self.net = nn.Sequential(*mods)

    def forward(self, x):
        for i in range(0, len(self.net), 2):
            x = self.net[i](x, ACT_RELU)
        return x


class SqueezeExcitation(nn.Module):
    """generator.py:70-84: global average pool -> 1x1 conv -> ReLU -> 1x1 conv -> sigmoid -> channel gate.
    Exported by the reference, not built by the train loop."""

    def __init__(self, channels: int, reduction_ratio: int = 16) -> None:
        super().__init__()
        red = max(1, channels // reduction_ratio)
        self.avg_pool = nn.Identity()            # parameter-free slots keep the reference's attribute names
        self.fc1 = Conv2d(channels, red, kernel_size=1)
        self.relu = ReLU()
        self.fc2 = Conv2d(red, channels, kernel_size=1)
        self.sigmoid = Sigmoid()

    def forward(self, x):
        B, Cn = x.shape[0], x.shape[1]
        a = ops.global_avg_pool(x).view(B, Cn, 1, 1)
        a = self.sigmoid(self.fc2(self.fc1(a, ACT_RELU)))
        return ops.gate_channels(x, a.view(B, Cn))


class CBAMBlock(nn.Module):
    """generator.py:87-101: channel gate (SqueezeExcitation), then a spatial gate from a 7x7 conv over
    [max_c, mean_c]."""

    def __init__(self, channels: int, reduction_ratio: int = 16) -> None:
        super().__init__()
        self.channel_attention = SqueezeExcitation(channels, reduction_ratio)
        self.spatial_attention = nn.Sequential(Conv2d(2, 1, kernel_size=7, padding=3, bias=False), Sigmoid())

    def forward(self, x):
        x = self.channel_attention(x)
        B, _, H, W = x.shape
        a = self.spatial_attention(ops.chan_maxmean(x))
        return ops.gate_pixels(x, a.view(B, H * W))


__all__ = ["OriginalRelationshipLearner", "FlexibleUpsamplingModule", "SqueezeExcitation", "CBAMBlock"]
