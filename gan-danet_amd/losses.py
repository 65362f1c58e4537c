"""Losses of the GAN-DANet G step on HIP kernels (models/losses.py): PerceptualLoss, TVLoss, SSIM, plus the
BCE-with-logits / MSE criteria the notebook takes from torch.nn (GAN_DANet_train.ipynb:L190-191)."""
from __future__ import annotations

import os
import warnings
from typing import Optional, Sequence, Set

import torch
from torch import nn

from . import ops
from .layers import ACT_RELU, Conv2d, ReLU

# torchvision.models.vgg19(...).features ("E" configuration, 37 layers): index -> layer.  The reference keeps
# features[: max(feature_layers) + 1] (losses.py:61); its default taps (1, 6, 11, 20) need the first 21.
_VGG19_HEAD = ("c64", "r", "c64", "r", "p", "c128", "r", "c128", "r", "p",
               "c256", "r", "c256", "r", "c256", "r", "c256", "r", "p",
               "c512", "r", "c512", "r", "c512", "r", "c512", "r", "p",
               "c512", "r", "c512", "r", "c512", "r", "c512", "r", "p")


class _MaxPool2(nn.Module):
    def forward(self, x):
        return ops.MaxPool2Fn.apply(x)


class PerceptualLoss(nn.Module):
    """losses.py:13-73: sum of L1 distances between VGG19 feature maps at ``feature_layers``; 1-channel inputs are
    repeated to 3 channels; VGG is frozen / eval.  Offline there are no pretrained weights: like the reference's
    fallback (losses.py:42-48) the features are randomly initialised unless ``weights_path`` is given.
    ``use_gpu`` is accepted because the notebook still passes it (L194)."""

    def __init__(self, feature_layers: Sequence[int] = (1, 6, 11, 20), weights_path: Optional[str] = None,
                 pretrained: bool = True, device: Optional[torch.device] = None, use_gpu: Optional[bool] = None,
                 chunk: int = 4) -> None:
        super().__init__()
        self.feature_layers: Set[int] = set(feature_layers)
        if not self.feature_layers:
            raise ValueError("feature_layers must contain at least one index")
        top = max(self.feature_layers)
        if top >= len(_VGG19_HEAD):
            raise IndexError(f"VGG19.features has {len(_VGG19_HEAD)} layers; feature layer {top} does not exist")
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = device
        self.chunk = chunk
        if weights_path is None and pretrained:
            warnings.warn("Falling back to randomly initialised VGG19 features (no pretrained weights offline). "
                          "Pass pretrained=False or provide weights_path to silence this warning.", RuntimeWarning)
        mods, cin = [], 3
        for ent in _VGG19_HEAD[: top + 1]:
            if ent[0] == "c":
                cout = int(ent[1:])
                mods.append(Conv2d(cin, cout, kernel_size=3, padding=1))
                mods[-1].layer_class = "vgg"
                cin = cout
            elif ent == "r":
                mods.append(ReLU())
            else:
                mods.append(_MaxPool2())
        self.vgg = nn.Sequential(*mods)
        if weights_path is not None:
            state = torch.load(weights_path, map_location="cpu", weights_only=True)
            missing, unexpected = self.vgg.load_state_dict(state, strict=False)
            if unexpected:
                warnings.warn(f"Unexpected keys when loading VGG weights: {unexpected}", RuntimeWarning)
            if missing:
                warnings.warn(f"Missing keys when loading VGG weights: {missing}", RuntimeWarning)
        self.vgg.to(device)
        self.vgg.eval()
        for p in self.vgg.parameters():
            p.requires_grad_(False)

    def _features(self, t: torch.Tensor):
        """walk the stack; conv+ReLU pairs run as one kernel (ReLU is in place in torchvision, so a tap at a conv
        index sees rectified values only when the NEXT index is a ReLU that has already run -- taps are at ReLU
        indices in every configuration the reference uses, which is what is supported here)."""
        feats = {}
        i, n = 0, len(self.vgg)
        while i < n:
            m = self.vgg[i]
            if isinstance(m, Conv2d):
                if i + 1 < n and isinstance(self.vgg[i + 1], ReLU) and i not in self.feature_layers:
                    t = m(t, ACT_RELU)
                    if i + 1 in self.feature_layers:
                        feats[i + 1] = t
                    i += 2
                    continue
                t = m(t)
            else:
                t = m(t)
            if i in self.feature_layers:
                feats[i] = t
            i += 1
        return feats

    def _nhwc_plan(self, split: bool = False):
        """layer plan + packed operators of the pixel-major bf16 path, or None when the configuration is not
        served by it (taps off ReLU indices or directly before a pool, non-VGG stacks).  Cached; rebuilt when a
        weight tensor changes (load_state_dict).  ``split``: the operators of the split-bf16 form of the same path
        (operand mode "x3": activations [hi | lo | hi], weights [hi ; hi ; lo] along the contraction axis)."""
        from . import kern as K
        convs = [m for m in self.vgg if isinstance(m, Conv2d)]
        version = tuple((id(c.weight), c.weight._version, c.weight.device) for c in convs) + (split,)
        cached = getattr(self, "_nhwc_cache", None)
        if cached is not None and cached[0] == version:
            return cached[1]
        sw = (lambda w, axis: K.split3_weights(w.contiguous(), axis)) if split else (lambda w, axis: w)
        plan = None
        n = len(self.vgg)
        ok = n >= 2 and isinstance(self.vgg[0], Conv2d) and self.vgg[0].weight.shape[1] == 3 and n - 1 in self.feature_layers
        pairs, ops_, taps = [], [], set()
        i = 0
        while ok and i < n:
            m = self.vgg[i]
            if isinstance(m, Conv2d):
                if not (i + 1 < n and isinstance(self.vgg[i + 1], ReLU)) or i in self.feature_layers:
                    ok = False
                    break
                k = len(pairs)
                w = m.weight.detach()
                if k == 0:
                    ops_.append(("stem", k, m.bias.detach()))
                    pairs.append([k, False, None, 3])
                else:
                    ops_.append(("conv", k, m.bias.detach(), K.conv3x3_nhwc_pack(sw(w, 1), False), w.shape[0]))
                    pairs.append([k, False, K.conv3x3_nhwc_pack(sw(w, 0), True), w.shape[1]])
                if i + 1 in self.feature_layers:
                    taps.add(k)
                i += 2
            elif isinstance(m, _MaxPool2):
                if not pairs or pairs[-1][1] or (len(pairs) - 1) in taps or i in self.feature_layers:
                    ok = False
                    break
                pairs[-1][1] = True
                ops_.append(("pool", -1))
                i += 1
            else:
                ok = False
        if ok and pairs and not pairs[-1][1] and (len(pairs) - 1) in taps:
            w0 = self.vgg[0].weight.detach()
            plan = {"ops": ops_, "pairs": [tuple(p) for p in pairs], "taps": taps,
                    # 1-channel inputs are repeated to 3 channels (losses.py:65-68): conv(repeat(x)) = conv_{sum_c w}(x)
                    # (the sum over the 3 input channels = a 1x1 conv with ones over the weight read as an image)
                    "w0": {3: w0.contiguous(),
                           1: K.conv2d_fwd(w0.contiguous(), torch.ones(1, w0.shape[1], 1, 1, device=w0.device), None, 1, 0,
                                           K.L.PREC_FP32)}}
        self._nhwc_cache = (version, plan)
        return plan

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        from .config import operand_mode
        mode = operand_mode("vgg")
        if mode == "x3" and not VGG_SPLIT_NHWC:
            mode = "layers"                            # A/B: the per-layer split route (fp32 NCHW between the convs)
        if (mode in ("16", "x3") and x.shape == y.shape and x.shape[1] in (1, 3) and x.shape[2] % 8 == 0
                and x.shape[3] % 8 == 0 and self._nhwc_plan(mode == "x3") is not None):
            return _PerceptualNhwcFn.apply(x, y, self, mode == "x3")
        x3 = x if x.shape[1] == 3 else ops.repeat_channels(x, 3)
        y3 = y if y.shape[1] == 3 else ops.repeat_channels(y, 3)
        with torch.no_grad():
            fy = self._features(y3)
        fx = self._features(x3)
        terms = [ops.l1_loss(fx[i], fy[i]) for i in sorted(self.feature_layers)]
        return ops.weighted_sum([1.0] * len(terms), terms)


VGG_SPLIT_NHWC = os.environ.get("GD_VGG_SPLIT_NHWC", "1") != "0"


class _PerceptualNhwcFn(torch.autograd.Function):
    """The whole perceptual term as ONE autograd node on pixel-major bf16 activations (bf16 mode): stem conv
    from the fp32 image, gd_conv3x3_nhwc for every other conv (+bias+ReLU fused), NHWC max-pool, L1 feature
    distances; the backward walks the frozen stack with the data-gradient operators, the ReLU backward fused into
    each producer (mask = the saved ReLU output) and the tap gradients added through the epilogue's ``res``.
    ``split`` (operand mode "x3", set_precision("mixed")): the same node on split activations -- every pixel-major tensor
    holds [hi | lo | hi] (3 C bf16 per pixel), every conv is three bf16 MFMAs per product (~2^-16 relative)."""

    @staticmethod
    def forward(ctx, x, y, mod, split=False):
        from . import kern as K
        plan = mod._nhwc_plan(split)
        x, y = x.contiguous(), y.contiguous()
        dev = x.device

        def run(img, keep_all):
            """returns {pair index: ReLU output} (all pairs when keep_all, else only the tapped ones)"""
            kept = {}
            a = None
            for op in plan["ops"]:
                if op[0] == "stem":
                    a = K.nhwc_stem_fwd(img, plan["w0"][img.shape[1]], op[2], True, split=split)
                elif op[0] == "conv":
                    a = K.conv3x3_nhwc(a, op[3], op[2], op[4], relu=True, split=split)
                else:
                    a = K.nhwc_maxpool2_fwd(a, split=split)
                    continue
                if keep_all or op[1] in plan["taps"]:
                    kept[op[1]] = a
            return kept

        with torch.no_grad():
            fy = run(y, False)
            fx = run(x, True)
            loss = torch.zeros(1, device=dev, dtype=torch.float32)
            for n, k in enumerate(sorted(plan["taps"])):
                K.nhwc_l1(fx[k], fy[k], loss, accumulate=n > 0, split=split)
        ctx.plan, ctx.fx, ctx.fy, ctx.ci, ctx.split = plan, fx, fy, x.shape[1], split
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        from . import kern as K
        plan, fx, fy, sp = ctx.plan, ctx.fx, ctx.fy, ctx.split
        up = dloss.reshape(1).to(torch.float32).contiguous()
        pairs = plan["pairs"]                 # forward order: (pair index, followed_by_pool, dgrad pack, Cin)
        top = len(pairs) - 1
        g = K.nhwc_l1_grad(fx[top], fy[top], up, True, split=sp)          # w.r.t. the top conv's pre-activation
        for k in range(top - 1, -1, -1):
            _, pooled, _, _ = pairs[k]
            _, _, pack_t, cin_next = pairs[k + 1]
            if pooled:
                dpool = K.conv3x3_nhwc(g, pack_t, None, cin_next, split=sp)
                g = K.nhwc_maxpool2_bwd(fx[k], dpool, True, split=sp)
            else:
                res = K.nhwc_l1_grad(fx[k], fy[k], up, True, split=sp) if k in plan["taps"] else None
                g = K.conv3x3_nhwc(g, pack_t, None, cin_next, mask=fx[k], res=res, split=sp)
        dx = K.nhwc_stem_bwd(g, plan["w0"][ctx.ci], split=sp)
        ctx.fx = ctx.fy = None
        return dx, None, None, None


class TVLoss(nn.Module):
    """losses.py:76-87"""

    def __init__(self, weight: float = 1.0) -> None:
        super().__init__()
        self.weight = weight

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.tv_loss(x, self.weight)


class SSIM(nn.Module):
    """losses.py:90-147: Gaussian-window SSIM, differentiable w.r.t. both images (``gd_ssim_samples`` /
    ``gd_ssim_bwd``); ``size_average=False`` returns one mean per sample (losses.py:136).  The window is rebuilt from
    ``window_size`` inside the kernels exactly as ``_gaussian`` does (fp32 exp, normalised by the fp32 sum); the
    ``window`` buffer is kept for ``state_dict`` / attribute parity with the reference."""

    def __init__(self, window_size: int = 11, size_average: bool = True) -> None:
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average
        self.channel = 1
        self.register_buffer("window", self._create_window(window_size, 1))

    @staticmethod
    def _create_window(window_size: int, channel: int) -> torch.Tensor:
        coords = torch.arange(window_size, dtype=torch.float32)
        g = torch.exp(-((coords - window_size // 2) ** 2) / (2 * 1.5 ** 2))
        g = (g / g.sum()).unsqueeze(1)
        return (g @ g.t()).unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()

    def forward(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        channel = img1.shape[1]
        if channel != self.channel:                       # losses.py:140-145: the buffer follows the channel count
            self.register_buffer("window", self._create_window(self.window_size, channel).to(img1.device))
            self.channel = channel
        return ops.ssim(img1, img2, self.window_size, self.size_average)


class BCEWithLogitsLoss(nn.Module):
    """torch.nn.BCEWithLogitsLoss() (mean reduction): a constant label (the train loop's all-ones / all-zeros targets,
    L249-253, L261) or a target tensor of the logits' shape."""

    def forward(self, logits: torch.Tensor, target) -> torch.Tensor:
        if torch.is_tensor(target):
            if target.dim() == 0 and not target.is_cuda and not target.requires_grad:
                target = float(target)         # a 0-dim host constant: no device sync, no gradient to keep
            else:
                if target.shape != logits.shape:   # as torch.nn.BCEWithLogitsLoss: no silent broadcast
                    raise ValueError(f"target {tuple(target.shape)} must match the logits {tuple(logits.shape)}")
                return ops.bce_with_logits_target(logits, target)
        return ops.bce_with_logits(logits, float(target))


class MSELoss(nn.Module):
    def forward(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return ops.mse_loss(a, b)


__all__ = ["PerceptualLoss", "TVLoss", "SSIM"]
