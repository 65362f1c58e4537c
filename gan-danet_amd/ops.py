"""Differentiable ops of the GAN-DANet hot path: ``torch.autograd.Function``s whose forward and
backward are sequences of HIP kernels (``kern.py`` -> C ABI).  torch's autograd engine is only the
tape; no ATen compute op runs on the product path.

Precision: ``config.precision`` ("bf16" | "fp32") selects the MFMA operand type of every
GEMM-shaped kernel (fp32 accumulate in both).  "fp32" is the bit-exact parity mode; in "bf16" mode
PAM runs the fused flash kernels, in "fp32" mode it runs the unfused reference-shaped product chain
(materialised N x N matrices, small N only).  CAM always runs in fp32 (its logits scale with N).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
from torch.autograd import Function

from . import _lib as L
from . import kern as K
from .config import config, operand_mode, sixteen_bit

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY02, L.ACT_SIGMOID


def _prec(layer: str = "other") -> int:
    """MFMA operand type of a layer class under the configured mode (config.operand_mode): 16-bit, split-bf16 (the generic
    conv / GEMM kernels split their fp32 operands while staging them) or exact f32"""
    mode = operand_mode(layer)
    return L.PREC_BF16 if mode == "16" else L.PREC_X3 if mode == "x3" else L.PREC_FP32


def _x3(layer: str) -> bool:
    """split-bf16 route (config.operand_mode): the layer's products run as three bf16 MFMAs on hi / lo operand splits"""
    return operand_mode(layer) == "x3"


def _x3_eligible(x, w, stride, pad, act) -> bool:
    """shapes the split-bf16 route serves (3x3 / stride 1 / pad 1 through the pixel-major kernels): Cin >= 32 keeps the
    weight gradient's 32-channel chunk reads inside a [hi | lo | hi] row"""
    return (x.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and stride == 1 and pad == 1 and act in (ACT_NONE, ACT_RELU)
            and w.shape[1] >= 32 and w.shape[1] % 8 == 0 and w.shape[0] % 8 == 0 and (x.shape[2] * x.shape[3]) % 8 == 0
            and x.shape[2] * x.shape[3] * 3 * max(w.shape[0], w.shape[1]) < (1 << 31))


def _pam_f16() -> bool:
    """IEEE fp16 operands in the fused PAM kernels ("fp16": BASELINE config 5; "mixed": 3 more mantissa bits than bf16 at
    the same MFMA rate); all other kernels never take fp16"""
    return config.precision in ("fp16", "mixed")


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# =====================================================================================================
# Discriminator1 conv trunk on pixel-major bf16      discriminator.py:60-63, 66-72
# =====================================================================================================
DISC_NHWC = os.environ.get("GD_DISC_NHWC", "1") != "0"


def disc1_trunk_eligible(x: torch.Tensor, ws) -> bool:
    """16-bit or split-bf16 operand mode, non-deterministic mode (the bias sums and weight gradients use atomics), the
    reference's channel ladder (.. -> 64 -> .. multiples of 8), at most 4 image channels"""
    return (DISC_NHWC and (sixteen_bit("disc") or _x3("disc")) and not K.DETERMINISTIC and x.dim() == 4 and x.shape[1] <= 4
            and ws[0].shape[0] == 64 and all(w.shape[0] % 8 == 0 and w.shape[2:] == (3, 3) for w in ws))


class Disc1TrunkFn(Function):
    """conv1..conv4 (3x3, stride 2, pad 1, + bias + LeakyReLU(0.2)) and ``x.flatten(1)`` as ONE autograd node on
    pixel-major bf16 activations.  The fp32-NCHW path rounds every conv input to bf16 while staging it, so storing the
    activations in bf16 changes no product; what changes is the traffic (2 bytes per element, 16-byte patch copies) and
    the data gradient, which runs by input-pixel parity instead of as four zero-skipping generic launches."""

    SLOPE = 0.2

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, w4, b4):
        x = _c(x)
        # operand mode "x3" (set_precision("mixed")): the same node on SPLIT activations -- every pixel-major tensor holds
        # [hi | lo | hi] (3 C bf16 channels per pixel) against weights split [hi ; hi ; lo] along the contraction axis
        sp = _x3("disc")
        ws, bs = (w1, w2, w3, w4), (b1, b2, b3, b4)
        acts = [K.disc_stem_fwd(x, _c(w1), b1, Disc1TrunkFn.SLOPE, split=sp)]
        for w, b in zip(ws[1:], bs[1:]):
            wp = K.conv3x3_nhwc_pack(K.split3_weights(_c(w), 1) if sp else _c(w), 0)
            acts.append(K.conv3x3_nhwc_s2(acts[-1], wp, b, w.shape[0], 2, Disc1TrunkFn.SLOPE, split=sp))
        ctx.save_for_backward(x, *ws, *acts)       # autograd's version check guards x / the weights against in-place edits
        ctx.has_bias = tuple(b is not None for b in bs)
        ctx.split = sp
        return K.nhwc_flatten_fwd(acts[-1], split=sp)

    @staticmethod
    def backward(ctx, df):
        x, *rest = ctx.saved_tensors
        ws, acts = rest[:4], rest[4:]
        need = ctx.needs_input_grad
        sp = ctx.split
        grads = [None] * 9
        g = K.nhwc_flatten_bwd(_c(df), acts[3], Disc1TrunkFn.SLOPE, split=sp)          # w.r.t. conv4's pre-activation
        for l in (3, 2, 1):
            if need[1 + 2 * l] or need[2 + 2 * l]:
                dw, db = K.conv3x3_wgrad_nhwc(g, acts[l - 1], 2, ctx.has_bias[l] and need[2 + 2 * l], split=sp)
                grads[1 + 2 * l], grads[2 + 2 * l] = (dw if need[1 + 2 * l] else None), db
            wp = K.conv3x3_nhwc_pack(K.split3_weights(_c(ws[l]), 0) if sp else _c(ws[l]), 2)
            g = K.conv3x3_nhwc_s2_dgrad(g, wp, acts[l - 1], Disc1TrunkFn.SLOPE, split=sp)
        if need[1] or need[2]:
            dw, db = K.disc_stem_wgrad(g, x, ctx.has_bias[0] and need[2], split=sp)
            grads[1], grads[2] = (dw if need[1] else None), db
        if need[0]:
            grads[0] = K.disc_stem_dgrad(g, _c(ws[0]), x.shape[2], x.shape[3], split=sp)
        return tuple(grads)


# =====================================================================================================
# convolution (+ bias + activation)        nn.Conv2d  (generator.py / discriminator.py / VGG)
# =====================================================================================================
CONV_WIDE_NHWC = os.environ.get("GD_CONV_WIDE_NHWC", "1") != "0"
DENSE_NHWC = os.environ.get("GD_DENSE_NHWC", "1") != "0"
PAM_CAT = os.environ.get("GD_PAM_CAT", "1") != "0"        # q / k / v projections (and their gradients) as one GEMM each
WIDE_MIN_CIN = int(os.environ.get("GD_WIDE_MIN_CIN", "128"))     # 64 -> 64 at 512 x 512: conv time halves, the two packs eat it (measured equal)


def _wide3x3(x, w, stride, pad, act, prec) -> bool:
    """wide 3x3 / stride 1 / pad 1 convs in 16-bit mode (the 2C -> C fuse conv of DANetAttention, generator.py:108, and the
    C -> 64 conv behind the last block): run on a pixel-major bf16 copy of the input through the NHWC kernel"""
    return (CONV_WIDE_NHWC and prec == L.PREC_BF16 and x.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and stride == 1
            and pad == 1 and act in (ACT_NONE, ACT_RELU) and w.shape[1] >= WIDE_MIN_CIN and w.shape[1] % 8 == 0 and w.shape[0] > 32
            and w.shape[0] % 8 == 0 and (x.shape[2] * x.shape[3]) % 8 == 0)


class Conv2dFn(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride: int, pad: int, act: int, prec=None, layer: str = "other"):
        x3 = prec is None and _x3(layer) and _x3_eligible(x, w, stride, pad, act)
        prec = _prec(layer) if prec is None else prec
        if x3:
            # split-bf16: x -> [hi | lo | hi] pixel-major (3 Cin channels) against the weights [hi ; hi ; lo]; the same
            # pack serves the weight gradient
            B, Cin, H, W = x.shape
            _, x16 = K.pack_split(x)
            y = K.conv3x3_nhwc_f32out(x16, K.conv3x3_nhwc_pack(K.split3_weights(_c(w), 1), 0), bias, w.shape[0], H, W,
                                      relu=act == ACT_RELU)
            ctx.save_for_backward(x16, w, y if act != ACT_NONE else None)
            ctx.cfg = (stride, pad, act, prec, bias is not None)
            ctx.wide = (H, W, True)
            return y
        if _wide3x3(x, w, stride, pad, act, prec):
            # one pixel-major bf16 copy of x serves the forward (16-byte patch staging, no gather / convert in the
            # kernel) and the weight gradient; only that copy is kept for the backward
            B, Cin, H, W = x.shape
            _, x16 = K.pack_bf16(x, Cin, H * W, t_shape=(H * W, Cin))
            y = K.conv3x3_nhwc_f32out(x16, K.conv3x3_nhwc_pack(_c(w), 0), bias, w.shape[0], H, W, relu=act == ACT_RELU)
            ctx.save_for_backward(x16, w, y if act != ACT_NONE else None)
            ctx.cfg = (stride, pad, act, prec, bias is not None)
            ctx.wide = (H, W, False)
            return y
        y = K.conv2d_fwd(x, w, bias, stride, pad, prec, act=act)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.cfg = (stride, pad, act, prec, bias is not None)
        ctx.wide = None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, act, prec, has_bias = ctx.cfg
        dy = _c(dy)
        # split route with frozen weights (the VGG stack of PerceptualLoss): the ReLU backward rides in the pack of dY
        fuse_mask = (ctx.wide is not None and ctx.wide[2] and act == ACT_RELU and not ctx.needs_input_grad[1]
                     and not (has_bias and ctx.needs_input_grad[2]))
        if act != ACT_NONE and not fuse_mask:
            dy = K.act_bwd(y, dy, act)
        dx = dw = db = None
        if ctx.wide is not None and ctx.wide[2]:
            H, W, _ = ctx.wide
            B, Cout = dy.shape[0], dy.shape[1]
            dy2, dyt = K.pack_split(dy.view(B, Cout, H * W), want_plain=ctx.needs_input_grad[1], want_tr=ctx.needs_input_grad[0],
                                    mask=y.view(B, Cout, H * W) if fuse_mask else None)
            if ctx.needs_input_grad[0]:
                dx = K.conv3x3_nhwc_f32out(dyt, K.conv3x3_nhwc_pack(K.split3_weights(_c(w), 0), 1), None, w.shape[1], H, W)
            if ctx.needs_input_grad[1]:
                dw = K.conv3x3_wgrad_x3(dy2, x, H, W)
            if has_bias and ctx.needs_input_grad[2]:
                db = K.channel_sum(dy)
            return dx, dw, db, None, None, None, None, None
        if ctx.wide is not None:
            H, W, _ = ctx.wide
            B, Cout = dy.shape[0], dy.shape[1]
            # dY once in both 16-bit layouts: channel-major for the weight gradient, pixel-major for the data gradient
            dy16, dyt16 = K.pack_bf16(dy.view(B, Cout, H * W), Cout, H * W, plain_shape=(Cout, H * W), t_shape=(H * W, Cout))
            if ctx.needs_input_grad[0]:
                dx = K.conv3x3_nhwc_f32out(dyt16, K.conv3x3_nhwc_pack(_c(w), 1), None, w.shape[1], H, W)
            if ctx.needs_input_grad[1]:
                dw = K.conv3x3_wgrad_packed(dy16, x, H, W)
            if has_bias and ctx.needs_input_grad[2]:
                db = K.channel_sum(dy)
            return dx, dw, db, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d_dgrad(dy, w, (x.shape[2], x.shape[3]), stride, pad, prec)
        if ctx.needs_input_grad[1]:
            dw = K.conv2d_wgrad(dy, x, w.shape[2], stride, pad, prec)
        if has_bias and ctx.needs_input_grad[2]:
            db = K.channel_sum(dy)
        return dx, dw, db, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, act=ACT_NONE, prec=None, layer="other"):
    """``prec``: L.PREC_* override of the configured operand type (weight-space products stay exact fp32);
    ``layer``: the layer class (config.LAYER_CLASSES) whose operand type applies otherwise"""
    return Conv2dFn.apply(x, w, bias, stride, pad, act, prec, layer)


class ActFn(Function):
    """stand-alone ReLU / LeakyReLU(0.2) (used where the activation is not fused into a producer)"""

    @staticmethod
    def forward(ctx, x, act: int):
        y = K.act_fwd(_c(x), act)
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return K.act_bwd(y, _c(dy), ctx.act), None


def activation(x, act: int):
    return ActFn.apply(x, act)


# =====================================================================================================
# BatchNorm2d (+ activation)               generator.py:32,61,149,189,219,223
# =====================================================================================================
def _sync_bn_active() -> bool:
    """SyncBN (config.sync_bn) applies when there is more than one rank to synchronise with"""
    if not config.sync_bn:
        return False
    from .parallel import is_distributed
    return is_distributed()


def _all_gather_records(stats: torch.Tensor) -> torch.Tensor:
    """(C, 3) per-rank BatchNorm records -> (world, C, 3) on every rank (torch.distributed: plumbing)"""
    import torch.distributed as dist
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(stats.shape), device=stats.device, dtype=stats.dtype)
    if dist.get_backend() == "gloo":
        dist.all_gather([out[r] for r in range(world)], stats)
    else:
        dist.all_gather_into_tensor(out, stats)
    return out


def _bn_prepare(x, gamma, beta, rmean, rvar, training, momentum, eps):
    """statistics (+ running update) and the folded affine; returns (scale, shift, mean, invstd)"""
    if training and _sync_bn_active():
        mean, invstd = K.bn_stats_merge(_all_gather_records(K.bn_stats_local(x)), eps, momentum, rmean, rvar)
        scale, shift = K.bn_fold(gamma, beta, mean, invstd)
    elif training:
        mean, invstd = K.bn_stats(x, eps, momentum, rmean, rvar)
        scale, shift = K.bn_fold(gamma, beta, mean, invstd)
    else:
        scale, shift, invstd = K.bn_fold_eval(gamma, beta, rmean, rvar, eps)
        mean = rmean
    return scale, shift, mean, invstd


def _bn_bwd(dy, x, scale, shift, mean, invstd, act, training, sync, dx=None, accumulate_dx=False):
    """backward of act(bn(x)): (dgamma, dbeta, dx).  sync (SyncBN, training): this rank's dgamma / dbeta are the parameter
    gradients (the gradient all-reduce sums them over ranks like every other parameter); dx uses their all-reduced sums and
    the global element count"""
    if not (sync and training):
        return K.bn_act_bwd(dy, x, scale, shift, mean, invstd, act, training, dx=dx, accumulate_dx=accumulate_dx)
    import torch.distributed as dist
    Cn = x.shape[1]
    sums = torch.empty(2, Cn, device=x.device, dtype=torch.float32)
    K.bn_act_bwd(dy, x, scale, shift, mean, invstd, act, training, want_dx=False, sums_out=sums)
    local = torch.empty_like(sums)
    K.copy_slab(sums.view(1, 2 * Cn, 1), local.view(1, 2 * Cn, 1))
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    n_global = x.shape[0] * x[0, 0].numel() * dist.get_world_size()
    dx = K.bn_act_bwd_dx(dy, x, scale, shift, mean, invstd, sums[0], sums[1], 1.0 / n_global, act, dx=dx,
                         accumulate_dx=accumulate_dx)
    return local[0], local[1], dx


class BnActFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training: bool, momentum: float, eps: float, act: int):
        scale, shift, mean, invstd = _bn_prepare(x, gamma, beta, rmean, rvar, training, momentum, eps)
        y = K.affine_act(x, scale, shift, act)
        ctx.save_for_backward(x, scale, shift, mean, invstd)
        ctx.cfg = (act, training, training and _sync_bn_active())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, invstd = ctx.saved_tensors
        act, training, sync = ctx.cfg
        dgamma, dbeta, dx = _bn_bwd(_c(dy), x, scale, shift, mean, invstd, act, training, sync)
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, rmean, rvar, training, momentum, eps, act=ACT_NONE):
    return BnActFn.apply(x, gamma, beta, rmean, rvar, training, momentum, eps, act)


class BnReluConvFn(Function):
    """conv(relu(bn(x))) with the BN affine + ReLU folded into the conv's operand load (no intermediate).
    TransitionLayer (generator.py:57-67)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, w, bias, training, momentum, eps, pad):
        prec = _prec("conv1x1" if w.shape[2] == 1 else "other")
        scale, shift, mean, invstd = _bn_prepare(x, gamma, beta, rmean, rvar, training, momentum, eps)
        y = K.conv2d_fwd(x, w, bias, 1, pad, prec, in_scale=scale, in_shift=shift, in_relu=True)
        ctx.save_for_backward(x, scale, shift, mean, invstd, w)
        ctx.cfg = (training, prec, pad, bias is not None, training and _sync_bn_active())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, invstd, w = ctx.saved_tensors
        training, prec, pad, has_bias, sync = ctx.cfg
        dy = _c(dy)
        dw = K.conv2d_wgrad(dy, x, w.shape[2], 1, pad, prec, in_scale=scale, in_shift=shift, in_relu=True)
        db = K.channel_sum(dy) if has_bias else None
        dxt = K.conv2d_dgrad(dy, w, (x.shape[2], x.shape[3]), 1, pad, prec)
        dgamma, dbeta, dx = _bn_bwd(dxt, x, scale, shift, mean, invstd, ACT_RELU, training, sync)
        return dx, dgamma, dbeta, None, None, dw, db, None, None, None, None


# =====================================================================================================
# DenseBlock: cat-free slab                generator.py:29-54
# =====================================================================================================
class DenseBlockFn(Function):
    """The whole block as one node: the output slab (B, C0 + L*g, H, W) is allocated once, the input is
    copied into its first C0 channels and every layer's conv writes its g new channels in place -- no
    torch.cat.  BN+ReLU of each layer is folded into that layer's conv operand load.
    args: x, training, momentum, eps, then per layer (bn_w, bn_b, bn_rm, bn_rv, conv_w, conv_b)."""

    @staticmethod
    def forward(ctx, x, training: bool, momentum: float, eps: float, *params):
        prec = _prec("dense3x3")
        nl = len(params) // 6
        B, C0, H, W = x.shape
        g = params[4].shape[0]
        slab = torch.empty(B, C0 + nl * g, H, W, device=x.device, dtype=torch.float32)
        K.copy_slab(x, slab[:, :C0])
        saved: List[torch.Tensor] = []
        # 16-bit mode: every layer's conv input max(0, bn(x)) is packed ONCE as a pixel-major bf16 copy; the forward conv
        # (NHWC kernel, fp32 result straight into the slab), and in the backward the weight gradient, read that copy
        # instead of gathering / normalising the fp32 NCHW slab in their staging loops
        aligned = ((H * W) % 8 == 0 and g % 8 == 0 and all((C0 + l * g) % 8 == 0 for l in range(nl))
                   and tuple(params[4].shape[2:]) == (3, 3))
        nhwc = DENSE_NHWC and prec == L.PREC_BF16 and aligned
        # "mixed": the same route on split-bf16 operands -- max(0, bn(x)) packed once as [hi | lo | hi] pixel-major
        x3 = prec == L.PREC_X3 and aligned and C0 >= 32 and H * W * 3 * (C0 + nl * g) < (1 << 31)
        packs: List[torch.Tensor] = []
        for l in range(nl):
            bw, bb, rm, rv, cw, cb = params[6 * l: 6 * l + 6]
            cl = C0 + l * g
            xin = slab[:, :cl]
            scale, shift, mean, invstd = _bn_prepare(xin, bw, bb, rm, rv, training, momentum, eps)
            if x3:
                _, x16 = K.pack_split(xin, scale=scale, shift=shift, relu=True)
                K.conv3x3_nhwc_f32out(x16, K.conv3x3_nhwc_pack(K.split3_weights(_c(cw), 1), 0), cb, g, H, W,
                                      out=slab[:, cl:cl + g])
                packs.append(x16)
            elif nhwc:
                x16 = K.pack_nhwc16_affine(xin, scale, shift, True)
                K.conv3x3_nhwc_f32out(x16, K.conv3x3_nhwc_pack(_c(cw), 0), cb, g, H, W, out=slab[:, cl:cl + g])
                packs.append(x16)
            else:
                K.conv2d_fwd(xin, cw, cb, 1, 1, prec, in_scale=scale, in_shift=shift, in_relu=True,
                             out=slab[:, cl:cl + g])
            saved += [scale, shift, mean, invstd, cw]
        ctx.save_for_backward(slab, *saved, *packs)
        ctx.cfg = (nl, C0, g, training, prec, [p is not None for p in params[5::6]], nhwc, x3, training and _sync_bn_active())
        return slab

    @staticmethod
    def backward(ctx, dslab_in):
        slab, *saved = ctx.saved_tensors
        nl, C0, g, training, prec, has_bias, nhwc, x3, sync = ctx.cfg
        packs = saved[5 * nl:]
        B, _, H, W = slab.shape
        dslab = torch.empty(slab.shape, device=slab.device, dtype=torch.float32)  # accumulated into below
        K.copy_slab(_c(dslab_in), dslab)
        grads: List[Optional[torch.Tensor]] = [None] * (6 * nl)
        for l in reversed(range(nl)):
            scale, shift, mean, invstd, cw = saved[5 * l: 5 * l + 5]
            cl = C0 + l * g
            xin = slab[:, :cl]
            dy = dslab[:, cl:cl + g]
            if x3:
                dy2, dyt = K.pack_split(_as3(dy), want_plain=True)
                grads[6 * l + 4] = K.conv3x3_wgrad_x3(dy2, packs[l], H, W)
                dxt = K.conv3x3_nhwc_f32out(dyt, K.conv3x3_nhwc_pack(K.split3_weights(_c(cw), 0), 1), None, cl, H, W)
            elif nhwc:
                dy16, dyt16 = K.pack_bf16(_as3(dy), g, H * W, plain_shape=(g, H * W), t_shape=(H * W, g))
                grads[6 * l + 4] = K.conv3x3_wgrad_packed(dy16, packs[l], H, W)
                dxt = K.conv3x3_nhwc_f32out(dyt16, K.conv3x3_nhwc_pack(_c(cw), 1), None, cl, H, W)
            else:
                grads[6 * l + 4] = K.conv2d_wgrad(dy, xin, 3, 1, 1, prec, in_scale=scale, in_shift=shift, in_relu=True)
                dxt = K.conv2d_dgrad(dy, cw, (H, W), 1, 1, prec)
            if has_bias[l]:
                grads[6 * l + 5] = K.channel_sum(dy)
            dgamma, dbeta, _ = _bn_bwd(dxt, xin, scale, shift, mean, invstd, ACT_RELU, training, sync,
                                       dx=dslab[:, :cl], accumulate_dx=True)
            grads[6 * l + 0], grads[6 * l + 1] = dgamma, dbeta
        dx = torch.empty(B, C0, H, W, device=slab.device, dtype=torch.float32)
        K.copy_slab(dslab[:, :C0], dx)
        return (dx, None, None, None, *grads)


# =====================================================================================================
# dual attention: PAM || CAM into one 2C slab      generator.py:104-157
# =====================================================================================================
# the pre-v3 dK/dV kernels (GD_PAM_DKV_V3=0, kept for A/B runs) also need q and gamma*dOut channel-major
def _npad(n: int) -> int:
    return (n + 255) // 256 * 256


def _cp(c: int) -> int:
    return (c + 31) // 32 * 32


def _pam_forward(x, wq, bq, wk, bk, wv, bv, gamma_p, out3, prec):
    """PAM into ``out3`` (B, C, N) view.  Returns (fused?, tensors to keep for backward)."""
    B, Cn, H, W = x.shape
    N = H * W
    r = wq.shape[0]
    x3 = x.view(B, Cn, N)
    fused = sixteen_bit("pam") and Cn <= 192 and r <= 31
    if fused and PAM_CAT and (bq is None) == (bk is None) == (bv is None):
        # q, k, v as ONE 1x1 conv over the concatenated weights: x is read once instead of three times
        wc = torch.empty(2 * r + Cn, Cn, 1, 1, device=x.device, dtype=torch.float32)
        for w_, lo in ((wq, 0), (wk, r), (wv, 2 * r)):
            K.copy_slab(_c(w_).view(1, w_.shape[0], Cn), wc.view(1, 2 * r + Cn, Cn)[:, lo:lo + w_.shape[0]])
        bc = None
        if bq is not None:
            bc = torch.empty(2 * r + Cn, device=x.device, dtype=torch.float32)
            for b_, lo in ((bq, 0), (bk, r), (bv, 2 * r)):
                K.copy_slab(b_.view(1, -1, 1), bc.view(1, -1, 1)[:, lo:lo + b_.numel()])
        y3 = K.conv2d_fwd(x, wc, bc, 1, 0, prec).view(B, 2 * r + Cn, N)
        q, k, v = y3[:, :r], y3[:, r:2 * r], y3[:, 2 * r:]
    else:
        q = K.conv2d_fwd(x, wq, bq, 1, 0, prec).view(B, r, N)
        k = K.conv2d_fwd(x, wk, bk, 1, 0, prec).view(B, r, N)
        v = K.conv2d_fwd(x, wv, bv, 1, 0, prec).view(B, Cn, N)
    if fused:
        Np, Cp = _npad(N), _cp(Cn)
        # q goes in pre-scaled by log2 e; a spare padded channel of V carries ones (softmax denominator by MFMA)
        ones = Cp - 1 if Cn < Cp else -1
        f16 = _pam_f16()
        _, qt = K.pack_bf16(q, r, N, scale_imm=K.LOG2E, t_shape=(Np, 32), f16=f16)
        kn, kt = K.pack_bf16(k, r, N, plain_shape=(32, Np), t_shape=(Np, 32), perm16=True, ones_row=31, f16=f16)
        vn, vt = K.pack_bf16(v, Cn, N, plain_shape=(Cp, Np), t_shape=(Np, Cp), perm16=True, ones_row=ones, f16=f16)
        del q, k, v
        y3 = None
        o_attn = torch.empty(B, Cn, N, device=x.device, dtype=torch.float32)
        lse = torch.empty(B, N, device=x.device, dtype=torch.float32)
        if K.PAM_SHIFT and not f16:
            # bf16 operands: per-query softmax shift from a strided key sample, max-free sweep for logits of any magnitude
            K.pam_flash_fwd_shift(qt, kt, vn, B, N, Np, Cn, Cp, gamma_p, x3, out3, o_attn, lse, r_alg=r, v_ones=ones >= 0)
        else:
            k_sqmax = K.pam_key_sqnorm_max(kt, N, f16) if K.PAM_NOMAX else None
            K.pam_flash_fwd(qt, kt, vn, B, N, Np, Cn, Cp, gamma_p, x3, out3, o_attn, lse, r_alg=r, v_ones=ones >= 0, f16=f16,
                            k_sqmax=k_sqmax)
        return True, (qt, kt, kn, vt, o_attn, lse)
    prec = prec if sixteen_bit("pam") else L.PREC_FP32     # the reference-shaped product chain below
    qt_, kt_ = K.transpose(q), K.transpose(k)              # (B, N, r)
    s = torch.empty(B, N, N, device=x.device, dtype=torch.float32)
    K.gemm_nt(B=B, M=N, N=N, kseg=1, klen=r, a=qt_, a_bs=N * r, a_ss=0, lda=r, bm=kt_, b_bs=N * r, b_ss=0,
              ldb=r, c=s, c_bs=N * N, ldc=N, precision=prec, splits=1)
    p = K.softmax_rows(s, 1.0, out=s)
    o_attn = torch.empty(B, Cn, N, device=x.device, dtype=torch.float32)
    K.gemm_nt(B=B, M=Cn, N=N, kseg=1, klen=N, a=v, a_bs=Cn * N, a_ss=0, lda=N, bm=p, b_bs=N * N, b_ss=0,
              ldb=N, c=o_attn, c_bs=Cn * N, ldc=N, precision=prec, splits=1)
    K.copy_slab(x3, out3)
    _axpy_dev3(o_attn, gamma_p, out3)
    return False, (qt_, kt_, v, p, o_attn)


def _pam_backward(fused, pam_saved, x, wq, wk, wv, gamma_p, d_pam, dx, prec, has_bias):
    """d_pam: (B, C, N) view of dOut.  Accumulates the projection data-gradients into ``dx`` (B, C, N; the
    residual term is added by the caller).  Returns (dwq, dbq, dwk, dbk, dwv, dbv, dgamma)."""
    B, Cn, H, W = x.shape
    N = H * W
    r = wq.shape[0]
    if fused:
        qt, kt, kn, vt, o_attn, lse = pam_saved
        Np, Cp = _npad(N), _cp(Cn)
        d_raw, delta = K.chan_dot(d_pam, o_attn, gamma_p)
        dgamma_p = K.dot(d_raw, None)
        f16 = qt.dtype == torch.float16          # the operand type the forward packed
        s_up, s_inv = gamma_p, None
        if f16:
            # fp16 has no range for real training gradients (gamma * dOut ~ 1e-8): power-of-two scale in, 2^-k out
            # through the alpha of the projection-gradient GEMMs (gd_pam_f16_scale; every output is linear in dOut)
            scales = K.pam_f16_scale(d_pam, gamma_p, delta)
            s_up, s_inv = scales[0:1], scales[1:2]
        _, dot_ = K.pack_bf16(d_pam, Cn, N, scale=s_up, t_shape=(Np, Cp), f16=f16)
        if PAM_CAT and Np == N and K.pam_bwd_form() == L.PAM_BWD_K64_ATOMIC:
            # dq | dk | dv as row blocks of ONE buffer: the three projection data / weight / bias gradients below become
            # one GEMM each over it (dx read-modify-written once instead of three times, x read once)
            R = 64 + Cp
            dcat = torch.empty(B, R, Np, device=x.device, dtype=torch.float32)
            K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dcat[:, 0:32], dcat[:, 32:64], dcat[:, 64:],
                            r_alg=r, c_alg=Cn, f16=f16, out_bs=R * Np)
            wcat = torch.zeros(R, Cn, 1, 1, device=x.device, dtype=torch.float32)     # rows beyond r / C stay zero
            for w_, lo in ((wq, 0), (wk, 32), (wv, 64)):
                K.copy_slab(_c(w_).view(1, w_.shape[0], Cn), wcat.view(1, R, Cn)[:, lo:lo + w_.shape[0]])
            dcat4 = dcat.view(B, R, H, W)
            dwc = K.conv2d_wgrad(dcat4, x, 1, 1, 0, prec, alpha=s_inv)
            dbc = K.channel_sum(dcat4) if any(has_bias) else None
            if dbc is not None and s_inv is not None:
                K.scale_dev(dbc, s_inv, out=dbc)
            K.conv2d_dgrad(dcat4, wcat, (H, W), 1, 0, prec, out=dx.view(B, Cn, H, W), accumulate=True, alpha=s_inv)
            grads = []
            for (lo, n), hb in zip(((0, r), (32, r), (64, Cn)), has_bias):
                grads.append(dwc[lo:lo + n])
                grads.append(dbc[lo:lo + n] if hb else None)
            return (*grads, dgamma_p)
        dqn = torch.empty(B, 32, Np, device=x.device, dtype=torch.float32)
        dkn = torch.empty(B, 32, Np, device=x.device, dtype=torch.float32)
        dvp = torch.empty(B, Cp, Np, device=x.device, dtype=torch.float32)
        K.pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Np, Cp, dqn, dkn, dvp, r_alg=r, c_alg=Cn, f16=f16)
        if s_inv is not None:
            for t in (dqn, dkn, dvp):
                K.scale_dev(t, s_inv, out=t)
        dq, dk, dv = _compact(dqn, r, N), _compact(dkn, r, N), _compact(dvp, Cn, N)
    else:
        qt_, kt_, v, p, o_attn = pam_saved
        pprec = prec if sixteen_bit("pam") else L.PREC_FP32
        dop = torch.empty(B, Cn, N, device=x.device, dtype=torch.float32)
        K.copy_slab(d_pam, dop)
        dgamma_p = K.dot(dop, o_attn)
        K.scale_dev(dop, gamma_p, out=dop)                       # gamma * dOut (in place)
        dp = torch.empty(B, N, N, device=x.device, dtype=torch.float32)
        # dP[i][j] = sum_c dO'[c][i] V[c][j]
        K.conv_nn(B=B, M=N, Ck=Cn, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=N, Ho=1, Wo=N, a=dop,
                  a_bs=Cn * N, a_sm=1, a_sc=N, a_st=0, x=v, x_bs=Cn * N, y=dp, y_bs=N * N, precision=pprec)
        ds = K.softmax_rows_bwd(p, dp, 1.0)
        del dp
        # dV[c][j] = sum_i dO'[c][i] P[i][j]
        dv = torch.empty(B, Cn, N, device=x.device, dtype=torch.float32)
        K.conv_nn(B=B, M=Cn, Ck=N, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=N, Ho=1, Wo=N, a=dop,
                  a_bs=Cn * N, a_sm=N, a_sc=1, a_st=0, x=p, x_bs=N * N, y=dv, y_bs=Cn * N, precision=pprec)
        # dQt[i][d] = sum_j dS[i][j] Kt[j][d] ; dKt[j][d] = sum_i dS[i][j] Qt[i][d]
        dqt = torch.empty(B, N, r, device=x.device, dtype=torch.float32)
        dkt = torch.empty(B, N, r, device=x.device, dtype=torch.float32)
        K.conv_nn(B=B, M=N, Ck=N, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=r, Ho=1, Wo=r, a=ds,
                  a_bs=N * N, a_sm=N, a_sc=1, a_st=0, x=kt_, x_bs=N * r, y=dqt, y_bs=N * r, precision=pprec)
        K.conv_nn(B=B, M=N, Ck=N, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=r, Ho=1, Wo=r, a=ds,
                  a_bs=N * N, a_sm=1, a_sc=N, a_st=0, x=qt_, x_bs=N * r, y=dkt, y_bs=N * r, precision=pprec)
        dq, dk = K.transpose(dqt), K.transpose(dkt)               # (B, r, N)
    grads = []
    for dy3, w, hb in ((dq, wq, has_bias[0]), (dk, wk, has_bias[1]), (dv, wv, has_bias[2])):
        dy4 = dy3.view(B, dy3.shape[1], H, W)
        grads.append(K.conv2d_wgrad(dy4, x, 1, 1, 0, prec))
        grads.append(K.channel_sum(dy4) if hb else None)
        K.conv2d_dgrad(dy4, w, (H, W), 1, 0, prec, out=dx.view(B, Cn, H, W), accumulate=True)
    return (*grads, dgamma_p)


def _cam_forward(x, gamma_c, out3, prec=L.PREC_FP32):
    """CAM into ``out3``.  The Gram matrix / logits are always fp32 MFMA (they scale with N: bf16 operands would
    scramble the softmax); ``prec`` is the operand type of the attention *apply* product only."""
    B, Cn, H, W = x.shape
    N = H * W
    x3 = x.view(B, Cn, N)
    e = torch.empty(B, Cn, Cn, device=x.device, dtype=torch.float32)
    K.gemm_nt(B=B, M=Cn, N=Cn, kseg=1, klen=N, a=x3, a_bs=Cn * N, a_ss=0, lda=N, bm=x3, b_bs=Cn * N, b_ss=0,
              ldb=N, c=e, c_bs=Cn * Cn, ldc=Cn, precision=L.PREC_FP32)
    att = K.softmax_rows(e, -1.0, out=e)   # softmax(rowmax(E) - E) == softmax(-E)
    K.conv_nn(B=B, M=Cn, Ck=Cn, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=N, Ho=1, Wo=N, a=att,
              a_bs=Cn * Cn, a_sm=Cn, a_sc=1, a_st=0, x=x3, x_bs=Cn * N, y=out3, y_bs=K._bview(out3),
              precision=prec, alpha=gamma_c, res=x3, res_bs=Cn * N)
    return att


def _cam_backward(att, x, gamma_c, d_cam, dx, prec=L.PREC_FP32):
    """accumulates gamma * (att^T dOut + (dE + dE^T) X) into dx (residual added by the caller); returns dgamma"""
    B, Cn, H, W = x.shape
    N = H * W
    x3 = x.view(B, Cn, N)
    d_bs = K._bview(d_cam)
    da = torch.empty(B, Cn, Cn, device=x.device, dtype=torch.float32)
    K.gemm_nt(B=B, M=Cn, N=Cn, kseg=1, klen=N, a=d_cam, a_bs=d_bs, a_ss=0, lda=N, bm=x3, b_bs=Cn * N, b_ss=0,
              ldb=N, c=da, c_bs=Cn * Cn, ldc=Cn, precision=L.PREC_FP32)
    dgamma_c = K.dot(att, da)
    de = K.softmax_rows_bwd(att, da, -1.0)
    sym = K.add_transpose(de)
    K.conv_nn(B=B, M=Cn, Ck=Cn, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=N, Ho=1, Wo=N, a=att,
              a_bs=Cn * Cn, a_sm=1, a_sc=Cn, a_st=0, x=d_cam, x_bs=d_bs, y=dx, y_bs=Cn * N,
              precision=prec, alpha=gamma_c, accumulate=True)
    K.conv_nn(B=B, M=Cn, Ck=Cn, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=N, Ho=1, Wo=N, a=sym,
              a_bs=Cn * Cn, a_sm=Cn, a_sc=1, a_st=0, x=x3, x_bs=Cn * N, y=dx, y_bs=Cn * N,
              precision=prec, alpha=gamma_c, accumulate=True)
    return dgamma_c


class PamFn(Function):
    """PAMModule.forward (generator.py:113-122)"""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma_p):
        x = _c(x)
        prec = _prec("conv1x1")
        out = torch.empty_like(x)
        fused, saved = _pam_forward(x, wq, bq, wk, bk, wv, bv, gamma_p, _as3(out), prec)
        ctx.save_for_backward(x, wq, wk, wv, gamma_p, *saved)
        ctx.cfg = (prec, fused, (bq is not None, bk is not None, bv is not None))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wq, wk, wv, gamma_p, *saved = ctx.saved_tensors
        prec, fused, has_bias = ctx.cfg
        B, Cn, H, W = x.shape
        d3 = _as3(_c(dout))
        dx = torch.empty(B, Cn, H * W, device=x.device, dtype=torch.float32)
        K.copy_slab(d3, dx)
        dwq, dbq, dwk, dbk, dwv, dbv, dg = _pam_backward(fused, saved, x, wq, wk, wv, gamma_p, d3, dx, prec, has_bias)
        return dx.view(B, Cn, H, W), dwq, dbq, dwk, dbk, dwv, dbv, dg


class CamFn(Function):
    """CAMModule.forward (generator.py:130-139)"""

    @staticmethod
    def forward(ctx, x, gamma_c):
        x = _c(x)
        out = torch.empty_like(x)
        ctx.prec = _prec("cam_apply")
        att = _cam_forward(x, gamma_c, _as3(out), ctx.prec)
        ctx.save_for_backward(x, gamma_c, att)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma_c, att = ctx.saved_tensors
        B, Cn, H, W = x.shape
        d3 = _as3(_c(dout))
        dx = torch.empty(B, Cn, H * W, device=x.device, dtype=torch.float32)
        K.copy_slab(d3, dx)
        dg = _cam_backward(att, x, gamma_c, d3, dx, ctx.prec)
        return dx.view(B, Cn, H, W), dg


class DualAttentionFn(Function):
    """features = cat([PAM(x), CAM(x)], 1) written as the two halves of one (B, 2C, H, W) buffer
    (DANetAttention.forward, generator.py:153-156, without the cat)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma_p, gamma_c):
        x = _c(x)
        B, Cn, H, W = x.shape
        prec, cprec = _prec("conv1x1"), _prec("cam_apply")
        feats = torch.empty(B, 2 * Cn, H, W, device=x.device, dtype=torch.float32)
        fused, saved = _pam_forward(x, wq, bq, wk, bk, wv, bv, gamma_p, _as3(feats[:, :Cn]), prec)
        att = _cam_forward(x, gamma_c, _as3(feats[:, Cn:]), cprec)
        ctx.save_for_backward(x, wq, wk, wv, gamma_p, gamma_c, att, *saved)
        ctx.cfg = (prec, fused, (bq is not None, bk is not None, bv is not None), cprec)
        return feats

    @staticmethod
    def backward(ctx, dfeat):
        x, wq, wk, wv, gamma_p, gamma_c, att, *saved = ctx.saved_tensors
        prec, fused, has_bias, cprec = ctx.cfg
        B, Cn, H, W = x.shape
        dfeat = _c(dfeat)
        d_pam, d_cam = _as3(dfeat[:, :Cn]), _as3(dfeat[:, Cn:])
        dx = torch.empty(B, Cn, H * W, device=x.device, dtype=torch.float32)
        K.copy_slab(d_pam, dx)                     # residual path of PAM
        K.copy_slab(d_cam, dx, accumulate=True)    # residual path of CAM
        dgc = _cam_backward(att, x, gamma_c, d_cam, dx, cprec)
        dwq, dbq, dwk, dbk, dwv, dbv, dgp = _pam_backward(fused, saved, x, wq, wk, wv, gamma_p, d_pam, dx, prec, has_bias)
        return dx.view(B, Cn, H, W), dwq, dbq, dwk, dbk, dwv, dbv, dgp, dgc


def _as3(t: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) view (possibly a channel slice) -> (B, C, H*W) view with the same batch stride"""
    B, Cn, H, W = t.shape
    return t.as_strided((B, Cn, H * W), (t.stride(0), H * W, 1), t.storage_offset())


def _axpy_dev3(src: torch.Tensor, s: torch.Tensor, dst3: torch.Tensor) -> None:
    """dst3 (strided per image) += s * src (dense), image by image stride aware"""
    B = src.shape[0]
    if dst3.is_contiguous():
        K.scale_dev(src, s, out=dst3, accumulate=True)
        return
    for b in range(B):
        K.scale_dev(src[b], s, out=dst3[b], accumulate=True)


def _compact(t: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """(B, Rp, Np) padded plane -> dense (B, rows, cols)"""
    B, Rp, Np = t.shape
    if Rp == rows and Np == cols:
        return t
    if Np == cols:
        return t[:, :rows]          # per-image dense channel slice: kernels take the batch stride
    out = torch.empty(B, rows, cols, device=t.device, dtype=torch.float32)
    K.copy_rows(t, Rp * Np, Np, out, rows * cols, cols, B, rows, cols)
    return out


# =====================================================================================================
# resampling
# =====================================================================================================
class BicubicFn(Function):
    @staticmethod
    def forward(ctx, x, Ho: int, Wo: int, rsh: float, rsw: float):
        x = _c(x)
        ctx.cfg = (x.shape[2], x.shape[3], rsh, rsw)
        return K.bicubic_fwd(x, Ho, Wo, rsh, rsw)

    @staticmethod
    def backward(ctx, dy):
        Hi, Wi, rsh, rsw = ctx.cfg
        return K.bicubic_bwd(_c(dy), Hi, Wi, rsh, rsw), None, None, None, None


def bicubic_up2(x):
    return BicubicFn.apply(x, 2 * x.shape[2], 2 * x.shape[3], 0.5, 0.5)


class BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, Ho: int, Wo: int):
        x = _c(x)
        ctx.cfg = (x.shape[2], x.shape[3])
        return K.bilinear_fwd(x, Ho, Wo)

    @staticmethod
    def backward(ctx, dy):
        Hi, Wi = ctx.cfg
        return K.bilinear_bwd(_c(dy), Hi, Wi), None, None


class SkipFuseFn(Function):
    """x + sum_k adjust_k(bilinear_up(f_k))  ==  x + bilinear_up(sum_k adjust_k(f_k))   (generator.py:243-245)
    The 1x1 ``channel_adjust`` convs (no bias) commute with the bilinear resize (both linear), so they run at
    the LOW resolution (16x fewer pixels) and all skips share one upsample-add.
    args: x, then n weights, then n features."""

    @staticmethod
    def forward(ctx, x, *wf):
        prec = _prec("conv1x1")
        n = len(wf) // 2
        ws, fs = wf[:n], [_c(f) for f in wf[n:]]
        x = _c(x)
        B, Co, Ho, Wo = x.shape
        s = K.conv2d_fwd(fs[0], ws[0], None, 1, 0, prec)
        for w, f in zip(ws[1:], fs[1:]):
            if f.shape[2:] != fs[0].shape[2:]:
                raise L.GandanetError("SkipFuseFn: all skip features must share one resolution")
            K.conv2d_fwd(f, w, None, 1, 0, prec, out=s, accumulate=True)
        out = K.bilinear_fwd(s, Ho, Wo, res=x)     # x + up(s): one pass over the 4H x 4W tensor
        ctx.save_for_backward(*ws, *fs)
        ctx.cfg = (prec, n)
        return out

    @staticmethod
    def backward(ctx, dout):
        prec, n = ctx.cfg
        saved = ctx.saved_tensors
        ws, fs = saved[:n], saved[n:]
        dout = _c(dout)
        Hi, Wi = fs[0].shape[2], fs[0].shape[3]
        ds = K.bilinear_bwd(dout, Hi, Wi)
        gw = [K.conv2d_wgrad(ds, f, 1, 1, 0, prec) for f in fs]
        gf = [K.conv2d_dgrad(ds, w, (Hi, Wi), 1, 0, prec) for w in ws]
        return (dout, *gw, *gf)


class ShiftSum9Fn(Function):
    """y = bias + sum_tap shift(u_tap): the 3x3 / pad 1 / one-output-channel conv applied to its nine pre-contracted tap
    planes (see FlexibleUpsamplingModule.forward)"""

    @staticmethod
    def forward(ctx, u, bias):
        ctx.has_bias = bias is not None
        return K.shift_sum9_fwd(_c(u), bias)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        return K.shift_sum9_bwd(dy), (K.dot(dy, None).view(1) if ctx.has_bias else None)


class MaxPool2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return K.maxpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.maxpool2_bwd(x, _c(dy))


# =====================================================================================================
# linear (+ activation)                    nn.Linear / nn.LazyLinear (discriminator.py:66-67,76-77)
# =====================================================================================================
class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, bias, act: int):
        prec = _prec("disc")
        if prec == L.PREC_X3:
            prec = L.PREC_FP32     # weight-streaming skinny GEMMs (fc1: 8.6 GB per pass) are HBM-bound: exact costs nothing
        x = _c(x)
        Bn, Kin = x.shape
        Nout = w.shape[0]
        y = torch.empty(Bn, Nout, device=x.device, dtype=torch.float32)
        K.gemm_nt(B=1, M=Bn, N=Nout, kseg=1, klen=Kin, a=x, a_bs=0, a_ss=0, lda=Kin, bm=w, b_bs=0, b_ss=0, ldb=Kin,
                  c=y, c_bs=0, ldc=Nout, precision=prec, bias=bias)
        if act != ACT_NONE:
            K.act_fwd(y, act, out=y)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.cfg = (act, prec, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        act, prec, has_bias = ctx.cfg
        dy = _c(dy)
        if act != ACT_NONE:
            dy = K.act_bwd(y, dy, act)
        Bn, Kin = x.shape
        Nout = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:   # dX[b][k] = sum_o dY[b][o] W[o][k]  (1x1 "conv" over the k axis)
            dx = torch.empty(Bn, Kin, device=x.device, dtype=torch.float32)
            K.conv_nn(B=1, M=Bn, Ck=Nout, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=Kin, Ho=1, Wo=Kin, a=dy,
                      a_bs=0, a_sm=Nout, a_sc=1, a_st=0, x=w, x_bs=0, y=dx, y_bs=0, precision=prec)
        if ctx.needs_input_grad[1]:   # dW[o][k] = sum_b dY[b][o] X[b][k]
            dw = torch.empty(Nout, Kin, device=x.device, dtype=torch.float32)
            K.conv_nn(B=1, M=Nout, Ck=Bn, ks=1, stride=1, pad=0, transposed=False, Hi=1, Wi=Kin, Ho=1, Wo=Kin, a=dy,
                      a_bs=0, a_sm=1, a_sc=Nout, a_st=0, x=x, x_bs=0, y=dw, y_bs=0, precision=prec)
        if has_bias and ctx.needs_input_grad[2]:
            db = K.channel_sum(dy.view(Bn, Nout, 1))
        return dx, dw, db, None


def linear(x, w, bias=None, act=ACT_NONE):
    return LinearFn.apply(x, w, bias, act)


# =====================================================================================================
# losses
# =====================================================================================================
class _ScalarLoss(Function):
    """common backward: upstream is a 0-dim device tensor; the stored gradient is scaled by it on device"""

    @staticmethod
    def _finish(ctx, out, grad):
        ctx.save_for_backward(grad)
        return out.view(())

    @staticmethod
    def _bwd(ctx, gout):
        (grad,) = ctx.saved_tensors
        return K.scale_dev(grad, _c(gout).view(1))


class BceLogitsFn(Function):
    @staticmethod
    def forward(ctx, z, label: float):
        out, dz = K.bce_logits(_c(z), label, True)
        return _ScalarLoss._finish(ctx, out, dz)

    @staticmethod
    def backward(ctx, g):
        return _ScalarLoss._bwd(ctx, g), None


class MseFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        out, da = K.diff_loss("mse", _c(a), _c(b), True)
        return _ScalarLoss._finish(ctx, out, da)

    @staticmethod
    def backward(ctx, g):
        return _ScalarLoss._bwd(ctx, g), None


class L1Fn(Function):
    @staticmethod
    def forward(ctx, a, b):
        out, da = K.diff_loss("l1", _c(a), _c(b), True)
        ctx.b_needs = b.requires_grad
        return _ScalarLoss._finish(ctx, out, da)

    @staticmethod
    def backward(ctx, g):
        da = _ScalarLoss._bwd(ctx, g)
        db = None
        if ctx.needs_input_grad[1]:
            db = K.axpby(da, -1.0, torch.empty_like(da), 0.0)
        return da, db


class TvFn(Function):
    @staticmethod
    def forward(ctx, x, weight: float):
        out, dx = K.tv(_c(x), weight, True)
        return _ScalarLoss._finish(ctx, out, dx)

    @staticmethod
    def backward(ctx, g):
        return _ScalarLoss._bwd(ctx, g), None


def bce_with_logits(z, label: float):
    return BceLogitsFn.apply(z, float(label))


class BceLogitsTargetFn(Function):
    """BCEWithLogitsLoss against a target tensor (mean reduction)"""

    @staticmethod
    def forward(ctx, z, t):
        out, dz, dt = K.bce_logits_target(_c(z), _c(t), True, t.requires_grad)
        ctx.save_for_backward(dz, dt if dt is not None else dz)
        ctx.has_dt = dt is not None
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        dz, dt = ctx.saved_tensors
        g1 = _c(g).view(1)
        return (K.scale_dev(dz, g1) if ctx.needs_input_grad[0] else None,
                K.scale_dev(dt, g1) if (ctx.has_dt and ctx.needs_input_grad[1]) else None)


def bce_with_logits_target(z, t):
    return BceLogitsTargetFn.apply(z, t)


class LeakyFn(Function):
    """LeakyReLU with a slope other than the fused 0.2"""

    @staticmethod
    def forward(ctx, x, slope: float):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.slope = slope
        return K.leaky_fwd(x, slope)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return K.leaky_bwd(x, _c(g), ctx.slope), None


def leaky_relu(x, slope: float):
    return activation(x, ACT_LEAKY) if abs(slope - 0.2) < 1e-12 else LeakyFn.apply(x, float(slope))


def mse_loss(a, b):
    return MseFn.apply(a, b)


def l1_loss(a, b):
    return L1Fn.apply(a, b)


def tv_loss(x, weight: float):
    return TvFn.apply(x, float(weight))


def ssim_value(a, b, window: int = 11):
    """forward-only SSIM mean: the trainer's explicit no-grad evaluation (the reference computes 1 - SSIM at L263 and
    never adds it to loss_G)"""
    with torch.no_grad():
        return K.ssim(_c(a), _c(b), window).view(())


class SsimFn(Function):
    """SSIM._ssim (losses.py:109-136), differentiable w.r.t. both images.  Returns the per-sample means (B,)."""

    @staticmethod
    def forward(ctx, a, b, window: int):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        ctx.window = window
        return K.ssim_samples(a, b, window)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n = a[0].numel()
        gs = torch.empty_like(_c(g))
        K.axpby(_c(g), 1.0 / n, gs, 0.0)                      # d(mean over C,H,W) of a sample's map
        da, db = K.ssim_bwd(a, b, gs, ctx.window, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return da, db, None


def ssim(a, b, window: int = 11, size_average: bool = True):
    """SSIM.forward (losses.py:138-147): mean of the SSIM map (size_average) or one mean per sample"""
    per_sample = SsimFn.apply(a, b, window)
    return mean_of(per_sample) if size_average else per_sample


class MeanFn(Function):
    """mean of a small device vector (composition of the per-sample losses; no host sync)"""

    @staticmethod
    def forward(ctx, v):
        v = _c(v)
        ctx.n = v.numel()
        out = K.dot(v, None)                                  # sum
        return K.axpby(out, 1.0 / ctx.n, torch.empty_like(out), 0.0).view(())

    @staticmethod
    def backward(ctx, g):
        w = torch.full((ctx.n,), 1.0 / ctx.n, device=g.device, dtype=torch.float32)
        return K.scale_dev(w, _c(g).view(1))


def mean_of(v):
    return MeanFn.apply(v)


class AddScalarsFn(Function):
    """sum_k coef_k * t_k of 0-dim device tensors without a host sync (loss_G composition, L267)."""

    @staticmethod
    def forward(ctx, coefs, *terms):
        ctx.coefs = coefs
        out = torch.zeros(1, device=terms[0].device, dtype=torch.float32)
        for c, t in zip(coefs, terms):
            K.axpby(t.reshape(1), float(c), out, 1.0)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        g1 = _c(g).view(1)
        outs = []
        for c in ctx.coefs:
            outs.append(K.axpby(g1, float(c), torch.empty_like(g1), 0.0).view(()))
        return (None, *outs)


def weighted_sum(coefs, terms):
    return AddScalarsFn.apply(tuple(coefs), *terms)


# =====================================================================================================
# small glue ops
# =====================================================================================================
class RepeatChannelsFn(Function):
    """x.repeat(1, n, 1, 1) for a 1-channel image (losses.py:64-65)"""

    @staticmethod
    def forward(ctx, x, n: int):
        x = _c(x)
        B, one, H, W = x.shape
        if one != 1:
            raise L.GandanetError("repeat_channels expects a 1-channel tensor")
        out = torch.empty(B, n, H, W, device=x.device, dtype=torch.float32)
        for i in range(n):
            K.copy_slab(x, out[:, i:i + 1])
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        B, n, H, W = dy.shape
        dx = torch.empty(B, 1, H, W, device=dy.device, dtype=torch.float32)
        K.copy_slab(dy[:, 0:1], dx)
        for i in range(1, n):
            K.copy_slab(dy[:, i:i + 1], dx, accumulate=True)
        return dx, None


def repeat_channels(x, n: int):
    return RepeatChannelsFn.apply(x, n)


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        K.copy_slab(a.view(1, -1, 1), out.view(1, -1, 1))
        K.axpby(b, 1.0, out, 1.0)
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(a, b)


class GlobalAvgPoolFn(Function):
    """AdaptiveAvgPool2d(1) + flatten: (B, C, H, W) -> (B, C)"""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, Cn, H, W = x.shape
        s = K.channel_sum(x.view(1, B * Cn, H * W))
        K.axpby(s, 1.0 / (H * W), s, 0.0)
        ctx.shape = (B, Cn, H, W)
        return s.view(B, Cn)

    @staticmethod
    def backward(ctx, dy):
        B, Cn, H, W = ctx.shape
        shift = K.axpby(_c(dy).view(-1), 1.0 / (H * W), torch.empty(B * Cn, device=dy.device), 0.0)
        zero = torch.zeros(B * Cn, device=dy.device, dtype=torch.float32)
        dx = torch.zeros(1, B * Cn, H * W, device=dy.device, dtype=torch.float32)
        K.affine_act(dx, zero, shift, ACT_NONE, out=dx)   # broadcast shift over the plane
        return dx.view(B, Cn, H, W)


def global_avg_pool(x):
    return GlobalAvgPoolFn.apply(x)


# =====================================================================================================
# attention gates of SqueezeExcitation / CBAMBlock       generator.py:70-101 (exported, not on the train path)
# =====================================================================================================
class BcastMulFn(Function):
    """x (B, C, H, W) * att: mode 0 att (B, C) channel gate (generator.py:84), mode 1 att (B, H*W) spatial gate
    (generator.py:101)"""

    @staticmethod
    def forward(ctx, x, att, mode: int):
        x, att = _c(x), _c(att)
        ctx.save_for_backward(x, att)
        ctx.mode = mode
        return K.bcast_mul(x, att, mode)

    @staticmethod
    def backward(ctx, dy):
        x, att = ctx.saved_tensors
        dy = _c(dy)
        B, Cn = x.shape[0], x.shape[1]
        dx = K.bcast_mul(dy, att, ctx.mode)
        if ctx.mode == 0:
            datt = K.row_dot(dy, x, B * Cn).view_as(att)
        else:
            one = torch.ones(1, device=x.device, dtype=torch.float32)
            datt = K.chan_dot(dy, x, one)[0].view_as(att)
        return dx, datt, None


def gate_channels(x, att):
    return BcastMulFn.apply(x, att, 0)


def gate_pixels(x, att):
    return BcastMulFn.apply(x, att, 1)


class ChanMaxMeanFn(Function):
    """cat([max over channels, mean over channels], 1)  (generator.py:98-100)"""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y, idx = K.chan_maxmean_fwd(x)
        ctx.save_for_backward(idx)
        ctx.channels = x.shape[1]
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return K.chan_maxmean_bwd(_c(dy), idx, ctx.channels)


def chan_maxmean(x):
    return ChanMaxMeanFn.apply(x)
