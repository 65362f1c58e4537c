"""Tensor-level wrappers of the C ABI (no autograd here).

torch is used for device memory and streams only: every function takes CUDA
(ROCm) fp32 tensors, checks layout on the host, allocates outputs/workspaces
with ``torch.empty`` and enqueues the HIP kernels on torch's current stream.
CPU tensors are rejected -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: Tensor, name: str = "tensor", dtype=torch.float32) -> None:
    if not t.is_cuda:
        raise L.GandanetError(f"{name}: expected a GPU tensor (there is no CPU fallback in the product path)")
    if t.dtype != dtype:
        raise L.GandanetError(f"{name}: expected {dtype}, got {t.dtype}")


def _dense(t: Tensor, name: str = "tensor") -> Tensor:
    _chk(t, name)
    if not t.is_contiguous():
        raise L.GandanetError(f"{name}: expected a contiguous tensor, got strides {t.stride()}")
    return t


def _bview(t: Tensor, name: str = "tensor") -> int:
    """Validate a (B, C, H, W) or (B, C, N) tensor whose per-image block is dense (a channel slice of a
    wider slab is fine) and return its batch stride in elements."""
    _chk(t, name)
    if t.dim() == 4:
        b, c, h, w = t.shape
        ok = t.stride(3) == 1 and t.stride(2) == w and t.stride(1) == h * w
        inner = c * h * w
    elif t.dim() == 3:
        b, c, n = t.shape
        ok = t.stride(2) == 1 and t.stride(1) == n
        inner = c * n
    else:
        raise L.GandanetError(f"{name}: expected a 3-D or 4-D tensor")
    if not ok:
        raise L.GandanetError(f"{name}: per-image block must be dense, got strides {t.stride()}")
    bs = t.stride(0) if b > 1 else inner
    if b > 1 and bs < inner:
        raise L.GandanetError(f"{name}: overlapping batch stride")
    return bs


def lib():
    return L.load()


WGRAD_X_NHWC = os.environ.get("GD_WGRAD_X_NHWC", "1") != "0"   # pixel-major bf16 x for wide 3x3 weight gradients (A/B switch)
USE_CONV3X3_FAST = True   # route eligible 3x3/s1/p1 bf16 convs to the LDS-patch kernel (tests flip it to A/B)


# ---- optional HIP-event brackets around the hot kernels (bench.py's live roofline measurement) -------------
PROFILE_ON = False
PROFILE_CONV = os.environ.get("GD_PROFILE_CONV", "0") == "1"     # also bracket every conv call, keyed by its shape
PROFILE: list = []   # (name, start_event, end_event, algorithmic_flops, algorithmic_bytes)


class _Bracket:
    """records a start/end event pair on the stream the kernel is enqueued on (torch's current stream)"""

    def __init__(self, name: str, flops: float, nbytes: float = 0.0):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        if PROFILE_ON:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE_ON:
            self.e1.record()
            PROFILE.append((self.name, self.e0, self.e1, self.flops, self.nbytes))
        return False


class _ConvBracket(_Bracket):
    """opt-in (GD_PROFILE_CONV=1) per-shape bracket of a convolution call: name = kind k s Cin->Cout @HoxWo"""

    def __init__(self, kind, k, stride, cin, cout, ho, wo, b):
        super().__init__(f"conv_{kind}_k{k}s{stride}_{cin}->{cout}@{ho}x{wo}", 2.0 * k * k * cin * cout * ho * wo * b)

    def __enter__(self):
        return super().__enter__() if PROFILE_CONV else self

    def __exit__(self, *exc):
        return super().__exit__(*exc) if PROFILE_CONV else False


def profile_summary():
    """name -> (launches, avg ms, algorithmic flop per launch, algorithmic bytes per launch)"""
    torch.cuda.synchronize()
    agg = {}
    for name, e0, e1, fl, nb in PROFILE:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += fl
        a[3] += nb
    return {k: (v[0], v[1] / v[0], v[2] / v[0], v[3] / v[0]) for k, v in agg.items()}


# ------------------------------------------------------------------------------------------------
# implicit-GEMM "NN" kernel
# ------------------------------------------------------------------------------------------------
def conv_nn(*, B: int, M: int, Ck: int, ks: int, stride: int, pad: int, transposed: bool, Hi: int, Wi: int,
            Ho: int, Wo: int, a: Tensor, a_bs: int, a_sm: int, a_sc: int, a_st: int, x: Tensor, x_bs: int,
            y: Tensor, y_bs: int, precision: int, Mstore: int = 0, in_scale: Optional[Tensor] = None,
            in_shift: Optional[Tensor] = None, in_relu: bool = False, out_layout: int = 0, out_bf16: bool = False,
            ldo: int = 0, alpha: Optional[Tensor] = None, bias: Optional[Tensor] = None,
            res: Optional[Tensor] = None, res_bs: int = 0, act: int = 0, accumulate: bool = False) -> None:
    d = L.ConvDesc()
    d.B, d.M, d.Mstore, d.Ck, d.ks, d.stride, d.pad = B, M, max(M, Mstore), Ck, ks, stride, pad
    d.transposed = int(transposed)
    d.Hi, d.Wi, d.Ho, d.Wo = Hi, Wi, Ho, Wo
    d.a, d.a_bs, d.a_sm, d.a_sc, d.a_st = _ptr(a), a_bs, a_sm, a_sc, a_st
    d.x, d.x_bs = _ptr(x), x_bs
    d.in_scale, d.in_shift, d.in_relu = _ptr(in_scale), _ptr(in_shift), int(in_relu)
    d.y, d.y_bs = _ptr(y), y_bs
    d.out_layout, d.out_bf16, d.ldo = out_layout, int(out_bf16), ldo
    d.alpha, d.bias, d.res, d.res_bs = _ptr(alpha), _ptr(bias), _ptr(res), res_bs
    d.act, d.accumulate, d.precision = act, int(accumulate), precision
    if USE_CONV3X3_FAST and lib().gd_conv3x3_eligible(C.byref(d)):
        nbytes = int(lib().gd_conv3x3_ws_bytes(M, Ck))
        ws = torch.empty(nbytes // 2, device=x.device, dtype=torch.bfloat16)   # packed bf16 weights
        L.check(lib().gd_conv3x3(C.byref(d), _ptr(ws), nbytes, _stream()), "gd_conv3x3")
        return
    L.check(lib().gd_conv2d(C.byref(d), _stream()), "gd_conv2d")


def conv_out_size(h: int, k: int, stride: int, pad: int) -> int:
    return (h + 2 * pad - k) // stride + 1


def conv2d_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int, pad: int, precision: int, *,
               act: int = 0, out: Optional[Tensor] = None, in_scale=None, in_shift=None, in_relu=False,
               accumulate: bool = False) -> Tensor:
    """nn.Conv2d forward, optional fused input affine+ReLU, bias, activation; ``out`` may be a channel slice."""
    xbs = _bview(x, "conv input")
    _dense(w, "conv weight")
    B, Cin, Hi, Wi = x.shape
    Cout, Cin_w, k, k2 = w.shape
    if Cin_w != Cin or k != k2:
        raise L.GandanetError(f"conv2d: weight {tuple(w.shape)} does not match input {tuple(x.shape)}")
    Ho, Wo = conv_out_size(Hi, k, stride, pad), conv_out_size(Wi, k, stride, pad)
    if out is None:
        out = torch.empty(B, Cout, Ho, Wo, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, Cout, Ho, Wo):
        raise L.GandanetError("conv2d: bad out shape")
    with _ConvBracket("fwd", k, stride, Cin, Cout, Ho, Wo, B):
        conv_nn(B=B, M=Cout, Ck=Cin, ks=k, stride=stride, pad=pad, transposed=False, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo,
                a=w, a_bs=0, a_sm=Cin * k * k, a_sc=k * k, a_st=1, x=x, x_bs=xbs, y=out, y_bs=_bview(out, "conv out"),
                precision=precision, in_scale=in_scale, in_shift=in_shift, in_relu=in_relu, bias=bias, act=act,
                accumulate=accumulate)
    return out


def conv2d_dgrad(dy: Tensor, w: Tensor, in_hw: Tuple[int, int], stride: int, pad: int, precision: int, *,
                 out: Optional[Tensor] = None, accumulate: bool = False, alpha: Optional[Tensor] = None) -> Tensor:
    """Gradient of nn.Conv2d w.r.t. its input: transposed gather of dy with the same OIHW weights."""
    dbs = _bview(dy, "conv dy")
    B, Cout, Ho, Wo = dy.shape
    Cout_w, Cin, k, _ = w.shape
    Hi, Wi = in_hw
    if Cout_w != Cout:
        raise L.GandanetError("conv2d_dgrad: weight/out-channel mismatch")
    if out is None:
        out = torch.empty(B, Cin, Hi, Wi, device=dy.device, dtype=torch.float32)
    with _ConvBracket("dgrad", k, stride, Cin, Cout, Ho, Wo, B):
        conv_nn(B=B, M=Cin, Ck=Cout, ks=k, stride=stride, pad=pad, transposed=True, Hi=Ho, Wi=Wo, Ho=Hi, Wo=Wi,
                a=w, a_bs=0, a_sm=k * k, a_sc=Cin * k * k, a_st=1, x=dy, x_bs=dbs, y=out, y_bs=_bview(out, "conv dx"),
                precision=precision, accumulate=accumulate, alpha=alpha)
    return out


def gemm_nt(*, B: int, M: int, N: int, kseg: int, klen: int, a: Tensor, a_bs: int, a_ss: int, lda: int,
            bm: Tensor, b_bs: int, b_ss: int, ldb: int, c: Tensor, c_bs: int, ldc: int, precision: int,
            im2col=None, in_scale=None, in_shift=None, in_relu=False, alpha=None, bias=None,
            accumulate: bool = False, splits: int = 0) -> None:
    d = L.GemmNTDesc()
    d.B, d.M, d.N, d.kseg, d.klen = B, M, N, kseg, klen
    d.a, d.a_bs, d.a_ss, d.lda = _ptr(a), a_bs, a_ss, lda
    d.bm, d.b_bs, d.b_ss, d.ldb = _ptr(bm), b_bs, b_ss, ldb
    if im2col is not None:
        d.im2col = 1
        d.ks, d.stride, d.pad, d.Hi, d.Wi, d.Ho, d.Wo = im2col
    d.in_scale, d.in_shift, d.in_relu = _ptr(in_scale), _ptr(in_shift), int(in_relu)
    d.c, d.c_bs, d.ldc = _ptr(c), c_bs, ldc
    d.alpha, d.bias = _ptr(alpha), _ptr(bias)
    d.accumulate, d.splits, d.precision = int(accumulate), splits, precision
    L.check(lib().gd_gemm_nt(C.byref(d), _stream()), "gd_gemm_nt")


def conv2d_wgrad(dy: Tensor, x: Tensor, k: int, stride: int, pad: int, precision: int, *, in_scale=None,
                 in_shift=None, in_relu=False, alpha: Optional[Tensor] = None) -> Tensor:
    """Gradient of nn.Conv2d w.r.t. its OIHW weight: dW[co][ci,kh,kw] = sum_{b,p} dy[b,co,p] x~[b,ci,p(+)tap].
    ``alpha`` (1x1 convs only): device scalar multiplying the result."""
    if alpha is not None and not (k == 1 and stride == 1 and pad == 0):
        raise L.GandanetError("conv2d_wgrad: alpha is supported for 1x1 convs only")
    dbs, xbs = _bview(dy, "wgrad dy"), _bview(x, "wgrad x")
    B, Cout, Ho, Wo = dy.shape
    _, Cin, Hi, Wi = x.shape
    dw = torch.empty(Cout, Cin, k, k, device=dy.device, dtype=torch.float32)
    with _ConvBracket("wgrad", k, stride, Cin, Cout, Ho, Wo, B):
        return _conv2d_wgrad(dy, x, k, stride, pad, precision, in_scale, in_shift, in_relu, dw, dbs, xbs, B, Cout, Cin, Hi,
                             Wi, Ho, Wo, alpha)


def _conv2d_wgrad(dy, x, k, stride, pad, precision, in_scale, in_shift, in_relu, dw, dbs, xbs, B, Cout, Cin, Hi, Wi, Ho, Wo,
                  alpha=None):
    if USE_CONV3X3_FAST and k == 3 and stride in (1, 2) and pad == 1 and precision == L.PREC_BF16:
        dy16 = x16 = None
        x_ld = 0
        if Cout > 32 and Cin >= 4 * 32:
            # every 32-channel chunk of ci re-reads each dY tile: hand the kernel a bf16 copy made once
            dy16, _ = pack_bf16(dy.view(B, Cout, Ho * Wo), Cout, Ho * Wo, plain_shape=(Cout, Ho * Wo))
            if WGRAD_X_NHWC and in_scale is None and x.dim() == 4:
                # and a pixel-major bf16 copy of x: each of the Cout/BM m-blocks re-stages every patch, as 16-byte
                # copies instead of strided 4-byte gathers (the 2C -> C fuse convs: 12.4 -> 10.5 ms incl. the pack, bench shape)
                x_ld = (Cin + 7) // 8 * 8
                _, x16 = pack_bf16(x, Cin, Hi * Wi, t_shape=(Hi * Wi, x_ld))
        L.check(lib().gd_conv3x3_wgrad(_ptr(dy), dbs, _ptr(dy16), _ptr(x), xbs, _ptr(x16), x_ld, _ptr(in_scale),
                                       _ptr(in_shift), int(in_relu), B, Cout, Cin, Hi, Wi, stride, 0, _ptr(dw), _stream()),
                "gd_conv3x3_wgrad")
        return dw
    if k == 1 and stride == 1 and pad == 0:
        # 1x1: dW = dY X~^T with both operands pixel-contiguous -> plain NT GEMM (float4-staged when aligned); a fused
        # BN affine (+ReLU) of the input is a per-row transform of the B operand
        gemm_nt(B=1, M=Cout, N=Cin, kseg=B, klen=Ho * Wo, a=dy, a_bs=0, a_ss=dbs, lda=Ho * Wo, bm=x, b_bs=0, b_ss=xbs,
                ldb=Hi * Wi, c=dw, c_bs=0, ldc=Cin, precision=precision, in_scale=in_scale, in_shift=in_shift,
                in_relu=in_relu, alpha=alpha)
        return dw
    gemm_nt(B=1, M=Cout, N=Cin * k * k, kseg=B, klen=Ho * Wo, a=dy, a_bs=0, a_ss=dbs, lda=Ho * Wo, bm=x, b_bs=0,
            b_ss=xbs, ldb=0, c=dw, c_bs=0, ldc=Cin * k * k, precision=precision,
            im2col=(k, stride, pad, Hi, Wi, Ho, Wo), in_scale=in_scale, in_shift=in_shift, in_relu=in_relu)
    return dw


def bn_ws(B: int, Cn: int, HW: int, device) -> Tensor:
    n = lib().gd_bn_stats_ws_floats(B, Cn, HW)
    return torch.empty(max(int(n), 1), device=device, dtype=torch.float32)


def channel_sum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    bs = _bview(x, "channel_sum input")
    B, Cn = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if out is None:
        out = torch.empty(Cn, device=x.device, dtype=torch.float32)
    L.check(lib().gd_channel_sum(_ptr(x), bs, B, Cn, HW, _ptr(out), int(accumulate), _ptr(bn_ws(B, Cn, HW, x.device)),
                                 _stream()), "gd_channel_sum")
    return out


# ------------------------------------------------------------------------------------------------
# batch norm
# ------------------------------------------------------------------------------------------------
def bn_stats(x: Tensor, eps: float, momentum: float, running_mean: Optional[Tensor],
             running_var: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    bs = _bview(x, "bn input")
    B, Cn = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    mean = torch.empty(Cn, device=x.device, dtype=torch.float32)
    invstd = torch.empty_like(mean)
    L.check(lib().gd_bn_stats(_ptr(x), bs, B, Cn, HW, eps, momentum, _ptr(mean), _ptr(invstd), _ptr(running_mean),
                              _ptr(running_var), _ptr(bn_ws(B, Cn, HW, x.device)), _stream()), "gd_bn_stats")
    return mean, invstd


def bn_stats_local(x: Tensor) -> Tensor:
    """SyncBN, local half: (C, 3) = (count, mean, M2) of this rank's shard per channel (gd_bn_stats_local)"""
    bs = _bview(x, "bn input")
    B, Cn = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    stats = torch.empty(Cn, 3, device=x.device, dtype=torch.float32)
    L.check(lib().gd_bn_stats_local(_ptr(x), bs, B, Cn, HW, _ptr(stats), _ptr(bn_ws(B, Cn, HW, x.device)), _stream()),
            "gd_bn_stats_local")
    return stats


def bn_stats_merge(stats_all: Tensor, eps: float, momentum: float, running_mean: Optional[Tensor],
                   running_var: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """SyncBN: (world, C, 3) all-gathered records -> mean, invstd of the global batch (+ running statistics)"""
    _dense(stats_all, "gathered BN records")
    world, Cn, _ = stats_all.shape
    mean = torch.empty(Cn, device=stats_all.device, dtype=torch.float32)
    invstd = torch.empty_like(mean)
    L.check(lib().gd_bn_stats_merge(_ptr(stats_all), world, Cn, eps, momentum, _ptr(mean), _ptr(invstd), _ptr(running_mean),
                                    _ptr(running_var), _stream()), "gd_bn_stats_merge")
    return mean, invstd


def bn_act_bwd_dx(dy: Tensor, x: Tensor, scale: Tensor, shift: Tensor, mean: Tensor, invstd: Tensor, dgamma_sum: Tensor,
                  dbeta_sum: Tensor, inv_n: float, act: int, dx: Optional[Tensor] = None, accumulate_dx: bool = False) -> Tensor:
    """SyncBN: dx from the all-reduced per-channel sums (gd_bn_act_bwd_dx)"""
    dbs, xbs = _bview(dy, "bn dy"), _bview(x, "bn x")
    B, Cn = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if dx is None:
        dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        accumulate_dx = False
    L.check(lib().gd_bn_act_bwd_dx(_ptr(dy), dbs, _ptr(x), xbs, _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd),
                                   _ptr(dgamma_sum), _ptr(dbeta_sum), float(inv_n), B, Cn, HW, act, _ptr(dx), _bview(dx),
                                   int(accumulate_dx), _stream()), "gd_bn_act_bwd_dx")
    return dx


def bn_fold(gamma: Tensor, beta: Tensor, mean: Tensor, invstd: Tensor) -> Tuple[Tensor, Tensor]:
    scale, shift = torch.empty_like(gamma), torch.empty_like(gamma)
    L.check(lib().gd_bn_fold(_ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), gamma.numel(), _ptr(scale),
                             _ptr(shift), _stream()), "gd_bn_fold")
    return scale, shift


def bn_fold_eval(gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor, eps: float):
    scale, shift, invstd = torch.empty_like(gamma), torch.empty_like(gamma), torch.empty_like(gamma)
    L.check(lib().gd_bn_fold_eval(_ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), eps,
                                  gamma.numel(), _ptr(scale), _ptr(shift), _ptr(invstd), _stream()), "gd_bn_fold_eval")
    return scale, shift, invstd


def affine_act(x: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], act: int,
               out: Optional[Tensor] = None) -> Tensor:
    bs = _bview(x, "affine input")
    B, Cn = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    L.check(lib().gd_affine_act(_ptr(x), bs, _ptr(scale), _ptr(shift), B, Cn, HW, act, _ptr(out), _bview(out),
                                _stream()), "gd_affine_act")
    return out


def bn_act_bwd(dy: Tensor, x: Tensor, scale: Tensor, shift: Tensor, mean: Tensor, invstd: Tensor, act: int,
               train: bool, dx: Optional[Tensor] = None, accumulate_dx: bool = False, want_dx: bool = True,
               sums_out: Optional[Tensor] = None):
    """``sums_out`` (2, C): dgamma / dbeta are written into its rows (SyncBN all-reduces that one buffer)"""
    dbs, xbs = _bview(dy, "bn dy"), _bview(x, "bn x")
    B, Cn = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if sums_out is not None:
        dgamma, dbeta = _dense(sums_out)[0], sums_out[1]
    else:
        dgamma = torch.empty(Cn, device=x.device, dtype=torch.float32)
        dbeta = torch.empty_like(dgamma)
    if want_dx and dx is None:
        dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    L.check(lib().gd_bn_act_bwd(_ptr(dy), dbs, _ptr(x), xbs, _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), None,
                                B, Cn, HW, act, int(train), _ptr(dgamma), _ptr(dbeta), _ptr(dx) if want_dx else None,
                                _bview(dx) if want_dx else 0, int(accumulate_dx), _ptr(bn_ws(B, Cn, HW, x.device)),
                                _stream()), "gd_bn_act_bwd")
    return dgamma, dbeta, dx


# ------------------------------------------------------------------------------------------------
# resampling
# ------------------------------------------------------------------------------------------------
def bicubic_fwd(x: Tensor, Ho: int, Wo: int, rsh: float, rsw: float) -> Tensor:
    _dense(x, "bicubic input")
    B, Cn, Hi, Wi = x.shape
    y = torch.empty(B, Cn, Ho, Wo, device=x.device, dtype=torch.float32)
    L.check(lib().gd_bicubic_fwd(_ptr(x), B * Cn, Hi, Wi, _ptr(y), Ho, Wo, rsh, rsw, _stream()), "gd_bicubic_fwd")
    return y


def shift_sum9_fwd(u: Tensor, bias: Optional[Tensor]) -> Tensor:
    _dense(u, "tap planes")
    B, nine, H, W = u.shape
    if nine != 9:
        raise L.GandanetError(f"shift_sum9: expected (B, 9, H, W) tap planes, got {tuple(u.shape)}")
    y = torch.empty(B, 1, H, W, device=u.device, dtype=torch.float32)
    L.check(lib().gd_shift_sum9_fwd(_ptr(u), _ptr(bias), _ptr(y), B, H, W, _stream()), "gd_shift_sum9_fwd")
    return y


def shift_sum9_bwd(dy: Tensor) -> Tensor:
    _dense(dy, "dy")
    B, _, H, W = dy.shape
    du = torch.empty(B, 9, H, W, device=dy.device, dtype=torch.float32)
    L.check(lib().gd_shift_sum9_bwd(_ptr(dy), _ptr(du), B, H, W, _stream()), "gd_shift_sum9_bwd")
    return du


def combine_inputs(lr: Tensor, aux: Tensor, s1: float = 0.5, s2: float = 0.25) -> Tensor:
    """cat([bicubic(lr, scale_factor=s1), bicubic(aux, scale_factor=s2)], 1) in one launch
    (GAN_DANet_train.ipynb:L218-224; output size floor(in * scale) like F.interpolate)"""
    _dense(lr, "lr_grace_05"), _dense(aux, "hr_aux")
    B, C1, H1, W1 = lr.shape
    B2, C2, H2, W2 = aux.shape
    Ho, Wo = int(math.floor(H1 * s1)), int(math.floor(W1 * s1))
    if B2 != B or (int(math.floor(H2 * s2)), int(math.floor(W2 * s2))) != (Ho, Wo):
        raise L.GandanetError(f"combine_inputs: {tuple(lr.shape)} x{s1} and {tuple(aux.shape)} x{s2} do not meet")
    out = torch.empty(B, C1 + C2, Ho, Wo, device=lr.device, dtype=torch.float32)
    L.check(lib().gd_combine_inputs(_ptr(lr), C1, H1, W1, 1.0 / s1, _ptr(aux), C2, H2, W2, 1.0 / s2, _ptr(out), B, Ho,
                                    Wo, _stream()), "gd_combine_inputs")
    return out


def hist_match(src: Tensor, ref: Tensor, weight: float) -> Tensor:
    """per-sample mild histogram matching (test.ipynb c1:69-85); src (B, ...), ref (B, ...) fp32 -> fp64 like src"""
    _dense(src, "histogram source"), _dense(ref, "histogram reference")
    B = src.shape[0]
    if ref.shape[0] != B:
        raise L.GandanetError("hist_match: batch sizes differ")
    ns, nt = src[0].numel(), ref[0].numel()
    nbytes = int(lib().gd_hist_match_ws_bytes(ns, nt))
    ws = torch.empty(nbytes, device=src.device, dtype=torch.uint8)
    out = torch.empty(src.shape, device=src.device, dtype=torch.float64)
    L.check(lib().gd_hist_match(_ptr(src), _ptr(ref), B, ns, nt, float(weight), _ptr(out), _ptr(ws), nbytes, _stream()),
            "gd_hist_match")
    return out


def blend_region(gen: Tensor, grace: Tensor, mask: Tensor, region) -> Tensor:
    """in place on ``gen``: gen[.., sr:er, sc:ec] = gen * (1 - mask) + grace * mask"""
    _dense(gen, "blend target"), _dense(grace, "blend source"), _dense(mask, "blend mask")
    sr, er, sc, ec = (int(v) for v in region)
    if gen.shape != grace.shape or tuple(mask.shape) != (er - sr, ec - sc):
        raise L.GandanetError(f"blend_region: shapes {tuple(gen.shape)} / {tuple(grace.shape)} / mask {tuple(mask.shape)}")
    B, Cn, H, W = gen.shape
    L.check(lib().gd_blend_region(_ptr(gen), _ptr(grace), _ptr(mask), B * Cn, H, W, sr, er, sc, ec, _stream()),
            "gd_blend_region")
    return gen


def bicubic_bwd(dy: Tensor, Hi: int, Wi: int, rsh: float, rsw: float) -> Tensor:
    _dense(dy, "bicubic dy")
    B, Cn, Ho, Wo = dy.shape
    dx = torch.empty(B, Cn, Hi, Wi, device=dy.device, dtype=torch.float32)
    L.check(lib().gd_bicubic_bwd(_ptr(dy), B * Cn, Hi, Wi, _ptr(dx), Ho, Wo, rsh, rsw, _stream()), "gd_bicubic_bwd")
    return dx


def bilinear_fwd(x: Tensor, Ho: int, Wo: int, out: Optional[Tensor] = None, accumulate: bool = False,
                 res: Optional[Tensor] = None) -> Tensor:
    """out = bilinear(x) [+ out if accumulate] [or res + bilinear(x) when ``res`` is given]"""
    _dense(x, "bilinear input")
    B, Cn, Hi, Wi = x.shape
    if out is None:
        out = torch.empty(B, Cn, Ho, Wo, device=x.device, dtype=torch.float32)
        accumulate = False
    _dense(out, "bilinear out")
    if res is not None:
        _dense(res, "bilinear residual")
        if res.shape != out.shape:
            raise L.GandanetError("bilinear_fwd: residual shape mismatch")
    L.check(lib().gd_bilinear_fwd(_ptr(x), B * Cn, Hi, Wi, _ptr(out), Ho, Wo, int(accumulate), _ptr(res), _stream()),
            "gd_bilinear_fwd")
    return out


def bilinear_bwd(dy: Tensor, Hi: int, Wi: int) -> Tensor:
    _dense(dy, "bilinear dy")
    B, Cn, Ho, Wo = dy.shape
    dx = torch.empty(B, Cn, Hi, Wi, device=dy.device, dtype=torch.float32)
    L.check(lib().gd_bilinear_bwd(_ptr(dy), B * Cn, Hi, Wi, _ptr(dx), Ho, Wo, _stream()), "gd_bilinear_bwd")
    return dx


def maxpool2_fwd(x: Tensor) -> Tensor:
    _dense(x, "maxpool input")
    B, Cn, Hi, Wi = x.shape
    y = torch.empty(B, Cn, Hi // 2, Wi // 2, device=x.device, dtype=torch.float32)
    L.check(lib().gd_maxpool2_fwd(_ptr(x), B * Cn, Hi, Wi, _ptr(y), _stream()), "gd_maxpool2_fwd")
    return y


def maxpool2_bwd(x: Tensor, dy: Tensor) -> Tensor:
    _dense(x), _dense(dy)
    B, Cn, Hi, Wi = x.shape
    dx = torch.empty_like(x)
    L.check(lib().gd_maxpool2_bwd(_ptr(x), _ptr(dy), B * Cn, Hi, Wi, _ptr(dx), _stream()), "gd_maxpool2_bwd")
    return dx


# ------------------------------------------------------------------------------------------------
# pointwise / reductions
# ------------------------------------------------------------------------------------------------
def act_fwd(x: Tensor, act: int, out: Optional[Tensor] = None) -> Tensor:
    _dense(x)
    out = torch.empty_like(x) if out is None else out
    L.check(lib().gd_act_fwd(_ptr(x), _ptr(out), x.numel(), act, _stream()), "gd_act_fwd")
    return out


def act_bwd(y: Tensor, dy: Tensor, act: int) -> Tensor:
    _dense(y), _dense(dy)
    dx = torch.empty_like(dy)
    L.check(lib().gd_act_bwd(_ptr(y), _ptr(dy), _ptr(dx), y.numel(), act, _stream()), "gd_act_bwd")
    return dx


def axpby(x: Tensor, a: float, y: Tensor, b: float) -> Tensor:
    """y = a*x + b*y (in place on y)"""
    _dense(x), _dense(y)
    L.check(lib().gd_axpby(_ptr(x), a, _ptr(y), b, x.numel(), _stream()), "gd_axpby")
    return y


def scale_dev(x: Tensor, s: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """out (+)= s * x with s a 1-element device tensor"""
    _dense(x)
    if out is None:
        out = torch.empty_like(x)
        accumulate = False
    L.check(lib().gd_scale_dev(_ptr(x), _ptr(s), _ptr(out), x.numel(), int(accumulate), _stream()), "gd_scale_dev")
    return out


def copy_slab(src: Tensor, dst: Tensor, accumulate: bool = False) -> Tensor:
    sbs, dbs = _bview(src, "copy src"), _bview(dst, "copy dst")
    if src.shape != dst.shape:
        raise L.GandanetError("copy_slab: shape mismatch")
    L.check(lib().gd_copy_slab(_ptr(src), sbs, _ptr(dst), dbs, src.shape[0], src[0].numel(), int(accumulate),
                               _stream()), "gd_copy_slab")
    return dst


def softmax_rows(x: Tensor, sign: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    _dense(x)
    out = torch.empty_like(x) if out is None else out
    cols = x.shape[-1]
    L.check(lib().gd_softmax_rows(_ptr(x), _ptr(out), x.numel() // cols, cols, sign, _stream()), "gd_softmax_rows")
    return out


def softmax_rows_bwd(p: Tensor, dp: Tensor, sign: float = 1.0) -> Tensor:
    _dense(p), _dense(dp)
    dx = torch.empty_like(p)
    cols = p.shape[-1]
    L.check(lib().gd_softmax_rows_bwd(_ptr(p), _ptr(dp), _ptr(dx), p.numel() // cols, cols, sign, _stream()),
            "gd_softmax_rows_bwd")
    return dx


def _red_ws(device) -> Tensor:
    return torch.empty(2048, device=device, dtype=torch.float32)


def dot(a: Tensor, b: Optional[Tensor], out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    _dense(a)
    if b is not None:
        _dense(b)
    if out is None:
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        accumulate = False
    L.check(lib().gd_dot(_ptr(a), _ptr(b), a.numel(), _ptr(out), int(accumulate), _ptr(_red_ws(a.device)), _stream()),
            "gd_dot")
    return out


def transpose(s: Tensor) -> Tensor:
    """(B, R, C) -> (B, C, R)"""
    _dense(s)
    B, R, Cc = s.shape
    t = torch.empty(B, Cc, R, device=s.device, dtype=torch.float32)
    L.check(lib().gd_transpose(_ptr(s), _ptr(t), B, R, Cc, _stream()), "gd_transpose")
    return t


def add_transpose(a: Tensor) -> Tensor:
    _dense(a)
    B, n, _ = a.shape
    out = torch.empty_like(a)
    L.check(lib().gd_add_transpose(_ptr(a), _ptr(out), B, n, _stream()), "gd_add_transpose")
    return out


def copy_rows(src: Tensor, s_bs: int, s_ld: int, dst: Tensor, d_bs: int, d_ld: int, B: int, R: int, Cc: int):
    L.check(lib().gd_copy_rows(_ptr(src), s_bs, s_ld, _ptr(dst), d_bs, d_ld, B, R, Cc, _stream()), "gd_copy_rows")
    return dst


# ---- losses ---------------------------------------------------------------------------------------
def bce_logits(z: Tensor, label: float, want_grad: bool):
    _dense(z)
    out = torch.empty(1, device=z.device, dtype=torch.float32)
    dz = torch.empty_like(z) if want_grad else None
    L.check(lib().gd_bce_logits(_ptr(z), z.numel(), label, _ptr(out), _ptr(dz), _ptr(_red_ws(z.device)), _stream()),
            "gd_bce_logits")
    return out, dz


def bce_logits_target(z: Tensor, t: Tensor, want_dz: bool, want_dt: bool):
    _dense(z), _dense(t)
    out = torch.empty(1, device=z.device, dtype=torch.float32)
    dz = torch.empty_like(z) if want_dz else None
    dt = torch.empty_like(t) if want_dt else None
    L.check(lib().gd_bce_logits_target(_ptr(z), _ptr(t), z.numel(), _ptr(out), _ptr(dz), _ptr(dt),
                                       _ptr(_red_ws(z.device)), _stream()), "gd_bce_logits_target")
    return out, dz, dt


def leaky_fwd(x: Tensor, slope: float) -> Tensor:
    _dense(x)
    y = torch.empty_like(x)
    L.check(lib().gd_leaky_fwd(_ptr(x), _ptr(y), x.numel(), float(slope), _stream()), "gd_leaky_fwd")
    return y


def leaky_bwd(x: Tensor, dy: Tensor, slope: float) -> Tensor:
    _dense(x), _dense(dy)
    dx = torch.empty_like(x)
    L.check(lib().gd_leaky_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.numel(), float(slope), _stream()), "gd_leaky_bwd")
    return dx


def diff_loss(kind: str, a: Tensor, b: Tensor, want_grad: bool):
    _dense(a), _dense(b)
    if a.shape != b.shape:
        raise L.GandanetError(f"{kind}: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    da = torch.empty_like(a) if want_grad else None
    fn = lib().gd_mse if kind == "mse" else lib().gd_l1
    L.check(fn(_ptr(a), _ptr(b), a.numel(), _ptr(out), _ptr(da), _ptr(_red_ws(a.device)), _stream()), f"gd_{kind}")
    return out, da


def tv(x: Tensor, weight: float, want_grad: bool):
    _dense(x)
    B, Cn, H, W = x.shape
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x) if want_grad else None
    L.check(lib().gd_tv(_ptr(x), B, Cn, H, W, weight, _ptr(out), _ptr(dx), _ptr(_red_ws(x.device)), _stream()), "gd_tv")
    return out, dx


def ssim(a: Tensor, b: Tensor, window: int) -> Tensor:
    _dense(a), _dense(b)
    B, Cn, H, W = a.shape
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    L.check(lib().gd_ssim(_ptr(a), _ptr(b), B * Cn, H, W, window, _ptr(out), _ptr(_red_ws(a.device)), _stream()),
            "gd_ssim")
    return out


def ssim_samples(a: Tensor, b: Tensor, window: int) -> Tensor:
    """per-sample SSIM means (B,)"""
    _dense(a), _dense(b)
    B, Cn, H, W = a.shape
    out = torch.empty(B, device=a.device, dtype=torch.float32)
    L.check(lib().gd_ssim_samples(_ptr(a), _ptr(b), B, Cn, H, W, window, _ptr(out), _ptr(_red_ws(a.device)), _stream()),
            "gd_ssim_samples")
    return out


def ssim_bwd(a: Tensor, b: Tensor, gscale: Tensor, window: int, need_a: bool = True, need_b: bool = True):
    """gradients of sum_s gscale[s] * sum_pixels ssim_map[s] w.r.t. a and b"""
    _dense(a), _dense(b), _dense(gscale)
    B, Cn, H, W = a.shape
    coef = torch.empty(4 * a.numel(), device=a.device, dtype=torch.float32)
    da = torch.empty_like(a) if need_a else None
    db = torch.empty_like(b) if need_b else None
    L.check(lib().gd_ssim_bwd(_ptr(a), _ptr(b), _ptr(gscale), B, Cn, H, W, window, _ptr(coef), _ptr(da), _ptr(db),
                              _stream()), "gd_ssim_bwd")
    return da, db


def adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float, beta2: float, eps: float,
          weight_decay: float, grad_scale: float = 1.0) -> None:
    for t in (p, g, m, v):
        _dense(t, "adamw tensor")
    L.check(lib().gd_adamw(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), step, lr, beta1, beta2, eps, weight_decay,
                           grad_scale, _stream()), "gd_adamw")


# ---- PAM helpers ------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634    # gd_pam_flash_* take q pre-scaled by log2(e) (include/gandanet.h)


def pack_bf16(s: Tensor, R: int, Cc: int, *, scale: Optional[Tensor] = None, scale_imm: float = 1.0, plain_shape=None,
              t_shape=None, perm16: bool = False, ones_row: int = -1, f16: bool = False):
    """s: (B, R, Cc)-like fp32 block (per-image dense).  Returns (plain, transposed) 16-bit tensors (or None):
    bf16, or IEEE fp16 with ``f16`` (the fused PAM kernels' fp16 operand mode)."""
    sbs = _bview(s, "pack input")
    B = s.shape[0]
    plain = tr = None
    rp = ldp = ccp = ldt = 0
    dt = torch.float16 if f16 else torch.bfloat16
    if plain_shape is not None:
        rp, ldp = plain_shape
        plain = torch.empty(B, rp, ldp, device=s.device, dtype=dt)
    if t_shape is not None:
        ccp, ldt = t_shape
        tr = torch.empty(B, ccp, ldt, device=s.device, dtype=dt)
    L.check(lib().gd_pack_16(_ptr(s), sbs, B, R, Cc, _ptr(scale), float(scale_imm), _ptr(plain), rp, ldp, _ptr(tr),
                             ccp, ldt, int(perm16), int(ones_row), int(f16), _stream()), "gd_pack_16")
    return plain, tr


def pack_nhwc16_affine(x: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], relu: bool) -> Tensor:
    """(B, C, H, W) fp32 (per-image dense, possibly a channel slice of a slab) -> (B, H*W, C) bf16 pixel-major copy of
    max(0, scale[c] x + shift[c]) (the BatchNorm + ReLU prologue of a dense layer); C % 8 == 0, H*W % 4 == 0"""
    sbs = _bview(x, "pack input")
    B, Cn, H, W = x.shape
    out = torch.empty(B, H * W, Cn, device=x.device, dtype=torch.bfloat16)
    L.check(lib().gd_pack_16_affine(_ptr(x), sbs, B, Cn, H * W, _ptr(scale), _ptr(shift), int(relu), None, 0, 0, _ptr(out),
                                    H * W, Cn, 0, _stream()), "gd_pack_16_affine")
    return out


def augment_d4(x: Tensor, ops: Tensor, noise: Optional[Tensor] = None, noise_scale: float = 0.05) -> Tensor:
    """per-sample flip / flip / rot90 (+ noise) of a (B, C, H, W) batch; ``ops`` int32 (B,) op words (gandanet.h)"""
    _dense(x, "augment input")
    _chk(ops, "augment ops", torch.int32)
    B, Cn, H, W = x.shape
    if ops.numel() != B or (noise is not None and _dense(noise, "noise").shape != x.shape):
        raise L.GandanetError("augment_d4: ops must hold one word per sample and noise must match the batch")
    y = torch.empty_like(x)
    L.check(lib().gd_augment_d4(_ptr(x), _ptr(y), B, Cn, H, W, _ptr(ops), _ptr(noise), float(noise_scale), _stream()),
            "gd_augment_d4")
    return y


def bcast_mul(x: Tensor, att: Tensor, mode: int) -> Tensor:
    """x (B, C, H, W) dense times a channel gate att (B, C) [mode 0] or a spatial gate att (B, H*W) [mode 1]"""
    _dense(x), _dense(att)
    B, Cn = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if att.numel() != (B * Cn if mode == 0 else B * HW):
        raise L.GandanetError(f"bcast_mul: gate of {att.numel()} elements does not match x {tuple(x.shape)} (mode {mode})")
    y = torch.empty_like(x)
    L.check(lib().gd_bcast_mul(_ptr(x), _ptr(att), _ptr(y), B, Cn, HW, mode, _stream()), "gd_bcast_mul")
    return y


def row_dot(a: Tensor, b: Tensor, rows: int) -> Tensor:
    _dense(a), _dense(b)
    n = a.numel() // rows
    out = torch.empty(rows, device=a.device, dtype=torch.float32)
    L.check(lib().gd_row_dot(_ptr(a), _ptr(b), _ptr(out), rows, n, _stream()), "gd_row_dot")
    return out


def chan_maxmean_fwd(x: Tensor):
    _dense(x)
    B, Cn, H, W = x.shape
    y = torch.empty(B, 2, H, W, device=x.device, dtype=torch.float32)
    idx = torch.empty(B, H * W, device=x.device, dtype=torch.int32)
    L.check(lib().gd_chan_maxmean_fwd(_ptr(x), _ptr(y), _ptr(idx), B, Cn, H * W, _stream()), "gd_chan_maxmean_fwd")
    return y, idx


def chan_maxmean_bwd(dy: Tensor, idx: Tensor, Cn: int) -> Tensor:
    _dense(dy)
    B, _, H, W = dy.shape
    dx = torch.empty(B, Cn, H, W, device=dy.device, dtype=torch.float32)
    L.check(lib().gd_chan_maxmean_bwd(_ptr(dy), _ptr(idx), _ptr(dx), B, Cn, H * W, _stream()), "gd_chan_maxmean_bwd")
    return dx


def chan_dot(a: Tensor, o: Tensor, gamma: Tensor):
    abs_, obs = _bview(a), _bview(o)
    B, Cn = a.shape[0], a.shape[1]
    N = a[0, 0].numel()
    d_raw = torch.empty(B, N, device=a.device, dtype=torch.float32)
    delta = torch.empty_like(d_raw)
    L.check(lib().gd_chan_dot(_ptr(a), abs_, _ptr(o), obs, B, Cn, N, _ptr(gamma), _ptr(d_raw), _ptr(delta), _stream()),
            "gd_chan_dot")
    return d_raw, delta


def pam_f16_scale(dout: Tensor, gamma: Tensor, delta: Tensor) -> Tensor:
    """power-of-two scale of the fp16 PAM backward (gd_pam_f16_scale): returns scales = [gamma * 2^k, 2^-k]; ``delta`` is
    multiplied by 2^k in place"""
    bs = _bview(dout, "dOut")
    B, Cn = dout.shape[0], dout.shape[1]
    N = dout[0, 0].numel()
    scales = torch.empty(2, device=dout.device, dtype=torch.float32)
    L.check(lib().gd_pam_f16_scale(_ptr(dout), bs, B, Cn, N, _ptr(gamma), _ptr(_dense(delta)), _ptr(scales),
                                   _ptr(_red_ws(dout.device)), _stream()), "gd_pam_f16_scale")
    return scales


PAM_NOMAX = os.environ.get("GD_PAM_NOMAX", "1") != "0"      # let the forward drop the running maximum where a bound allows


def pam_key_sqnorm_max(kt: Tensor, N: int, f16: bool = False) -> Tensor:
    """max_j |k_j|^2 per image of the packed keys kt (B, Npad, 32) (gd_pam_key_sqnorm_max)"""
    B, Npad = kt.shape[0], kt.shape[1]
    out = torch.empty(B, device=kt.device, dtype=torch.float32)
    L.check(lib().gd_pam_key_sqnorm_max(_ptr(kt), B, N, Npad, int(f16), _ptr(out), _stream()), "gd_pam_key_sqnorm_max")
    return out


def pam_flash_fwd(qt, kt, v, B, N, Npad, Cn, Cp, gamma, x, out, o_attn, lse, r_alg: int = 32, v_ones: bool = False,
                  f16: bool = False, k_sqmax: Optional[Tensor] = None):
    # algorithmic (unpadded) work: 2 N^2 (r + C) per image
    with _Bracket("pam_flash_fwd", 2.0 * N * N * (r_alg + Cn) * B):
        L.check(lib().gd_pam_flash_fwd(_ptr(qt), _ptr(kt), _ptr(v), B, N, Npad, Cn, Cp, int(v_ones), int(f16),
                                       _ptr(gamma), _ptr(x), _bview(x), _ptr(out), _bview(out), _ptr(o_attn), _ptr(lse),
                                       _ptr(k_sqmax), _stream()), "gd_pam_flash_fwd")


PAM_SHIFT = os.environ.get("GD_PAM_SHIFT", "0") != "0"       # bf16 forward: sampled per-query shift + max-free sweep (opt-in: see DESIGN 5.2)
PAM_SHIFT_SAMPLES = int(os.environ.get("GD_PAM_SHIFT_SAMPLES", "256"))


def pam_flash_fwd_shift(qt, kt, v, B, N, Npad, Cn, Cp, gamma, x, out, o_attn, lse, r_alg: int = 32, v_ones: bool = False,
                        nsample: Optional[int] = None, return_flags: bool = False):
    """gd_pam_flash_fwd_shift: bf16 forward with the sampled-shift max-free sweep + fallback pass for flagged workgroups"""
    _bf(qt, "qt"), _bf(kt, "kt")
    nbytes = int(lib().gd_pam_fwd_shift_ws_bytes(B, Npad))
    ws = torch.empty(nbytes // 4, device=qt.device, dtype=torch.int32)
    with _Bracket("pam_flash_fwd", 2.0 * N * N * (r_alg + Cn) * B):
        L.check(lib().gd_pam_flash_fwd_shift(_ptr(qt), _ptr(kt), _ptr(v), B, N, Npad, Cn, Cp, int(v_ones), _ptr(gamma), _ptr(x),
                                             _bview(x), _ptr(out), _bview(out), _ptr(o_attn), _ptr(lse),
                                             int(nsample or PAM_SHIFT_SAMPLES), _ptr(ws), nbytes, _stream()),
                "gd_pam_flash_fwd_shift")
    if return_flags:
        return ws[B * Npad:].clone()           # 1 = the workgroup (256 queries) was redone with the running maximum
    return None


# backward form (gandanet.h GD_PAM_BWD_*): GD_PAM_BWD=0 K64 + fp32 atomics for dQ (default), 1 K64 + bf16 parts
# (bitwise reproducible), 2 the round-1 K32 kernel with bf16 parts, 3 two kernels without scratch.
# set_deterministic(True) moves the default from 0 to 1.
PAM_BWD_FORM = int(os.environ.get("GD_PAM_BWD", "0"))
PAM_SCRATCH_CAP = 40 << 30                                         # scratch of one call: at most 40 GiB
DETERMINISTIC = False


def set_deterministic(on: bool) -> None:
    """bitwise-reproducible kernels wherever a faster order-dependent form exists: PAM dQ through bf16 parts instead of
    fp32 atomics, split-K GEMMs and 3x3 weight gradients unsplit (gd_set_deterministic)"""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    lib().gd_set_deterministic(int(DETERMINISTIC))


def pam_bwd_form() -> int:
    return L.PAM_BWD_K64_PARTS if (DETERMINISTIC and PAM_BWD_FORM == 0) else PAM_BWD_FORM


def pam_flash_bwd(qt, kt, kn, vt, dot_, lse, delta, B, N, Npad, Cp, dqn, dkn, dv, r_alg: int = 32, c_alg: int = 0,
                  f16: bool = False, form: Optional[int] = None, out_bs: int = 0):
    """out_bs != 0 (form 0 only): dqn / dkn / dv are row blocks of one (B, rows, Npad) buffer with that batch stride"""
    form = pam_bwd_form() if form is None else form
    scratch, scratch_bytes = None, 0
    per_image = int(lib().gd_pam_bwd_scratch_bytes(Npad, form))
    if per_image:
        nslices = -(-(B * per_image) // PAM_SCRATCH_CAP)              # even slices of the batch that fit the cap
        images = max(1, -(-B // nslices))
        scratch_bytes = per_image * images
        scratch = torch.empty(scratch_bytes, device=dqn.device, dtype=torch.uint8)
    # algorithmic work of the backward = 2x forward: 4 N^2 (r + C) per image
    with _Bracket("pam_flash_bwd", 4.0 * N * N * (r_alg + (c_alg or Cp)) * B):
        L.check(lib().gd_pam_flash_bwd(_ptr(qt), _ptr(kt), _ptr(kn), _ptr(vt), _ptr(dot_), _ptr(lse), _ptr(delta), B,
                                       N, Npad, Cp, int(f16), int(form), _ptr(dqn), _ptr(dkn), _ptr(dv), int(out_bs),
                                       _ptr(scratch), scratch_bytes, _stream()), "gd_pam_flash_bwd")


# =====================================================================================================
# NHWC bf16 kernels (frozen VGG19 feature stack of PerceptualLoss)
# =====================================================================================================
def _bf(t: Tensor, name: str = "tensor") -> Tensor:
    _chk(t, name, torch.bfloat16)
    if not t.is_contiguous():
        raise L.GandanetError(f"{name}: expected a contiguous tensor, got strides {t.stride()}")
    return t


def conv3x3_nhwc_pack(w: Tensor, transposed: bool) -> Tensor:
    """w (Cout, Cin, 3, 3) fp32 -> packed bf16 operator (transposed: 0 forward, 1 stride-1 data gradient, 2 stride-2
    data gradient)"""
    _dense(w, "conv weight")
    Cout, Cin = w.shape[0], w.shape[1]
    M, Kc = (Cin, Cout) if transposed else (Cout, Cin)
    nbytes = int(lib().gd_conv3x3_ws_bytes(M, Kc))
    ws = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    L.check(lib().gd_conv3x3_nhwc_pack(_ptr(w), Cout, Cin, int(transposed), _ptr(ws), nbytes, _stream()),
            "gd_conv3x3_nhwc_pack")
    return ws


def conv3x3_nhwc(x: Tensor, wpack: Tensor, bias: Optional[Tensor], M: int, relu: bool = False,
                 mask: Optional[Tensor] = None, res: Optional[Tensor] = None, split: bool = False) -> Tensor:
    """x (B, H, W, K) bf16 -> (B, H, W, M) bf16: [mask > 0] * act(conv3x3(x) + bias) + res; split: K = 3 Cin physical
    channels, wpack from split weights, y / mask / res with 3 M channels ([hi | lo | hi])"""
    _bf(x, "nhwc conv input")
    B, H, W, Kc = x.shape
    y = torch.empty(B, H, W, (3 if split else 1) * M, device=x.device, dtype=torch.bfloat16)
    for t, nm in ((mask, "mask"), (res, "res")):
        if t is not None and (_bf(t, nm).shape != y.shape):
            raise L.GandanetError(f"conv3x3_nhwc: {nm} {tuple(t.shape)} does not match the output {tuple(y.shape)}")
    with _Bracket("conv3x3_nhwc", 2.0 * 9 * Kc * M * H * W * B):
        L.check(lib().gd_conv3x3_nhwc(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(mask), _ptr(res), _ptr(y), B, H, W, Kc, M,
                                      int(relu), int(split), _stream()), "gd_conv3x3_nhwc")
    return y


def conv3x3_nhwc_f32out(x16: Tensor, wpack: Tensor, bias: Optional[Tensor], M: int, H: int, W: int, relu: bool = False,
                        out: Optional[Tensor] = None) -> Tensor:
    """x16 (B, H*W, K) pixel-major bf16 (pack_bf16 transposed output) -> conv3x3 s1 p1 (+bias, ReLU) as (B, M, H, W) fp32"""
    _bf(x16, "nhwc conv input")
    B, HW, Kc = x16.shape
    if HW != H * W:
        raise L.GandanetError(f"conv3x3_nhwc_f32out: input {tuple(x16.shape)} is not a {H} x {W} image")
    if out is None:
        out = torch.empty(B, M, H, W, device=x16.device, dtype=torch.float32)
    with _ConvBracket("nhwc_f32out", 3, 1, Kc, M, H, W, B):
        L.check(lib().gd_conv3x3_nhwc_f32out(_ptr(x16), _ptr(wpack), _ptr(bias), _ptr(out), _bview(out, "conv out"), B, H, W, Kc,
                                             M, int(relu), _stream()), "gd_conv3x3_nhwc_f32out")
    return out


def conv3x3_wgrad_packed(dy16: Tensor, x16: Tensor, H: int, W: int, stride: int = 1) -> Tensor:
    """weight gradient from the two 16-bit copies: dy16 (B, Cout, Ho*Wo) channel-major, x16 (B, H*W, Cin) pixel-major"""
    _bf(dy16, "dy16"), _bf(x16, "x16")
    B, Cout = dy16.shape[0], dy16.shape[1]
    Cin = x16.shape[2]
    dw = torch.empty(Cout, Cin, 3, 3, device=dy16.device, dtype=torch.float32)
    with _ConvBracket("wgrad_packed", 3, stride, Cin, Cout, (H - 1) // stride + 1, (W - 1) // stride + 1, B):
        L.check(lib().gd_conv3x3_wgrad(None, 0, _ptr(dy16), None, 0, _ptr(x16), Cin, None, None, 0, B, Cout, Cin, H, W, stride,
                                       0, _ptr(dw), _stream()), "gd_conv3x3_wgrad")
    return dw


# ---- split-bf16 ("x3") operands: set_precision("mixed") -------------------------------------------------------------
HLH, HL, H_ONLY = 0b010, 0b10, 0b0      # copy patterns of pack_split: bit j set = copy j holds the lo part


def pack_split(x: Tensor, *, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, relu: bool = False,
               want_plain: bool = False, want_tr: bool = True, mask: Optional[Tensor] = None):
    """(B, C, H, W) / (B, C, N) fp32 (per-image dense) -> split-bf16 copies (gd_pack_16_split):
    tr    (B, N, 3 C) pixel-major [hi | lo | hi]: the forward / data-gradient operand against gd_split3_weights
    plain (2, B, C, N) channel-major, [0] hi and [1] lo: the dY operand of the weight gradient's three launches
    mask: a ReLU output of x's shape -- the ReLU backward fused into the pack (x packed as zero where mask <= 0)"""
    sbs = _bview(x, "pack input")
    if mask is not None and tuple(mask.shape[:2]) + (mask[0, 0].numel(),) != tuple(x.shape[:2]) + (x[0, 0].numel(),):
        raise L.GandanetError(f"pack_split: mask {tuple(mask.shape)} does not match {tuple(x.shape)}")
    B, Cn = x.shape[0], x.shape[1]
    N = x[0, 0].numel()
    plain = torch.empty(2, B, Cn, N, device=x.device, dtype=torch.bfloat16) if want_plain else None
    tr = torch.empty(B, N, 3 * Cn, device=x.device, dtype=torch.bfloat16) if want_tr else None
    L.check(lib().gd_pack_16_split_masked(_ptr(x), sbs, B, Cn, N, _ptr(scale), _ptr(shift), int(relu),
                                          _ptr(plain), Cn * N, N, B * Cn * N, 2, HL,
                                          _ptr(tr), N * 3 * Cn, 3 * Cn, Cn, 3, HLH,
                                          _ptr(mask), 0 if mask is None else _bview(mask, "pack mask"), _stream()),
            "gd_pack_16_split")
    return plain, tr


def split3_weights(w: Tensor, axis: int) -> Tensor:
    """(Cout, Cin, k, k) fp32 -> [hi ; hi ; lo] along ``axis`` (1: input channels, forward; 0: output channels, the
    data gradient's contraction axis)"""
    _dense(w, "conv weight")
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    if axis == 1:
        out = torch.empty(Cout, 3 * Cin, *w.shape[2:], device=w.device, dtype=torch.float32)
        L.check(lib().gd_split3_weights(_ptr(w), Cout, Cin, taps, _ptr(out), _stream()), "gd_split3_weights")
    else:
        out = torch.empty(3 * Cout, Cin, *w.shape[2:], device=w.device, dtype=torch.float32)
        L.check(lib().gd_split3_weights(_ptr(w), 1, Cout, Cin * taps, _ptr(out), _stream()), "gd_split3_weights")
    return out


def conv3x3_wgrad_x3(dy2: Tensor, x3: Tensor, H: int, W: int) -> Tensor:
    """weight gradient from split operands: dy2 (2, B, Cout, N) [hi, lo] channel-major, x3 (B, N, 3 Cin) [hi | lo | hi]
    pixel-major (the forward's own pack): dW = dy_hi (x) x_hi + dy_lo (x) x_hi + dy_hi (x) x_lo, three launches adding
    into one dW"""
    _bf(dy2, "dy2"), _bf(x3, "x3")
    _, B, Cout, _ = dy2.shape
    Cin = x3.shape[2] // 3
    dw = torch.empty(Cout, Cin, 3, 3, device=dy2.device, dtype=torch.float32)
    with _ConvBracket("wgrad_x3", 3, 1, Cin, Cout, H, W, B):
        for j, (dpart, xoff) in enumerate(((0, 0), (1, 0), (0, Cin))):
            L.check(lib().gd_conv3x3_wgrad(None, 0, dy2[dpart].data_ptr(), None, 0, x3.data_ptr() + 2 * xoff, 3 * Cin, None, None,
                                           0, B, Cout, Cin, H, W, 1, int(j > 0), _ptr(dw), _stream()), "gd_conv3x3_wgrad")
    return dw


# ---- Discriminator1 on pixel-major bf16 (discriminator.py:57-77) ------------------------------------------------------
def _half_up(n: int) -> int:
    return (n - 1) // 2 + 1


def disc_stem_fwd(img: Tensor, w: Tensor, bias: Optional[Tensor], slope: float, split: bool = False) -> Tensor:
    """conv1: fp32 NCHW image -> LeakyReLU(conv3x3 s2 p1 + bias) as (B, Ho, Wo, Co) bf16; ``split`` (here and in the
    functions below): pixel-major tensors carry 3 C channels per pixel, the values split [hi | lo | hi] (operand mode "x3")"""
    _dense(img, "stem image"), _dense(w, "stem weight")
    B, Ci, H, W = img.shape
    Co = w.shape[0]
    if w.shape[1] != Ci:
        raise L.GandanetError(f"disc_stem_fwd: weight {tuple(w.shape)} does not match image {tuple(img.shape)}")
    y = torch.empty(B, _half_up(H), _half_up(W), (3 if split else 1) * Co, device=img.device, dtype=torch.bfloat16)
    L.check(lib().gd_disc_stem_fwd(_ptr(img), B, Ci, H, W, _ptr(w), _ptr(bias), Co, float(slope), _ptr(y), int(split), _stream()),
            "gd_disc_stem_fwd")
    return y


def disc_stem_wgrad(g: Tensor, img: Tensor, want_bias: bool = True, split: bool = False):
    """(dw (Co, Ci, 3, 3), db (Co) or None) of conv1 from the pre-activation gradient g (B, Ho, Wo, Co) bf16"""
    _bf(g, "stem gradient"), _dense(img, "stem image")
    B, Ci, H, W = img.shape
    Co = g.shape[3] // (3 if split else 1)
    if g.shape[:3] != (B, _half_up(H), _half_up(W)):
        raise L.GandanetError(f"disc_stem_wgrad: gradient {tuple(g.shape)} does not match image {tuple(img.shape)}")
    dw = torch.empty(Co, Ci, 3, 3, device=g.device, dtype=torch.float32)
    db = torch.empty(Co, device=g.device, dtype=torch.float32) if want_bias else None
    L.check(lib().gd_disc_stem_wgrad(_ptr(g), _ptr(img), B, Ci, H, W, Co, _ptr(dw), _ptr(db), int(split), _stream()),
            "gd_disc_stem_wgrad")
    return dw, db


def disc_stem_dgrad(g: Tensor, w: Tensor, H: int, W: int, split: bool = False) -> Tensor:
    _bf(g, "stem gradient"), _dense(w, "stem weight")
    B, Ho, Wo, Co = g.shape
    Co //= 3 if split else 1
    Ci = w.shape[1]
    if (Ho, Wo) != (_half_up(H), _half_up(W)):
        raise L.GandanetError(f"disc_stem_dgrad: gradient {tuple(g.shape)} does not match a {H} x {W} image")
    dimg = torch.empty(B, Ci, H, W, device=g.device, dtype=torch.float32)
    L.check(lib().gd_disc_stem_dgrad(_ptr(g), B, Ci, H, W, _ptr(w), Co, _ptr(dimg), int(split), _stream()), "gd_disc_stem_dgrad")
    return dimg


def conv3x3_nhwc_s2(x: Tensor, wpack: Tensor, bias: Optional[Tensor], M: int, act: int, slope: float = 0.2,
                    split: bool = False) -> Tensor:
    """x (B, H, W, K) bf16 -> act(conv3x3 s2 p1 + bias) (B, Ho, Wo, M) bf16; act 0 none / 1 ReLU / 2 LeakyReLU(slope);
    split: K = 3 Cin physical channels in, 3 M out, wpack from split3_weights(w, 1)"""
    _bf(x, "nhwc conv input")
    B, H, W, Kc = x.shape
    y = torch.empty(B, _half_up(H), _half_up(W), (3 if split else 1) * M, device=x.device, dtype=torch.bfloat16)
    with _Bracket("conv3x3_nhwc_s2", 2.0 * 9 * Kc * M * y.shape[1] * y.shape[2] * B):
        L.check(lib().gd_conv3x3_nhwc_s2(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(y), B, H, W, Kc, M, int(act), float(slope),
                                         int(split), _stream()), "gd_conv3x3_nhwc_s2")
    return y


def conv3x3_nhwc_s2_dgrad(dy: Tensor, wpack_t: Tensor, act_out: Tensor, slope: float = 0.2, split: bool = False) -> Tensor:
    """dy (B, Ho, Wo, M) bf16 -> dx (B, H, W, K) bf16 = convT(dy) * LeakyReLU'(act_out); shape taken from act_out;
    split: dy carries M = 3 Cout physical channels, act_out and dx 3 K, wpack_t from split3_weights(w, 0)"""
    _bf(dy, "nhwc conv gradient"), _bf(act_out, "activation output")
    B, H, W, Kc = act_out.shape
    Kc //= 3 if split else 1
    M = dy.shape[3]
    if dy.shape[:3] != (B, _half_up(H), _half_up(W)):
        raise L.GandanetError(f"conv3x3_nhwc_s2_dgrad: dy {tuple(dy.shape)} does not match input {tuple(act_out.shape)}")
    dx = torch.empty_like(act_out)
    with _Bracket("conv3x3_nhwc_s2_dgrad", 2.0 * 9 * Kc * M * dy.shape[1] * dy.shape[2] * B):
        L.check(lib().gd_conv3x3_nhwc_s2_dgrad(_ptr(dy), _ptr(wpack_t), _ptr(act_out), float(slope), _ptr(dx), B, H, W, Kc, M,
                                               int(split), _stream()), "gd_conv3x3_nhwc_s2_dgrad")
    return dx


def nhwc_flatten_fwd(y: Tensor, split: bool = False) -> Tensor:
    """(B, h, w, C) bf16 -> (B, C*h*w) fp32 in the (c, h, w) order of NCHW .flatten(1)"""
    _bf(y, "flatten input")
    B, H, W, Cc = y.shape
    Cc //= 3 if split else 1
    f = torch.empty(B, Cc * H * W, device=y.device, dtype=torch.float32)
    L.check(lib().gd_nhwc_flatten_fwd(_ptr(y), B, H * W, Cc, _ptr(f), int(split), _stream()), "gd_nhwc_flatten_fwd")
    return f


def nhwc_flatten_bwd(df: Tensor, y: Tensor, slope: float, split: bool = False) -> Tensor:
    _bf(y, "flatten input"), _dense(df, "flatten gradient")
    B, H, W, Cc = y.shape
    Cc //= 3 if split else 1
    if df.numel() * (3 if split else 1) != y.numel():
        raise L.GandanetError(f"nhwc_flatten_bwd: gradient {tuple(df.shape)} does not match {tuple(y.shape)}")
    g = torch.empty_like(y)
    L.check(lib().gd_nhwc_flatten_bwd(_ptr(df), _ptr(y), float(slope), B, H * W, Cc, _ptr(g), int(split), _stream()),
            "gd_nhwc_flatten_bwd")
    return g


def nhwc_to_nchw16(g: Tensor, want_sum: bool, split: bool = False):
    """(B, h, w, C) bf16 -> ((B, C, h, w) bf16, per-channel sums (C) fp32 or None); split: g carries 3 C channels
    [hi | lo | hi] and the result is (2, B, C, h, w), hi then lo, with the sums of hi + lo"""
    _bf(g, "nhwc gradient")
    B, H, W, Cc = g.shape
    Cc //= 3 if split else 1
    gt = torch.empty((2, B, Cc, H, W) if split else (B, Cc, H, W), device=g.device, dtype=torch.bfloat16)
    cs = torch.empty(Cc, device=g.device, dtype=torch.float32) if want_sum else None
    L.check(lib().gd_nhwc_to_nchw16(_ptr(g), B, H * W, Cc, _ptr(gt), _ptr(cs), int(split), _stream()), "gd_nhwc_to_nchw16")
    return gt, cs


def conv3x3_wgrad_nhwc(g: Tensor, x: Tensor, stride: int, want_bias: bool, split: bool = False):
    """Weight (and bias) gradient of a 3x3 / pad 1 conv whose input x (B, H, W, Cin) and output gradient g
    (B, Ho, Wo, Cout) are pixel-major bf16: g goes channel-major once (fused with the bias sums), x is staged as is.
    split: both carry [hi | lo | hi]; dW = g_hi (x) x_hi + g_lo (x) x_hi + g_hi (x) x_lo in three accumulating launches."""
    _bf(g, "nhwc gradient"), _bf(x, "nhwc input")
    B, H, W, Cin = x.shape
    Cout = g.shape[3]
    gt, db = nhwc_to_nchw16(g, want_bias, split)
    if split:
        Cin //= 3
        Cout //= 3
        dw = torch.empty(Cout, Cin, 3, 3, device=g.device, dtype=torch.float32)
        with _ConvBracket("wgrad_nhwc_x3", 3, stride, Cin, Cout, g.shape[1], g.shape[2], B):
            for j, (dpart, xoff) in enumerate(((0, 0), (1, 0), (0, Cin))):
                L.check(lib().gd_conv3x3_wgrad(None, 0, gt[dpart].data_ptr(), None, 0, x.data_ptr() + 2 * xoff, 3 * Cin, None, None,
                                               0, B, Cout, Cin, H, W, stride, int(j > 0), _ptr(dw), _stream()), "gd_conv3x3_wgrad")
        return dw, db
    dw = torch.empty(Cout, Cin, 3, 3, device=g.device, dtype=torch.float32)
    with _ConvBracket("wgrad_nhwc", 3, stride, Cin, Cout, g.shape[1], g.shape[2], B):
        L.check(lib().gd_conv3x3_wgrad(None, 0, _ptr(gt), None, 0, _ptr(x), Cin, None, None, 0, B, Cout, Cin, H, W, stride,
                                       0, _ptr(dw), _stream()), "gd_conv3x3_wgrad")
    return dw, db


def nhwc_stem_fwd(img: Tensor, w: Tensor, bias: Optional[Tensor], relu: bool, split: bool = False) -> Tensor:
    """``split`` (here and in the nhwc_* functions below): the pixel-major tensors carry 3 C channels per pixel, the values
    split [hi | lo | hi] (operand mode "x3")"""
    _dense(img, "stem image"), _dense(w, "stem weight")
    B, Ci, H, W = img.shape
    Co = w.shape[0]
    if w.shape[1] != Ci:
        raise L.GandanetError(f"nhwc_stem_fwd: weight {tuple(w.shape)} does not match image {tuple(img.shape)}")
    y = torch.empty(B, H, W, (3 if split else 1) * Co, device=img.device, dtype=torch.bfloat16)
    L.check(lib().gd_nhwc_stem_fwd(_ptr(img), B, Ci, H, W, _ptr(w), _ptr(bias), Co, int(relu), _ptr(y), int(split), _stream()),
            "gd_nhwc_stem_fwd")
    return y


def nhwc_stem_bwd(g: Tensor, w: Tensor, split: bool = False) -> Tensor:
    _bf(g, "stem gradient"), _dense(w, "stem weight")
    B, H, W, Co = g.shape
    Co //= 3 if split else 1
    Ci = w.shape[1]
    dimg = torch.empty(B, Ci, H, W, device=g.device, dtype=torch.float32)
    L.check(lib().gd_nhwc_stem_bwd(_ptr(g), B, Ci, H, W, _ptr(w), Co, _ptr(dimg), int(split), _stream()), "gd_nhwc_stem_bwd")
    return dimg


def nhwc_maxpool2_fwd(x: Tensor, split: bool = False) -> Tensor:
    _bf(x, "pool input")
    B, H, W, Cn = x.shape
    y = torch.empty(B, H // 2, W // 2, Cn, device=x.device, dtype=torch.bfloat16)
    L.check(lib().gd_nhwc_maxpool2_fwd(_ptr(x), B, H, W, Cn // (3 if split else 1), _ptr(y), int(split), _stream()),
            "gd_nhwc_maxpool2_fwd")
    return y


def nhwc_maxpool2_bwd(x: Tensor, dy: Tensor, relu_mask: bool, split: bool = False) -> Tensor:
    _bf(x, "pool input"), _bf(dy, "pool dy")
    B, H, W, Cn = x.shape
    dx = torch.empty_like(x)
    L.check(lib().gd_nhwc_maxpool2_bwd(_ptr(x), _ptr(dy), B, H, W, Cn // (3 if split else 1), int(relu_mask), _ptr(dx), int(split),
                                       _stream()),
            "gd_nhwc_maxpool2_bwd")
    return dx


def nhwc_l1(a: Tensor, b: Tensor, out: Tensor, accumulate: bool, split: bool = False) -> None:
    """out[0] (+)= mean |a - b| over bf16 tensors of equal shape (split: (..., 3 C) tensors of hi + lo values)"""
    _bf(a, "l1 a"), _bf(b, "l1 b")
    ws = torch.empty(1024, device=a.device, dtype=torch.float32)
    sc = a.shape[-1] // 3 if split else 0
    L.check(lib().gd_nhwc_l1(_ptr(a), _ptr(b), a.numel() // (3 if split else 1), _ptr(out), int(accumulate), _ptr(ws), sc,
                             _stream()), "gd_nhwc_l1")


def nhwc_l1_grad(a: Tensor, b: Tensor, upstream: Tensor, relu_mask: bool, split: bool = False) -> Tensor:
    _bf(a, "l1 a"), _bf(b, "l1 b"), _dense(upstream, "upstream gradient")
    g = torch.empty_like(a)
    sc = a.shape[-1] // 3 if split else 0
    L.check(lib().gd_nhwc_l1_grad(_ptr(a), _ptr(b), a.numel() // (3 if split else 1), _ptr(upstream), int(relu_mask), _ptr(g), sc,
                                  _stream()),
            "gd_nhwc_l1_grad")
    return g
