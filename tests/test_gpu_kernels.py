"""GPU parity, op level: each HIP kernel (through the C ABI) against the CPU oracle on seeded inputs.
fp32 mode = exact f32 MFMA -> tight tolerances; bf16 mode = bf16 MFMA operands, fp32 accumulate ->
tolerance ~ 2^-8 relative to the output scale (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, assert_close, bf16_round, rell2, relmax, seeded

pytestmark = pytest.mark.gpu

FP32_TOL = 2e-5
BF16_TOL = 2e-2


@pytest.fixture(scope="module")
def gd():
    import gan_danet_amd as g
    from gan_danet_amd import _lib
    _lib.load()
    return g


def _ops():
    from gan_danet_amd import kern, ops
    return ops, kern


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, bias
    (2, 8, 16, 16, 64, 3, 1, 1, False),     # stem
    (2, 88, 8, 8, 24, 3, 1, 1, True),       # dense layer (Cin not a multiple of 32)
    (1, 184, 16, 16, 23, 1, 1, 0, True),    # PAM query projection
    (2, 64, 16, 16, 1, 3, 1, 1, True),      # final conv (Cout = 1)
    (2, 1, 32, 32, 64, 3, 2, 1, True),      # D conv1 (Cin = 1, stride 2)
    (2, 64, 16, 16, 128, 3, 2, 1, True),    # D conv2
    (1, 3, 20, 12, 64, 3, 1, 1, True),      # VGG first conv, non-square image
    (1, 40, 9, 7, 136, 3, 1, 1, False),     # ragged: odd sizes, M > 128
    (1, 16, 13, 11, 32, 4, 2, 1, True),     # SRGAND-style 4x4 stride 2
]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(gd, case, prec):
    ops, _ = _ops()
    B, Cin, H, W, Cout, k, s, p, bias = case
    x = seeded((B, Cin, H, W), 1)
    w = seeded((Cout, Cin, k, k), 2, 1.0 / math.sqrt(Cin * k * k))
    b = seeded((Cout,), 3, 0.1) if bias else None
    if prec == "bf16":
        x, w = bf16_round(x), bf16_round(w)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.leaky_relu(F.conv2d(xr, wr, br, stride=s, padding=p), 0.2)
    go = seeded(tuple(yr.shape), 4)
    if prec == "bf16":
        go = bf16_round(go)
    yr.backward(go)

    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True) if bias else None
    with gd.precision(prec):
        y = ops.conv2d(xg, wg, bg, s, p, ops.ACT_LEAKY)
        y.backward(go.to(DEV))
    tol = FP32_TOL if prec == "fp32" else BF16_TOL
    assert_close(y, yr, tol, "y")
    # in bf16 mode dy*act' is re-rounded to bf16 inside the kernels: compare in L2
    assert_close(xg.grad, xr.grad, tol, "dx", rell2 if prec == "bf16" else relmax)
    assert_close(wg.grad, wr.grad, tol, "dw", rell2 if prec == "bf16" else relmax)
    if bias:
        assert_close(bg.grad, br.grad, 1e-4, "db")


@pytest.mark.parametrize("shape", [(2, 136, 12, 20, 48, False, False), (1, 368, 16, 32, 184, False, False),
                                   (2, 128, 9, 8, 40, True, True)])
def test_wide_conv3x3_through_the_pixel_major_kernel(gd, shape):
    """wide 3x3 / stride 1 / pad 1 convs in 16-bit mode (ops._wide3x3: DANetAttention's fuse conv, generator.py:108) run on
    a pixel-major bf16 copy of x through the NHWC kernel with an fp32 NCHW result; forward, both gradients and the input
    given as a channel slice of a wider buffer, against ATen on the bf16-rounded operands and against the fp32-NCHW
    patch kernel (GD_CONV_WIDE_NHWC off), which rounds the same operands"""
    ops, _ = _ops()
    B, Cin, H, W, Cout, bias, relu = shape
    assert ops._wide3x3(torch.empty(B, Cin, H, W), torch.empty(Cout, Cin, 3, 3), 1, 1, ops.ACT_NONE, ops.L.PREC_BF16)
    wide = bf16_round(seeded((B, Cin + 8, H, W), 11))
    x = wide[:, 4:4 + Cin]                                   # batch stride (Cin + 8) H W
    w = bf16_round(seeded((Cout, Cin, 3, 3), 12, 1.0 / math.sqrt(Cin * 9)))
    b = seeded((Cout,), 13, 0.1) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=1)
    yr = F.relu(yr) if relu else yr
    go = bf16_round(seeded(tuple(yr.shape), 14))
    yr.backward(go)
    res = {}
    for on in (True, False):
        old = ops.CONV_WIDE_NHWC
        ops.CONV_WIDE_NHWC = on
        try:
            wd = wide.to(DEV).requires_grad_(True)
            wg = w.to(DEV).requires_grad_(True)
            with gd.precision("bf16"):
                y = ops.conv2d(wd[:, 4:4 + Cin], wg, b.to(DEV) if bias else None, 1, 1, ops.ACT_RELU if relu else ops.ACT_NONE)
                y.backward(go.to(DEV))
        finally:
            ops.CONV_WIDE_NHWC = old
        res[on] = (y.detach(), wd.grad[:, 4:4 + Cin], wg.grad)
        assert_close(y, yr, BF16_TOL, "y")
        assert_close(res[on][1], xr.grad, BF16_TOL, "dx", rell2)
        assert_close(res[on][2], wr.grad, BF16_TOL, "dw", rell2)
        assert wd.grad[:, :4].abs().max().item() == 0 and wd.grad[:, 4 + Cin:].abs().max().item() == 0
    for a, bb, nm in zip(res[True], res[False], ("y", "dx", "dw")):
        assert_close(a, bb, 2e-3, f"pixel-major vs patch kernel: {nm}", rell2)


def test_conv2d_slab_views_and_prologue(gd):
    """conv reading a channel slice of a slab with the fused BN-affine+ReLU prologue, writing another slice"""
    ops, K = _ops()
    from gan_danet_amd import _lib as L
    B, Ctot, H, W = 2, 48, 8, 8
    slab = seeded((B, Ctot, H, W), 5).to(DEV)
    w = seeded((8, 24, 3, 3), 6, 0.1).to(DEV)
    bias = seeded((8,), 7).to(DEV)
    sc, sh = (seeded((24,), 8).abs() + 0.5).to(DEV), seeded((24,), 9, 0.3).to(DEV)
    ref_in = F.relu(slab[:, :24].cpu() * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None])
    ref = F.conv2d(ref_in, w.cpu(), bias.cpu(), padding=1)
    before = slab.clone()
    K.conv2d_fwd(slab[:, :24], w, bias, 1, 1, L.PREC_FP32, in_scale=sc, in_shift=sh, in_relu=True, out=slab[:, 40:48])
    assert_close(slab[:, 40:48], ref, FP32_TOL, "slab conv")
    assert torch.equal(slab[:, :40], before[:, :40]), "conv wrote outside its slice"
    # weight gradient through the same prologue
    dy = seeded((B, 8, H, W), 10).to(DEV)
    dw = K.conv2d_wgrad(dy, slab[:, :24], 3, 1, 1, L.PREC_FP32, in_scale=sc, in_shift=sh, in_relu=True)
    xin = ref_in.clone()
    wr = w.cpu().clone().requires_grad_(True)
    F.conv2d(xin, wr, None, padding=1).backward(dy.cpu())
    assert_close(dw, wr.grad, FP32_TOL, "dw prologue")


@pytest.mark.parametrize("shape", [(2, 24, 8, 8), (3, 64, 33, 17), (2, 5, 150, 130)])
@pytest.mark.parametrize("train", [True, False])
def test_batchnorm_act(gd, shape, train):
    ops, _ = _ops()
    B, Cn, H, W = shape
    x = seeded(shape, 11) * 2 + 3.0     # offset mean: exercises the cancellation-safe statistics
    gmm, bta = 1 + 0.1 * seeded((Cn,), 12), 0.1 * seeded((Cn,), 13)
    rm, rv = 0.05 * seeded((Cn,), 14), 1 + 0.2 * seeded((Cn,), 15) ** 2
    bn = torch.nn.BatchNorm2d(Cn)
    with torch.no_grad():
        bn.weight.copy_(gmm); bn.bias.copy_(bta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(train)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(bn(xr))
    go = seeded(shape, 16)
    yr.backward(go)

    xg = x.to(DEV).requires_grad_(True)
    g, b = gmm.to(DEV).requires_grad_(True), bta.to(DEV).requires_grad_(True)
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    y = ops.batch_norm_act(xg, g, b, rmg, rvg, train, 0.1, 1e-5, ops.ACT_RELU)
    y.backward(go.to(DEV))
    assert_close(y, yr, 2e-5, "y")
    assert_close(xg.grad, xr.grad, 1e-4, "dx")
    assert_close(g.grad, bn.weight.grad, 1e-4, "dgamma")
    assert_close(b.grad, bn.bias.grad, 1e-4, "dbeta")
    assert_close(rmg, bn.running_mean, 1e-5, "running_mean")
    assert_close(rvg, bn.running_var, 1e-5, "running_var")


@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (1, 2, 45, 22), (2, 4, 16, 32), (1, 2, 40, 100), (2, 1, 27, 131)])
def test_bicubic_up2(gd, shape):
    ops, _ = _ops()
    x = seeded(shape, 21)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bicubic", align_corners=False)
    go = seeded(tuple(yr.shape), 22)
    yr.backward(go)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.bicubic_up2(xg)
    y.backward(go.to(DEV))
    assert_close(y, yr, 1e-5, "y")
    assert_close(xg.grad, xr.grad, 1e-5, "dx")


def test_bicubic_downscale_forward(gd):
    """the train-loop preamble F.interpolate(scale_factor=0.25 / 0.5, 'bicubic') (GAN_DANet_train.ipynb:L226,L231)"""
    _, K = _ops()
    x = seeded((2, 3, 32, 48), 23)
    for sf in (0.5, 0.25):
        yr = F.interpolate(x, scale_factor=sf, mode="bicubic", align_corners=False)
        y = K.bicubic_fwd(x.to(DEV), yr.shape[2], yr.shape[3], 1 / sf, 1 / sf)
        assert_close(y, yr, 1e-5, f"bicubic x{sf}")


@pytest.mark.parametrize("shape,out", [((2, 3, 8, 8), (32, 32)), ((1, 2, 45, 22), (180, 88)), ((1, 2, 7, 9), (14, 27)),
                                       ((1, 2, 40, 100), (160, 400)), ((2, 1, 27, 131), (108, 524))])
def test_bilinear(gd, shape, out):
    ops, _ = _ops()
    x = seeded(shape, 24)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=out, mode="bilinear", align_corners=False)
    go = seeded(tuple(yr.shape), 25)
    yr.backward(go)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.BilinearFn.apply(xg, out[0], out[1])
    y.backward(go.to(DEV))
    assert_close(y, yr, 1e-5, "y")
    assert_close(xg.grad, xr.grad, 1e-5, "dx")


def test_maxpool2(gd):
    ops, _ = _ops()
    x = seeded((2, 5, 12, 20), 26)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    go = seeded(tuple(yr.shape), 27)
    yr.backward(go)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.MaxPool2Fn.apply(xg)
    y.backward(go.to(DEV))
    assert torch.equal(y.cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("dims", [(2, 8192, 1024), (3, 1024, 1), (5, 777, 130)])
def test_linear(gd, dims, prec):
    ops, _ = _ops()
    Bn, Kin, Nout = dims
    x, w, b = seeded((Bn, Kin), 31), seeded((Nout, Kin), 32, 1 / math.sqrt(Kin)), seeded((Nout,), 33, 0.1)
    if prec == "bf16":
        x, w = bf16_round(x), bf16_round(w)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.leaky_relu(F.linear(xr, wr, br), 0.2)
    go = seeded((Bn, Nout), 34)
    if prec == "bf16":
        go = bf16_round(go)
    yr.backward(go)
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    with gd.precision(prec):
        y = ops.linear(xg, wg, bg, ops.ACT_LEAKY)
        y.backward(go.to(DEV))
    tol = 5e-5 if prec == "fp32" else BF16_TOL
    m = relmax if prec == "fp32" else rell2
    assert_close(y, yr, tol, "y")
    assert_close(xg.grad, xr.grad, tol, "dx", m)
    assert_close(wg.grad, wr.grad, tol, "dw", m)
    assert_close(bg.grad, br.grad, 1e-4, "db")


def test_softmax_rows(gd):
    _, K = _ops()
    x = seeded((37, 184), 41, 5.0)
    for sign in (1.0, -1.0):
        p = K.softmax_rows(x.to(DEV), sign)
        pr = torch.softmax(sign * x, -1)
        assert_close(p, pr, 1e-5, "softmax")
        dp = seeded((37, 184), 42)
        xr = x.clone().requires_grad_(True)
        torch.softmax(sign * xr, -1).backward(dp)
        dx = K.softmax_rows_bwd(p, dp.to(DEV), sign)
        assert_close(dx, xr.grad, 1e-4, "softmax bwd")


def test_losses(gd, golden_dir):
    from gpu_util import load_golden
    ops, K = _ops()
    fx = load_golden(golden_dir, "losses_32x32")
    a = fx["a"].to(DEV).requires_grad_(True)
    b = fx["b"].to(DEV)
    tv = ops.tv_loss(a, 1e-5)
    (g,) = torch.autograd.grad(tv, a)
    assert_close(tv, fx["tv"], 1e-5, "tv")
    assert_close(g, fx["gtv"], 1e-5, "dtv")
    assert_close(ops.ssim_value(a, b, 11), fx["ssim"], 1e-4, "ssim")
    # differentiable SSIM (reference fixture: value and gradient w.r.t. img1 from losses.SSIM(11, True))
    ss = ops.ssim(a, b, 11, True)
    assert ss.requires_grad
    (gs,) = torch.autograd.grad(ss, a)
    assert_close(ss, fx["ssim"], 1e-4, "ssim (differentiable)")
    assert_close(gs, fx["gssim"], 1e-4, "dssim/dimg1 vs reference fixture", rell2)
    z = fx["z"].to(DEV).requires_grad_(True)
    l1 = ops.bce_with_logits(z, 1.0)
    assert_close(l1, fx["bce1"], 1e-5, "bce1")
    (gz,) = torch.autograd.grad(l1, z)
    zr = fx["z"].clone().requires_grad_(True)
    F.binary_cross_entropy_with_logits(zr, torch.ones_like(zr)).backward()
    assert_close(gz, zr.grad, 1e-5, "dbce")
    assert_close(ops.bce_with_logits(z, 0.0), fx["bce0"], 1e-5, "bce0")
    ms = ops.mse_loss(a, b)
    assert_close(ms, fx["mse"], 1e-5, "mse")
    (gm,) = torch.autograd.grad(ms, a)
    assert_close(gm, 2 * (fx["a"] - fx["b"]) / fx["a"].numel(), 1e-5, "dmse")
    l1v = ops.l1_loss(a, b)
    assert_close(l1v, (fx["a"] - fx["b"]).abs().mean(), 1e-5, "l1")
    # weighted sum of scalars with a scalar upstream (the loss_G composition)
    tot = ops.weighted_sum([0.5, 2.0], [ops.mse_loss(a, b), ops.tv_loss(a, 1e-5)])
    (gt,) = torch.autograd.grad(tot, a)
    assert_close(gt, 0.5 * gm.cpu() + 2.0 * fx["gtv"], 1e-5, "weighted sum grad")


def test_adamw_matches_torch(gd):
    from gan_danet_amd import AdamW
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    pr = torch.nn.Parameter(p0.clone())
    pg = torch.nn.Parameter(p0.clone().to(DEV))
    o_r = torch.optim.AdamW([pr], lr=4e-4, betas=(0.5, 0.999), weight_decay=1e-4)
    o_g = AdamW([pg], lr=4e-4, betas=(0.5, 0.999), weight_decay=1e-4)
    for t in range(5):
        g = torch.randn(1000)
        pr.grad, pg.grad = g.clone(), g.clone().to(DEV)
        o_r.step(); o_g.step()
    assert_close(pg, pr, 1e-6, "adamw params")


def test_transpose_and_pack(gd):
    _, K = _ops()
    s = seeded((2, 23, 70), 51)
    t = K.transpose(s.to(DEV))
    assert torch.equal(t.cpu(), s.transpose(1, 2).contiguous())
    plain, tr = K.pack_bf16(s.to(DEV), 23, 70, plain_shape=(32, 128), t_shape=(128, 32))
    ref = torch.zeros(2, 32, 128); ref[:, :23, :70] = s
    assert torch.equal(plain.float().cpu(), ref.to(torch.bfloat16).float())
    assert torch.equal(tr.float().cpu(), ref.transpose(1, 2).to(torch.bfloat16).float())


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("R,Cc,Rp,ldp,ldt", [(23, 200, 32, 256, 32), (184, 456, 192, 512, 192), (70, 64, 72, 64, 72)])
def test_pack16_tile64_kernel_aligned_shapes(gd, R, Cc, Rp, ldp, ldt, f16):
    """the 64 x 64-tile pack (16-byte accesses; taken when Cc % 4 == 0 and the leading dimensions % 8 == 0) against the
    definition: zero padding, scale, the row of ones, the perm16 key order of the plain copy, both 16-bit types, the
    source as a channel slice of a wider tensor; and gd_pack_16_affine (per-row affine + ReLU first)"""
    from gan_danet_amd import kern as K
    B = 2
    wide = seeded((B, R + 5, Cc), 61)
    src = wide[:, 3:3 + R]
    dt = torch.float16 if f16 else torch.bfloat16
    ones = Rp - 1 if R < Rp else -1
    plain, tr = K.pack_bf16(wide.to(DEV)[:, 3:3 + R], R, Cc, scale_imm=1.5, plain_shape=(Rp, ldp), t_shape=(ldp, ldt),
                            perm16=True, ones_row=ones, f16=f16)
    ref = torch.zeros(B, Rp, ldp)
    ref[:, :R, :Cc] = 1.5 * src
    if ones >= 0:
        ref[:, ones, :] = 1.0
    reft = torch.zeros(B, ldp, ldt)
    reft[:, :, :Rp] = ref.transpose(1, 2)[:, :, :min(Rp, ldt)] if Rp <= ldt else ref.transpose(1, 2)[:, :, :ldt]
    c = torch.arange(ldp)
    perm = (c & ~15) | ((c & 3) | ((c & 4) << 1) | ((c & 8) >> 1))     # plain[.., perm(c)] = value of column c
    refp = torch.zeros_like(ref)
    refp[:, :, perm] = ref
    assert torch.equal(plain.float().cpu(), refp.to(dt).float())
    assert torch.equal(tr.float().cpu(), reft.to(dt).float())
    if not f16 and R % 8 == 0:
        sc, sh = seeded((R,), 62, 0.5), seeded((R,), 63, 0.3)
        xin = wide.to(DEV).view(B, R + 5, Cc // 4, 4)[:, 3:3 + R]          # (B, R, H, W) channel slice
        out = K.pack_nhwc16_affine(xin, sc.to(DEV), sh.to(DEV), True)
        refa = torch.relu(src * sc[None, :, None] + sh[None, :, None]).transpose(1, 2)
        assert_close(out.float(), refa, 8e-3, "affine + ReLU pack (the kernel fuses the multiply-add: one bf16 ulp)")


@pytest.mark.parametrize("shape", [(2, 72, 24, 64, 184), (1, 368, 16, 32, 184), (1, 64, 40, 96, 64), (2, 3, 33, 70, 64),
                                   # Cout <= 32 (dense layers): weight gradient with waves = ci chunks (2, 4, 3+2, 3)
                                   (2, 64, 16, 32, 24), (1, 112, 9, 40, 24), (1, 160, 8, 32, 24), (1, 88, 8, 64, 32)])
def test_conv3x3_patch_kernel_matches_generic_and_oracle(gd, shape):
    """the LDS-patch 3x3 kernel against the generic implicit-GEMM kernel (same bf16 operands -> agreement to
    fp32 accumulation order) and against the oracle; forward with fused BN-affine+ReLU prologue, bias, ReLU,
    and the data gradient."""
    ops, K = _ops()
    from gan_danet_amd import _lib as L
    B, Cin, H, W, Cout = shape
    x = bf16_round(seeded((B, Cin, H, W), 61)).to(DEV)
    w = bf16_round(seeded((Cout, Cin, 3, 3), 62, 1.0 / math.sqrt(Cin * 9))).to(DEV)
    bias = seeded((Cout,), 63, 0.1).to(DEV)
    sc, sh = (seeded((Cin,), 64).abs() * 0.5 + 0.75).to(DEV), seeded((Cin,), 65, 0.2).to(DEV)
    outs = {}
    for fast in (True, False):
        K.USE_CONV3X3_FAST = fast
        try:
            y = K.conv2d_fwd(x, w, bias, 1, 1, L.PREC_BF16, act=ops.ACT_RELU, in_scale=sc, in_shift=sh, in_relu=True)
            dy = bf16_round(seeded((B, Cout, H, W), 66)).to(DEV)
            dx = K.conv2d_dgrad(dy, w, (H, W), 1, 1, L.PREC_BF16)
            dw = K.conv2d_wgrad(dy, x, 3, 1, 1, L.PREC_BF16, in_scale=sc, in_shift=sh, in_relu=True)
        finally:
            K.USE_CONV3X3_FAST = True
        outs[fast] = (y, dx, dw)
    assert_close(outs[True][2], outs[False][2].cpu(), 1e-4, "patch vs generic wgrad")
    assert_close(outs[True][0], outs[False][0].cpu(), 1e-5, "patch vs generic fwd")
    assert_close(outs[True][1], outs[False][1].cpu(), 1e-5, "patch vs generic dgrad")
    xin = F.relu(x.cpu() * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None])
    yr = F.relu(F.conv2d(xin, w.cpu(), bias.cpu(), padding=1))
    assert_close(outs[True][0], yr, BF16_TOL, "patch fwd vs oracle")
    dxr = torch.nn.grad.conv2d_input((B, Cin, H, W), w.cpu(), dy.cpu(), padding=1)
    assert_close(outs[True][1], dxr, BF16_TOL, "patch dgrad vs oracle", rell2)
    dwr = torch.nn.grad.conv2d_weight(xin, (Cout, Cin, 3, 3), dy.cpu(), padding=1)
    assert_close(outs[True][2], dwr, BF16_TOL, "patch wgrad vs oracle", rell2)


@pytest.mark.parametrize("shape", [(2, 1, 64, 96, 64), (2, 64, 32, 64, 128), (1, 128, 17, 35, 256), (1, 40, 9, 130, 24)])
def test_conv3x3_stride2_patch_kernel(gd, shape):
    """the discriminator's stride-2 3x3 convs (discriminator.py:11-26): LDS-patch forward and weight gradient
    against the generic kernels (same bf16 operands) and the oracle; odd sizes exercise the ragged tiles."""
    ops, K = _ops()
    from gan_danet_amd import _lib as L
    B, Cin, H, W, Cout = shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = bf16_round(seeded((B, Cin, H, W), 81)).to(DEV)
    w = bf16_round(seeded((Cout, Cin, 3, 3), 82, 1.0 / math.sqrt(Cin * 9))).to(DEV)
    bias = seeded((Cout,), 83, 0.1).to(DEV)
    dy = bf16_round(seeded((B, Cout, Ho, Wo), 84)).to(DEV)
    outs = {}
    for fast in (True, False):
        K.USE_CONV3X3_FAST = fast
        try:
            y = K.conv2d_fwd(x, w, bias, 2, 1, L.PREC_BF16, act=ops.ACT_LEAKY)
            dw = K.conv2d_wgrad(dy, x, 3, 2, 1, L.PREC_BF16)
        finally:
            K.USE_CONV3X3_FAST = True
        outs[fast] = (y, dw)
    assert_close(outs[True][0], outs[False][0].cpu(), 1e-5, "stride-2 patch vs generic fwd")
    assert_close(outs[True][1], outs[False][1].cpu(), 1e-4, "stride-2 patch vs generic wgrad")
    yr = F.leaky_relu(F.conv2d(x.cpu(), w.cpu(), bias.cpu(), stride=2, padding=1), 0.2)
    assert_close(outs[True][0], yr, BF16_TOL, "stride-2 fwd vs oracle")
    dwr = torch.nn.grad.conv2d_weight(x.cpu(), (Cout, Cin, 3, 3), dy.cpu(), stride=2, padding=1)
    assert_close(outs[True][1], dwr, BF16_TOL, "stride-2 wgrad vs oracle", rell2)


@pytest.mark.parametrize("shape", [(2, 184, 16, 16, 184), (1, 88, 8, 12, 44), (2, 520, 8, 8, 64), (1, 32, 5, 7, 24)])
def test_conv1x1_transpose_read_kernel(gd, shape):
    """1x1 convs (projections, transitions, channel_adjust): transpose-read kernel with and without the fused
    BN-affine+ReLU prologue, forward / data gradient / weight gradient, against the oracle (bf16 operands).
    The last shape (35 pixels) is not 16-byte friendly and must fall back to the generic kernel."""
    ops, K = _ops()
    from gan_danet_amd import _lib as L
    B, Cin, H, W, Cout = shape
    x = bf16_round(seeded((B, Cin, H, W), 71)).to(DEV)
    w = bf16_round(seeded((Cout, Cin, 1, 1), 72, 1.0 / math.sqrt(Cin))).to(DEV)
    bias = seeded((Cout,), 73, 0.1).to(DEV)
    sc, sh = (seeded((Cin,), 74).abs() * 0.5 + 0.75).to(DEV), seeded((Cin,), 75, 0.2).to(DEV)
    dy = bf16_round(seeded((B, Cout, H, W), 76)).to(DEV)
    y = K.conv2d_fwd(x, w, bias, 1, 0, L.PREC_BF16)
    yp = K.conv2d_fwd(x, w, bias, 1, 0, L.PREC_BF16, in_scale=sc, in_shift=sh, in_relu=True, act=ops.ACT_RELU)
    dx = K.conv2d_dgrad(dy, w, (H, W), 1, 0, L.PREC_BF16)
    dw = K.conv2d_wgrad(dy, x, 1, 1, 0, L.PREC_BF16)
    assert_close(y, F.conv2d(x.cpu(), w.cpu(), bias.cpu()), BF16_TOL, "1x1 fwd")
    xin = F.relu(x.cpu() * sc.cpu()[None, :, None, None] + sh.cpu()[None, :, None, None])
    assert_close(yp, F.relu(F.conv2d(xin, w.cpu(), bias.cpu())), BF16_TOL, "1x1 fwd prologue")
    assert_close(dx, torch.nn.grad.conv2d_input((B, Cin, H, W), w.cpu(), dy.cpu()), BF16_TOL, "1x1 dgrad", rell2)
    assert_close(dw, torch.nn.grad.conv2d_weight(x.cpu(), (Cout, Cin, 1, 1), dy.cpu()), BF16_TOL, "1x1 wgrad", rell2)


# ---- pixel-major bf16 kernels of the VGG feature stack -------------------------------------------------------------
def _nhwc(t):      # (B, C, H, W) fp32 cpu -> (B, H, W, C) bf16 gpu
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def _nchw(t):      # (B, H, W, C) bf16 gpu -> (B, C, H, W) fp32 cpu
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("shape", [(2, 64, 16, 32, 64), (1, 128, 17, 35, 256), (2, 40, 9, 70, 24), (1, 256, 8, 8, 512)])
def test_conv3x3_nhwc_forward_and_data_gradient(gd, shape):
    """gd_conv3x3_nhwc: forward (+bias+ReLU) and the data-gradient operator (ReLU mask and residual fused) against
    ATen on the same bf16-rounded operands; results are bf16-rounded on store (tolerance = one bf16 ulp + the
    usual accumulation-order slack)"""
    _, K = _ops()
    B, Cin, H, W, Cout = shape
    x = bf16_round(seeded((B, Cin, H, W), 131))
    w = bf16_round(seeded((Cout, Cin, 3, 3), 132, 1.0 / math.sqrt(Cin * 9)))
    bias = seeded((Cout,), 133, 0.1)
    y = K.conv3x3_nhwc(_nhwc(x), K.conv3x3_nhwc_pack(w.to(DEV), False), bias.to(DEV), Cout, relu=True)
    yr = F.relu(F.conv2d(x, w, bias, padding=1))
    assert_close(_nchw(y), yr, 1e-2, "nhwc fwd")
    dy = bf16_round(seeded((B, Cout, H, W), 134))
    act = bf16_round(seeded((B, Cin, H, W), 135))           # stands for the ReLU output feeding this conv
    res = bf16_round(seeded((B, Cin, H, W), 136, 0.1))
    dx = K.conv3x3_nhwc(_nhwc(dy), K.conv3x3_nhwc_pack(w.to(DEV), True), None, Cin, mask=_nhwc(act), res=_nhwc(res))
    dxr = torch.nn.grad.conv2d_input((B, Cin, H, W), w, dy, padding=1) * (act > 0).float() + res
    assert_close(_nchw(dx), dxr, 1e-2, "nhwc dgrad with mask + res")


@pytest.mark.parametrize("shape", [(2, 64, 32, 64, 128), (1, 128, 17, 35, 256), (2, 40, 9, 130, 24), (1, 256, 8, 8, 512),
                                   (3, 8, 13, 11, 64)])
def test_conv3x3_nhwc_stride2_forward_data_and_weight_gradient(gd, shape):
    """Discriminator1 conv2..4 on pixel-major bf16 (gd_conv3x3_nhwc_s2, gd_conv3x3_nhwc_s2_dgrad by input parity,
    gd_nhwc_to_nchw16 + gd_conv3x3_wgrad on the pixel-major input) against ATen on the same bf16-rounded operands;
    odd sizes exercise every tile edge and the odd-row / odd-column tail of the parity split"""
    _, K = _ops()
    B, Cin, H, W, Cout = shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = bf16_round(seeded((B, Cin, H, W), 231))
    w = bf16_round(seeded((Cout, Cin, 3, 3), 232, 1.0 / math.sqrt(Cin * 9)))
    bias = seeded((Cout,), 233, 0.1)
    y = K.conv3x3_nhwc_s2(_nhwc(x), K.conv3x3_nhwc_pack(w.to(DEV), 0), bias.to(DEV), Cout, 2, 0.2)
    yr = F.leaky_relu(F.conv2d(x, w, bias, stride=2, padding=1), 0.2)
    assert tuple(y.shape) == (B, Ho, Wo, Cout)
    assert_close(_nchw(y), yr, 1e-2, "nhwc s2 fwd")
    dy = bf16_round(seeded((B, Cout, Ho, Wo), 234))
    act = bf16_round(seeded((B, Cin, H, W), 235))           # stands for the LeakyReLU output feeding this conv
    dx = K.conv3x3_nhwc_s2_dgrad(_nhwc(dy), K.conv3x3_nhwc_pack(w.to(DEV), 2), _nhwc(act), 0.2)
    dxr = torch.nn.grad.conv2d_input((B, Cin, H, W), w, dy, stride=2, padding=1) * torch.where(act > 0, 1.0, 0.2)
    assert_close(_nchw(dx), dxr, 1e-2, "nhwc s2 dgrad with LeakyReLU mask")
    gt, cs = K.nhwc_to_nchw16(_nhwc(dy), True)
    assert torch.equal(gt.float().cpu(), dy)
    assert_close(cs, dy.sum((0, 2, 3)), 1e-4, "channel sums")
    dw, db = K.conv3x3_wgrad_nhwc(_nhwc(dy), _nhwc(x), 2, True)
    dwr = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, stride=2, padding=1)
    assert_close(dw, dwr, 1e-3, "nhwc wgrad")
    assert_close(db, dy.sum((0, 2, 3)), 1e-4, "bias gradient")


@pytest.mark.parametrize("ci,hw", [(1, (24, 40)), (3, (37, 51)), (1, (64, 66))])
def test_disc_stem_flatten_kernels(gd, ci, hw):
    """Discriminator1 conv1 from the fp32 image (gd_disc_stem_fwd / _wgrad / _dgrad, fp32 FMAs) and x.flatten(1)"""
    _, K = _ops()
    B, (H, W), Co = 2, hw, 64
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    img = seeded((B, ci, H, W), 241)
    w = seeded((Co, ci, 3, 3), 242, 0.3)
    bias = seeded((Co,), 243, 0.1)
    a = K.disc_stem_fwd(img.to(DEV), w.to(DEV), bias.to(DEV), 0.2)
    ar = F.leaky_relu(F.conv2d(img, w, bias, stride=2, padding=1), 0.2)
    assert_close(_nchw(a), ar, 1e-2, "stem fwd")
    g = bf16_round(seeded((B, Co, Ho, Wo), 244))
    dw, db = K.disc_stem_wgrad(_nhwc(g), img.to(DEV))
    assert_close(dw, torch.nn.grad.conv2d_weight(img, (Co, ci, 3, 3), g, stride=2, padding=1), 1e-4, "stem wgrad")
    assert_close(db, g.sum((0, 2, 3)), 1e-4, "stem bias gradient")
    dimg = K.disc_stem_dgrad(_nhwc(g), w.to(DEV), H, W)
    assert_close(dimg, torch.nn.grad.conv2d_input((B, ci, H, W), w, g, stride=2, padding=1), 1e-4, "stem dgrad")
    f = K.nhwc_flatten_fwd(a)
    assert torch.equal(f.cpu(), _nchw(a).flatten(1))
    df = seeded(tuple(f.shape), 245)
    gb = K.nhwc_flatten_bwd(df.to(DEV), a, 0.2)
    gr = bf16_round(df.view(B, Co, Ho, Wo) * torch.where(_nchw(a) > 0, 1.0, 0.2))
    assert torch.equal(_nchw(gb), gr)


def _nhwc_split(t):      # (B, C, H, W) fp32 cpu -> (B, H, W, 3 C) bf16 gpu, [hi | lo | hi]
    hi = bf16_round(t)
    lo = bf16_round(t - hi)
    return torch.cat([hi, lo, hi], 1).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def _nchw_split(t):      # (B, H, W, 3 C) bf16 gpu -> (B, C, H, W) fp32 cpu = hi + lo; the second hi copy must equal the first
    v = t.float().cpu().permute(0, 3, 1, 2)
    C = v.shape[1] // 3
    assert torch.equal(v[:, :C], v[:, 2 * C:])
    return (v[:, :C] + v[:, C:2 * C]).contiguous()


@pytest.mark.parametrize("shape", [(2, 64, 16, 32, 128), (1, 128, 17, 35, 256), (2, 32, 9, 70, 24)])
def test_disc_trunk_kernels_on_split_activations_vs_fp64(gd, shape):
    """operand mode "x3" of the Discriminator1 trunk: the stride-2 forward, its parity-split data gradient, the transposer
    and the weight gradient's three accumulating launches, the stem kernels and flatten on [hi | lo | hi] pixel-major
    tensors (split = 1) against torch in fp64 on the UNROUNDED operands: ~2^-16, asserted at 1e-4 where plain bf16
    storage gives 1e-2."""
    _, K = _ops()
    B, Cin, H, W, Cout = shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = seeded((B, Cin, H, W), 251)
    w = seeded((Cout, Cin, 3, 3), 252, 1.0 / math.sqrt(9 * Cin))
    bias = seeded((Cout,), 253, 0.1)
    xs = _nhwc_split(x)
    y = K.conv3x3_nhwc_s2(xs, K.conv3x3_nhwc_pack(K.split3_weights(w.to(DEV), 1), 0), bias.to(DEV), Cout, 2, 0.2, split=True)
    yr = F.leaky_relu(F.conv2d(x.double(), w.double(), bias.double(), stride=2, padding=1), 0.2).float()
    assert y.shape == (B, Ho, Wo, 3 * Cout)
    assert_close(_nchw_split(y), yr, 1e-4, "split s2 forward")
    dy = seeded((B, Cout, Ho, Wo), 254)
    act = seeded((B, Cin, H, W), 255)
    dx = K.conv3x3_nhwc_s2_dgrad(_nhwc_split(dy), K.conv3x3_nhwc_pack(K.split3_weights(w.to(DEV), 0), 2), _nhwc_split(act), 0.2,
                                 split=True)
    dxr = (torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double(), stride=2, padding=1)
           * torch.where(act > 0, 1.0, 0.2).double()).float()
    assert_close(_nchw_split(dx), dxr, 1e-4, "split s2 dgrad with LeakyReLU mask")
    gt, cs = K.nhwc_to_nchw16(_nhwc_split(dy), True, split=True)
    assert gt.shape == (2, B, Cout, Ho, Wo)
    assert torch.equal(gt[0].float().cpu(), bf16_round(dy)) and torch.equal(gt[1].float().cpu(), bf16_round(dy - bf16_round(dy)))
    assert_close(cs, dy.sum((0, 2, 3)), 1e-4, "channel sums of hi + lo")
    dw, db = K.conv3x3_wgrad_nhwc(_nhwc_split(dy), xs, 2, True, split=True)
    dwr = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), stride=2, padding=1).float()
    assert_close(dw, dwr, 1e-4, "split nhwc wgrad")
    assert_close(db, dy.sum((0, 2, 3)), 1e-4, "bias gradient")
    # stem (fp32 FMAs from the image) writing / reading split tensors, and flatten
    img = seeded((B, 1, 2 * H, 2 * W), 256)
    ws_ = seeded((64, 1, 3, 3), 257, 0.3)
    bs_ = seeded((64,), 258, 0.1)
    a = K.disc_stem_fwd(img.to(DEV), ws_.to(DEV), bs_.to(DEV), 0.2, split=True)
    ar = F.leaky_relu(F.conv2d(img, ws_, bs_, stride=2, padding=1), 0.2)
    assert_close(_nchw_split(a), ar, 1e-5, "split stem fwd")
    g = seeded((B, 64, H, W), 259)
    dws, dbs = K.disc_stem_wgrad(_nhwc_split(g), img.to(DEV), split=True)
    assert_close(dws, torch.nn.grad.conv2d_weight(img, (64, 1, 3, 3), g, stride=2, padding=1), 1e-4, "split stem wgrad")
    assert_close(dbs, g.sum((0, 2, 3)), 1e-4, "split stem bias gradient")
    dimg = K.disc_stem_dgrad(_nhwc_split(g), ws_.to(DEV), 2 * H, 2 * W, split=True)
    assert_close(dimg, torch.nn.grad.conv2d_input((B, 1, 2 * H, 2 * W), ws_, g, stride=2, padding=1), 1e-4, "split stem dgrad")
    f = K.nhwc_flatten_fwd(a, split=True)
    assert torch.equal(f.cpu(), _nchw_split(a).flatten(1))
    df = seeded(tuple(f.shape), 260)
    gb = K.nhwc_flatten_bwd(df.to(DEV), a, 0.2, split=True)
    gr = df.view(B, 64, H, W) * torch.where(_nchw_split(a) > 0, 1.0, 0.2)
    assert_close(_nchw_split(gb), gr, 1e-5, "split flatten bwd")


@pytest.mark.parametrize("ci", [1, 3])
def test_nhwc_vgg_kernels_on_split_activations(gd, ci):
    """operand mode "x3" of the pixel-major VGG path: stem, 3x3 conv (+bias+ReLU; data-gradient form with mask and res),
    2x2 max-pool forward / backward and the L1 distance / gradient on [hi | lo | hi] tensors (split = 1), against fp64 on
    the unrounded values"""
    _, K = _ops()
    B, H, W, Co = 2, 24, 40, 64
    img = seeded((B, ci, H, W), 271)
    w0 = seeded((Co, ci, 3, 3), 272, 0.3)
    b0 = seeded((Co,), 273, 0.1)
    a = K.nhwc_stem_fwd(img.to(DEV), w0.to(DEV), b0.to(DEV), True, split=True)
    ar = F.relu(F.conv2d(img, w0, b0, padding=1))
    assert a.shape == (B, H, W, 3 * Co)
    assert_close(_nchw_split(a), ar, 1e-5, "split stem fwd")
    g = seeded((B, Co, H, W), 274)
    dimg = K.nhwc_stem_bwd(_nhwc_split(g), w0.to(DEV), split=True)
    assert_close(dimg, torch.nn.grad.conv2d_input((B, ci, H, W), w0, g, padding=1), 1e-4, "split stem bwd")
    # conv + bias + ReLU, then its data-gradient form: [mask > 0] * convT(dy) + res
    M = 48
    w1 = seeded((M, Co, 3, 3), 275, 1.0 / math.sqrt(9 * Co))
    b1 = seeded((M,), 276, 0.1)
    x = seeded((B, Co, H, W), 277)
    y = K.conv3x3_nhwc(_nhwc_split(x), K.conv3x3_nhwc_pack(K.split3_weights(w1.to(DEV), 1), False), b1.to(DEV), M, relu=True,
                       split=True)
    yr = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1)).float()
    assert_close(_nchw_split(y), yr, 1e-4, "split nhwc conv")
    dy, act, res = seeded((B, M, H, W), 278), seeded((B, Co, H, W), 279), seeded((B, Co, H, W), 280)
    dx = K.conv3x3_nhwc(_nhwc_split(dy), K.conv3x3_nhwc_pack(K.split3_weights(w1.to(DEV), 0), True), None, Co,
                        mask=_nhwc_split(act), res=_nhwc_split(res), split=True)
    dxr = (torch.nn.grad.conv2d_input((B, Co, H, W), w1.double(), dy.double(), padding=1) * (act > 0).double() + res.double()).float()
    assert_close(_nchw_split(dx), dxr, 1e-4, "split nhwc data gradient with mask + res")
    # pooling: the maximum of split values splits back into the same (hi, lo); backward exact incl. the ReLU gate
    xv = x.clone().requires_grad_(True)
    p = K.nhwc_maxpool2_fwd(_nhwc_split(x), split=True)
    pr = F.max_pool2d(xv, 2)
    assert_close(_nchw_split(p), pr.detach(), 2.0 ** -16, "split pool fwd")
    gp = seeded(tuple(pr.shape), 281)
    pr.backward(gp)
    dpx = K.nhwc_maxpool2_bwd(_nhwc_split(x), _nhwc_split(gp), True, split=True)
    assert_close(_nchw_split(dpx), xv.grad * (x > 0).float(), 2.0 ** -16, "split pool bwd")
    fa, fb = seeded((B, Co, H, W), 282), seeded((B, Co, H, W), 283)
    out = torch.zeros(1, device=DEV)
    K.nhwc_l1(_nhwc_split(fa), _nhwc_split(fb), out, False, split=True)
    K.nhwc_l1(_nhwc_split(fa), _nhwc_split(fb), out, True, split=True)
    assert_close(out, 2 * (fa - fb).abs().mean().view(1), 1e-5, "split l1 (accumulated twice)")
    up = torch.tensor([0.7], device=DEV)
    gl = K.nhwc_l1_grad(_nhwc_split(fa), _nhwc_split(fb), up, True, split=True)
    glr = 0.7 / fa.numel() * torch.sign(fa - fb) * (fa > 0).float()
    assert_close(_nchw_split(gl), glr, 1e-5, "split l1 grad")


@pytest.mark.parametrize("ci", [1, 3])
def test_nhwc_stem_pool_l1(gd, ci):
    _, K = _ops()
    B, H, W, Co = 2, 24, 40, 64
    img = seeded((B, ci, H, W), 141)
    w = seeded((Co, ci, 3, 3), 142, 0.3)
    bias = seeded((Co,), 143, 0.1)
    a = K.nhwc_stem_fwd(img.to(DEV), w.to(DEV), bias.to(DEV), True)
    ar = F.relu(F.conv2d(img, w, bias, padding=1))
    assert_close(_nchw(a), ar, 1e-2, "stem fwd")
    g = bf16_round(seeded((B, Co, H, W), 144))
    dimg = K.nhwc_stem_bwd(_nhwc(g), w.to(DEV))
    assert_close(dimg, torch.nn.grad.conv2d_input((B, ci, H, W), w, g, padding=1), 1e-4, "stem bwd")
    # pooling on the (exactly representable) bf16 activations: forward exact, backward exact incl. the ReLU gate
    av = _nchw(a).requires_grad_(True)
    p = K.nhwc_maxpool2_fwd(a)
    pr = F.max_pool2d(av, 2)
    assert torch.equal(_nchw(p), pr.detach())
    gp = bf16_round(seeded(tuple(pr.shape), 145))
    pr.backward(gp)
    dpx = K.nhwc_maxpool2_bwd(a, _nhwc(gp), True)
    assert torch.equal(_nchw(dpx), av.grad * (av.detach() > 0).float())
    # L1 distance and its gated gradient
    b2 = K.nhwc_stem_fwd(seeded((B, ci, H, W), 146).to(DEV), w.to(DEV), bias.to(DEV), True)
    out = torch.zeros(1, device=DEV)
    K.nhwc_l1(a, b2, out, False)
    fa, fb = _nchw(a), _nchw(b2)
    assert_close(out, (fa - fb).abs().mean().view(1), 1e-5, "l1 value")
    up = torch.full((1,), 0.5, device=DEV)
    gl = K.nhwc_l1_grad(a, b2, up, True)
    ref = 0.5 / fa.numel() * torch.sign(fa - fb) * (fa > 0).float()
    assert_close(_nchw(gl), ref, 1e-2, "l1 grad")


def test_copy_slab_vector_and_scalar_paths(gd):
    """gd_copy_slab: channel slices of wider slabs in and out, plain and accumulating; 16-byte path (aligned, multiples of 4)
    and the scalar path (odd sizes / a slice starting at an odd channel of an odd-sized plane)"""
    _, K = _ops()
    for (B, Ctot, C, H, W, c0) in ((3, 12, 8, 8, 8, 4), (2, 7, 5, 5, 7, 1), (2, 9, 4, 6, 6, 3)):
        src = seeded((B, Ctot, H, W), 401).to(DEV)
        dst = seeded((B, Ctot + 3, H, W), 402).to(DEV)
        ref = dst.clone()
        K.copy_slab(src[:, c0:c0 + C], dst[:, 1:1 + C])
        ref[:, 1:1 + C] = src[:, c0:c0 + C]
        assert torch.equal(dst, ref)
        K.copy_slab(src[:, :C], dst[:, 2:2 + C], accumulate=True)
        ref[:, 2:2 + C] += src[:, :C]
        assert torch.equal(dst, ref)


def test_shift_sum9_is_the_one_hot_3x3_conv(gd):
    """ShiftSum9Fn (last step of the collapsed generator tail): bias + sum_tap shift(u_tap) equals conv3x3(u, E), E the
    one-hot (1, 9, 3, 3) kernel with E[0][tap][tap // 3][tap % 3] = 1; forward and both gradients, exact up to the order
    of nine fp32 additions"""
    ops, _ = _ops()
    u = seeded((2, 9, 13, 37), 161)
    bias = seeded((1,), 162)
    E = torch.zeros(1, 9, 3, 3)
    for t in range(9):
        E[0, t, t // 3, t % 3] = 1.0
    ur, br = u.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yr = F.conv2d(ur, E, br, padding=1)
    go = seeded(tuple(yr.shape), 163)
    yr.backward(go)
    ug, bg = u.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    y = ops.ShiftSum9Fn.apply(ug, bg)
    y.backward(go.to(DEV))
    assert_close(y, yr, 1e-6, "y")
    assert_close(ug.grad, ur.grad, 1e-7, "du")
    assert_close(bg.grad, br.grad, 1e-5, "dbias")


@pytest.mark.parametrize("shape,size_average", [((2, 3, 24, 20), False), ((3, 1, 40, 33), True)])
def test_ssim_module_gradients_vs_oracle(gd, shape, size_average):
    """SSIM drop-in (losses.py:90-147): both reductions, multi-channel windows, gradients w.r.t. BOTH images, through
    the exported module and a weighted upstream -- against the oracle's autograd (fp64)"""
    import gan_danet_amd as G
    from oracle import functional as OF
    g = torch.Generator().manual_seed(5)
    a = torch.rand(shape, generator=g) * 2 - 1
    b = (a + 0.3 * torch.randn(shape, generator=g)).clamp(-1.5, 1.5)
    w = torch.rand(shape[0], generator=g) + 0.5
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = OF.ssim(ar, br, 11, size_average)
    (ref * (w.double() if not size_average else 1.0)).sum().backward()
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    m = G.SSIM(11, size_average).to(DEV)
    out = m(ad, bd)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close(out, ref.float(), 1e-4, "ssim value")
    if size_average:
        out.backward()
    else:
        out.backward(w.to(DEV))
    assert_close(ad.grad, ar.grad.float(), 2e-4, "dssim/dimg1", rell2)
    assert_close(bd.grad, br.grad.float(), 2e-4, "dssim/dimg2", rell2)
    assert m.window.shape == (shape[1], 1, 11, 11)


def test_bce_target_tensor_and_general_leaky_slope(gd):
    """API width of the drop-in criteria / activations: BCEWithLogitsLoss with a target TENSOR and
    LeakyReLU with a slope other than the fused 0.2, against ATen on the CPU"""
    import gan_danet_amd as G
    from gan_danet_amd.layers import LeakyReLU
    z = seeded((6, 1), 71)
    t = torch.rand(6, 1, generator=torch.Generator().manual_seed(72))
    zr, tr = z.clone().requires_grad_(True), t.clone().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(zr, tr)
    ref.backward()
    zd, td = z.to(DEV).requires_grad_(True), t.to(DEV).requires_grad_(True)
    out = G.BCEWithLogitsLoss()(zd, td)
    out.backward()
    assert_close(out, ref.detach(), 1e-6, "bce(target tensor)")
    assert_close(zd.grad, zr.grad, 1e-5, "dbce/dz")
    assert_close(td.grad, tr.grad, 1e-5, "dbce/dt")
    assert_close(G.BCEWithLogitsLoss()(zd.detach(), torch.ones(())), F.binary_cross_entropy_with_logits(z, torch.ones_like(z)), 1e-6, "0-dim target")
    for slope in (0.01, 0.2, 0.0):
        x = seeded((3, 5, 7, 4), 73)
        xr = x.clone().requires_grad_(True)
        yr = F.leaky_relu(xr, slope)
        go = seeded(tuple(x.shape), 74)
        yr.backward(go)
        xd = x.to(DEV).requires_grad_(True)
        yd = LeakyReLU(slope)(xd)
        yd.backward(go.to(DEV))
        assert torch.equal(yd.cpu(), yr.detach()) and torch.equal(xd.grad.cpu(), xr.grad), f"LeakyReLU({slope})"


def test_deterministic_mode_makes_split_reductions_reproducible(gd):
    """gd.set_deterministic(True): the 3x3 weight gradient and the split-K NT GEMM (fp32 atomics across splits by
    default) run unsplit -- two launches agree bit for bit and match the default mode to fp32 round-off"""
    ops, K = _ops()
    from gan_danet_amd import _lib as L
    x = seeded((4, 64, 96, 96), 81).to(DEV)
    dy = seeded((4, 24, 96, 96), 82).to(DEV)
    a = seeded((8, 200000), 83).to(DEV)
    bm = seeded((24, 200000), 84).to(DEV)

    def run():
        dw = K.conv2d_wgrad(dy, x, 3, 1, 1, L.PREC_BF16)
        c = torch.empty(1, 8, 24, device=DEV)
        K.gemm_nt(B=1, M=8, N=24, kseg=1, klen=200000, a=a, a_bs=a.numel(), a_ss=0, lda=200000, bm=bm, b_bs=bm.numel(), b_ss=0,
                  ldb=200000, c=c, c_bs=8 * 24, ldc=24, precision=L.PREC_FP32)
        return dw.clone(), c.clone()

    base = run()
    gd.set_deterministic(True)
    try:
        assert L.load().gd_get_deterministic() == 1
        d1, d2 = run(), run()
    finally:
        gd.set_deterministic(False)
    assert torch.equal(d1[0], d2[0]) and torch.equal(d1[1], d2[1])
    assert_close(d1[0], base[0].cpu(), 1e-5, "deterministic vs default wgrad", rell2)
    assert_close(d1[1], base[1].cpu(), 1e-5, "deterministic vs default split-K", rell2)


def test_rccl_comm_c_abi_single_rank(gd):
    """gd_comm_* / gd_allreduce exist and drive RCCL (dlopen'ed) -- exercised with a world of one on the
    box's single GPU: sum all-reduce, reduce-scatter and all-gather are identities; a second init is refused"""
    import ctypes
    from gan_danet_amd import _lib as L
    lib = L.load()
    ident = ctypes.create_string_buffer(128)
    L.check(lib.gd_comm_unique_id(ident), "gd_comm_unique_id")
    L.check(lib.gd_comm_init(0, 1, ident), "gd_comm_init")
    try:
        assert lib.gd_comm_world() == 1
        assert lib.gd_comm_init(0, 1, ident) != 0 and "already" in L.last_error()
        x = torch.arange(1000, device=DEV, dtype=torch.float32)
        s = torch.cuda.current_stream().cuda_stream
        L.check(lib.gd_allreduce(x.data_ptr(), x.numel(), 0, s), "gd_allreduce")
        y = torch.empty_like(x)
        L.check(lib.gd_reduce_scatter(x.data_ptr(), y.data_ptr(), x.numel(), 0, s), "gd_reduce_scatter")
        z = torch.empty_like(x)
        L.check(lib.gd_allgather(y.data_ptr(), z.data_ptr(), y.numel(), 0, s), "gd_allgather")
        torch.cuda.synchronize()
        ref = torch.arange(1000, dtype=torch.float32)
        assert torch.equal(x.cpu(), ref) and torch.equal(y.cpu(), ref) and torch.equal(z.cpu(), ref)
        h = torch.ones(64, device=DEV, dtype=torch.bfloat16)
        L.check(lib.gd_allreduce(h.data_ptr(), h.numel(), 1, s), "gd_allreduce bf16")
        assert lib.gd_allreduce(h.data_ptr(), h.numel(), 7, s) != 0
    finally:
        L.check(lib.gd_comm_destroy(), "gd_comm_destroy")
    assert lib.gd_comm_world() == 0


# ---- split-bf16 ("x3") operands of set_precision("mixed") -------------------------------------------------------------
def test_split_pack_layouts_and_weight_split(gd):
    """gd_pack_16_split / gd_split3_weights against their definition: hi = bf16(v), lo = bf16(v - hi); pixel-major
    [hi | lo | hi] (3 C channels per pixel) and channel-major hi / lo images; optional BatchNorm-affine + ReLU first;
    the input may be a channel slice of a wider slab.  hi + lo reproduces v to 2^-16."""
    _, K = _ops()
    B, Ctot, C, H, W = 2, 96, 72, 8, 12
    slab = seeded((B, Ctot, H, W), 51).to(DEV)
    x = slab[:, :C]
    sc, sh = seeded((C,), 52).to(DEV), seeded((C,), 53, 0.3).to(DEV)
    for affine in (False, True):
        ref = x.cpu()
        if affine:
            ref = torch.relu(ref * sc.cpu().view(1, C, 1, 1) + sh.cpu().view(1, C, 1, 1))
        hi = bf16_round(ref)
        lo = bf16_round(ref - hi)
        plain, tr = K.pack_split(x, scale=sc if affine else None, shift=sh if affine else None, relu=affine, want_plain=True)
        assert plain.shape == (2, B, C, H * W) and tr.shape == (B, H * W, 3 * C)
        tol = 0.0 if not affine else 1e-6          # fmaf in the kernel vs mul + add here: a last-bit matter before rounding
        for got, want, nm in ((plain[0].float().cpu(), hi.view(B, C, -1), "plain hi"), (plain[1].float().cpu(), lo.view(B, C, -1), "plain lo"),
                              (tr[..., :C].float().cpu(), hi.view(B, C, -1).transpose(1, 2), "tr hi"),
                              (tr[..., C:2 * C].float().cpu(), lo.view(B, C, -1).transpose(1, 2), "tr lo"),
                              (tr[..., 2 * C:].float().cpu(), hi.view(B, C, -1).transpose(1, 2), "tr hi (second copy)")):
            if affine:     # a value that lands on a rounding boundary may fall either way: compare hi + lo instead of bits
                continue
            assert torch.equal(got, want), nm
        rec = (plain[0].float() + plain[1].float()).cpu().view(B, C, H, W)
        assert ((rec - ref).abs() <= 2.0 ** -16 * ref.abs() + 1e-30 + tol).all()
        rec_t = (tr[..., :C].float() + tr[..., C:2 * C].float()).cpu().transpose(1, 2).reshape(B, C, H, W)
        assert torch.equal(rec_t, rec)
    # the ReLU backward fused into the pack (gd_pack_16_split_masked): bitwise the pack of act_bwd's result
    ymask = torch.relu(seeded((B, C, H, W), 55)).to(DEV)
    xm = x.contiguous()
    p_ref, t_ref = K.pack_split(K.act_bwd(ymask, xm, 1), want_plain=True)
    p_got, t_got = K.pack_split(xm, want_plain=True, mask=ymask)
    assert torch.equal(p_ref.view(torch.int16), p_got.view(torch.int16)) and torch.equal(t_ref.view(torch.int16), t_got.view(torch.int16))
    assert (p_got[0].float().view(B, C, H, W)[ymask <= 0] == 0).all() and (ymask <= 0).float().mean() > 0.3
    w = seeded((24, 40, 3, 3), 54, 0.2).to(DEV)
    whi = bf16_round(w.cpu())
    w1 = K.split3_weights(w, 1).cpu()
    assert torch.equal(w1[:, :40], whi) and torch.equal(w1[:, 40:80], whi) and torch.equal(w1[:, 80:], w.cpu() - whi)
    w0 = K.split3_weights(w, 0).cpu()
    assert torch.equal(w0[:24], whi) and torch.equal(w0[24:48], whi) and torch.equal(w0[48:], w.cpu() - whi)


@pytest.mark.parametrize("case", [(2, 136, 16, 24, 24, True, False), (1, 368, 16, 16, 184, False, True),
                                  (2, 64, 24, 8, 64, True, True), (1, 32, 8, 8, 8, False, False)])
def test_conv3x3_split_bf16_vs_fp64(gd, case):
    """the split-bf16 route of a 3x3 / stride 1 / pad 1 conv (forward, data gradient, weight gradient in three
    accumulating launches, bias gradient) against torch in fp64: three bf16 MFMAs per product reach ~2^-16, two orders
    of magnitude under plain bf16 operands (2e-3) -- asserted at 1e-4 (max-norm, relative to the output scale).
    Cin = 136 is not a multiple of the weight gradient's 32-channel chunk: its last chunk reads into the lo columns of
    the [hi | lo | hi] row, and those output rows must be dropped."""
    ops, _ = _ops()
    B, Cin, H, W, Cout, bias, relu = case
    x = seeded((B, Cin, H, W), 61)
    w = seeded((Cout, Cin, 3, 3), 62, 1.0 / math.sqrt(Cin * 9))
    b = seeded((Cout,), 63, 0.1) if bias else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=1)
    if relu:
        yr = torch.relu(yr)
    go = seeded(tuple(yr.shape), 64)
    yr.backward(go.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if bias else None
    with gd.precision("fp32"), gd.layer_override(other="x3"):
        y = ops.conv2d(xd, wd, bd, 1, 1, ops.ACT_RELU if relu else ops.ACT_NONE)
        y.backward(go.to(DEV))
    assert_close(y, yr.float(), 1e-4, "y")
    assert_close(xd.grad, xr.grad.float(), 1e-4, "dx")
    assert_close(wd.grad, wr.grad.float(), 1e-4, "dw")
    if bias:
        assert_close(bd.grad, br.grad.float(), 1e-5, "db")
    # and it is NOT the exact route by accident: plain bf16 operands on the same inputs are ~100x further away
    xb = x.to(DEV).requires_grad_(True)
    with gd.precision("bf16"):
        yb = ops.conv2d(xb, w.to(DEV), None if b is None else b.to(DEV), 1, 1, ops.ACT_RELU if relu else ops.ACT_NONE)
    assert relmax(yb, yr.float()) > 20 * relmax(y, yr.float())


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_generic_split_bf16_vs_fp64(gd, case):
    """GD_PREC_X3 in the generic kernels (gd_conv2d: implicit-GEMM gather and the 1x1 transpose-read kernel; gd_gemm_nt:
    the im2col / plain split-K weight gradients): fp32 operands split into hi + lo bf16 while they are staged, three MFMAs
    per product.  Every generic conv shape of the net (1x1, stride 2, 4x4, Cin = 1 / 3, ragged) against torch in fp64 at
    1e-4; LeakyReLU keeps the 3x3 / stride-1 cases off the pixel-major split route (ops._x3_eligible)."""
    ops, _ = _ops()
    B, Cin, H, W, Cout, k, s_, p_, bias = case
    x = seeded((B, Cin, H, W), 81)
    w = seeded((Cout, Cin, k, k), 82, 1.0 / math.sqrt(Cin * k * k))
    b = seeded((Cout,), 83, 0.1) if bias else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.leaky_relu(F.conv2d(xr, wr, br, stride=s_, padding=p_), 0.2)
    go = seeded(tuple(yr.shape), 84)
    yr.backward(go.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if bias else None
    with gd.precision("fp32"), gd.layer_override(other="x3"):
        assert ops._prec("other") == ops.L.PREC_X3
        y = ops.conv2d(xd, wd, bd, s_, p_, ops.ACT_LEAKY)
        y.backward(go.to(DEV))
    assert_close(y, yr.float(), 1e-4, "y")
    assert_close(xd.grad, xr.grad.float(), 1e-4, "dx")
    assert_close(wd.grad, wr.grad.float(), 1e-4, "dw")
    if bias:
        assert_close(bd.grad, br.grad.float(), 1e-5, "db")
    xb = x.to(DEV)
    with gd.precision("bf16"):
        yb = ops.conv2d(xb, w.to(DEV), None if b is None else b.to(DEV), s_, p_, ops.ACT_LEAKY)
    if Cin * k * k >= 64:        # plain bf16 operands are an order of magnitude further away once the reduction is long
        assert relmax(yb, yr.float()) > 10 * relmax(y, yr.float())


@pytest.mark.parametrize("shape", [(1, 48, 200, 1, 4096, True), (3, 130, 70, 5, 333, False), (2, 24, 129, 2, 64, True)])
def test_gemm_nt_split_bf16_vs_fp64(gd, shape):
    """gd_gemm_nt with GD_PREC_X3 (float4-staged and scalar-staged variants, split-K atomics) against fp64"""
    _, K = _ops()
    Bn, M, N, kseg, klen, vec = shape
    a = seeded((Bn, kseg, M, klen), 91)
    bm = seeded((Bn, kseg, N, klen), 92)
    ref = torch.einsum("bsmk,bsnk->bmn", a.double(), bm.double()).float()
    c = torch.empty(Bn, M, N, device=DEV)
    ad, bd = a.to(DEV), bm.to(DEV)
    K.gemm_nt(B=Bn, M=M, N=N, kseg=kseg, klen=klen, a=ad, a_bs=kseg * M * klen, a_ss=M * klen, lda=klen, bm=bd,
              b_bs=kseg * N * klen, b_ss=N * klen, ldb=klen, c=c, c_bs=M * N, ldc=N, precision=K.L.PREC_X3)
    assert_close(c, ref, 1e-4, "split-bf16 NT GEMM")
    cb = torch.empty_like(c)
    K.gemm_nt(B=Bn, M=M, N=N, kseg=kseg, klen=klen, a=ad, a_bs=kseg * M * klen, a_ss=M * klen, lda=klen, bm=bd,
              b_bs=kseg * N * klen, b_ss=N * klen, ldb=klen, c=cb, c_bs=M * N, ldc=N, precision=K.L.PREC_BF16)
    assert relmax(cb, ref) > 10 * relmax(c, ref)


def test_syncbn_kernels_merge_of_two_shards_equals_full_batch(gd):
    """SyncBN's kernels without torch.distributed: the (count, mean, M2) records of two unequal shards, stacked as the
    all-gather would deliver them, merge (gd_bn_stats_merge) to the statistics gd_bn_stats takes on the whole batch
    -- mean, invstd, running statistics -- and gd_bn_act_bwd_dx on a shard with the summed dy sums and the global count
    equals the shard's rows of the full-batch backward."""
    _, K = _ops()
    B, C, H, W = 5, 24, 12, 16
    x = (seeded((B, C, H, W), 71) * 2.0 + 0.7).to(DEV)
    dy = seeded((B, C, H, W), 72).to(DEV)
    gamma, beta = (seeded((C,), 73) * 0.3 + 1.0).to(DEV), seeded((C,), 74, 0.2).to(DEV)
    eps, mom = 1e-5, 0.1
    rm_a, rv_a = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    rm_b, rv_b = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, invstd = K.bn_stats(x, eps, mom, rm_a, rv_a)
    shards = (x[:2].contiguous(), x[2:].contiguous())
    recs = torch.stack([K.bn_stats_local(s_) for s_ in shards], 0)            # (2, C, 3): unequal counts 2 and 3 images
    assert recs.shape == (2, C, 3) and recs[0, 0, 0].item() == 2 * H * W and recs[1, 0, 0].item() == 3 * H * W
    m2, is2 = K.bn_stats_merge(recs.contiguous(), eps, mom, rm_b, rv_b)
    assert_close(m2, mean, 1e-6, "merged mean")
    assert_close(is2, invstd, 1e-6, "merged invstd")
    assert_close(rm_b, rm_a, 1e-6, "running mean")
    assert_close(rv_b, rv_a, 1e-6, "running var (unbiased, global count)")
    scale, shift = K.bn_fold(gamma, beta, mean, invstd)
    dg, db, dx = K.bn_act_bwd(dy, x, scale, shift, mean, invstd, 1, True)
    sums = torch.zeros(2, C, device=DEV)
    for lo, hi in ((0, 2), (2, 5)):
        part = torch.empty(2, C, device=DEV)
        K.bn_act_bwd(dy[lo:hi].contiguous(), x[lo:hi].contiguous(), scale, shift, mean, invstd, 1, True, want_dx=False, sums_out=part)
        sums += part                                                             # the all-reduce (test plumbing)
    assert_close(sums[0], dg, 1e-5, "summed dgamma")
    assert_close(sums[1], db, 1e-5, "summed dbeta")
    for lo, hi in ((0, 2), (2, 5)):
        dxs = K.bn_act_bwd_dx(dy[lo:hi].contiguous(), x[lo:hi].contiguous(), scale, shift, mean, invstd, sums[0], sums[1],
                              1.0 / (B * H * W), 1)
        assert_close(dxs, dx[lo:hi].cpu(), 1e-5, f"dx of shard {lo}:{hi}")


def test_pam_f16_scale_is_a_power_of_two_and_fills_the_range(gd):
    """gd_pam_f16_scale: scales[0] = gamma * 2^k, scales[1] = 2^-k with max |gamma * 2^k * dOut| in [0.5, 1); delta is
    multiplied by 2^k in place; an all-zero gradient leaves k = 0"""
    _, K = _ops()
    for amp in (3e-9, 1.0, 7e4):
        do = (seeded((2, 16, 8, 8), 81) * amp).to(DEV)
        gamma = torch.full((1,), 0.3, device=DEV)
        delta = seeded((2, 64), 82).to(DEV)
        d0 = delta.clone()
        sc = K.pam_f16_scale(do, gamma, delta).cpu()
        up = sc[0].item() / 0.3
        assert abs(up * sc[1].item() - 1.0) < 1e-6
        assert abs(round(torch.log2(torch.tensor(up)).item()) - torch.log2(torch.tensor(up)).item()) < 1e-5, "not a power of two"
        top = (do.abs().max().item() * 0.3) * up
        assert 0.5 <= top < 1.0, top
        assert_close(delta, d0.cpu() * up, 1e-6, "delta scaled in place")
    do = torch.zeros(1, 8, 4, 4, device=DEV)
    sc = K.pam_f16_scale(do, torch.full((1,), 0.3, device=DEV), torch.zeros(1, 16, device=DEV)).cpu()
    assert abs(sc[0].item() - 0.3) < 1e-7 and sc[1].item() == 1.0
