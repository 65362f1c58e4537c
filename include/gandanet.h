/* gandanet.h -- C ABI of libgandanet_hip.so: the MI355X (gfx950) kernels behind the
 * GAN-DANet G+D training hot path.
 *
 * The reference (Aster32/GAN-DANet) has no native code and no FFI: its boundary for this
 * path is the Python nn.Module surface (models/__init__.py:12-23), and the arithmetic is
 * whatever ATen kernel each nn.* / F.* line dispatches to.  Each entry point below replaces
 * one such ATen call site; the reference line(s) it stands in for are cited per function.
 * INTEGRATION.md shows the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into memory owned by the caller (outputs and
 *     workspaces included); the library allocates nothing and keeps no pointer after return;
 *   - tensors are dense row-major fp32 NCHW unless a parameter says otherwise; "bs" = batch
 *     stride in ELEMENTS (lets a call address a channel slice of a wider slab);
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work (no host sync);
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch failure); the message is
 *     retrievable with gd_last_error(); nothing throws across the boundary;
 *   - re-entrant; no global state except the last-error string (thread-local).
 */
#ifndef GANDANET_H
#define GANDANET_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GD_VERSION 100 /* 0.1.0 */
/* deterministic mode (process-global): on = every reduction in a fixed order (split-K GEMM and the 3x3 weight
 * gradient run unsplit instead of combining partial sums with fp32 atomics) -- bitwise reproducible, slower. */
void gd_set_deterministic(int on);
int gd_get_deterministic(void);
int gd_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
int gd_last_error(char* buf, int n);
/* sizeof() of the two descriptor structs below, so a foreign binding can verify its mirror */
int gd_sizeof_conv_desc(void);
int gd_sizeof_gemm_nt_desc(void);

enum { GD_PREC_FP32 = 0, /* exact f32 MFMA (v_mfma_f32_32x32x2_f32) */
       GD_PREC_BF16 = 1, /* bf16 operands, f32 accumulate (v_mfma_f32_32x32x16_bf16) */
       GD_PREC_X3 = 2    /* split-bf16: fp32 operands staged as hi + lo bf16, hi*hi + lo*hi + hi*lo (~2^-16 relative):
                            gd_conv2d and gd_gemm_nt only */ };
enum { GD_ACT_NONE = 0, GD_ACT_RELU = 1, GD_ACT_LEAKY02 = 2, GD_ACT_SIGMOID = 3 };   /* sigmoid: gd_act_fwd/bwd only */

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, "NN" form:   out[b][m][p] = sum_{tap,c} A[b][m][c][tap] * X~[b][c][p (+) tap]
 * One kernel serves nn.Conv2d forward (generator.py:20,34,63,108-110,148,188,214,218,222,228;
 * discriminator.py:62-65; VGG convs losses.py:41), its data gradient (transposed gather),
 * CAM's attention*X product with per-batch "weights" (generator.py:137), and the
 * data/weight gradients of nn.Linear (discriminator.py:66-67) viewed as 1x1 convolutions.
 * ---------------------------------------------------------------------------------------- */
typedef struct gd_conv_desc {
    int B;          /* batch count (grid z) */
    int M;          /* rows of A that exist (Cout forward, Cin for the data gradient) */
    int Mstore;     /* rows written (>= M; rows in [M, Mstore) are written as zeros + epilogue) */
    int Ck;         /* reduction channels (Cin forward, Cout for the data gradient) */
    int ks;         /* square kernel size (1, 3, ...) */
    int stride, pad;
    int transposed; /* 0: iy = oy*stride - pad + kh;  1: iy = (oy + pad - kh)/stride where divisible */
    int Hi, Wi;     /* spatial size of the tensor X being gathered */
    int Ho, Wo;     /* spatial size of the output */
    /* A element (b, m, c, tap) at a[b*a_bs + m*a_sm + c*a_sc + tap*a_st] */
    const float* a; long a_bs, a_sm, a_sc, a_st;
    /* X element (b, c, y, x) at x[b*x_bs + c*Hi*Wi + y*Wi + x] */
    const float* x; long x_bs;
    /* optional fused input transform X~ = relu?(X*in_scale[c] + in_shift[c]) (BatchNorm+ReLU of
       generator.py:36 folded into the consumer); padding stays zero.  NULL = identity */
    const float* in_scale; const float* in_shift; int in_relu;
    /* output */
    void* y; long y_bs;
    int out_layout;  /* 0: y[b*y_bs + m*Ho*Wo + p];  1: y[b*y_bs + p*ldo + m] (pixel-major) */
    int out_bf16;    /* 0 fp32, 1 bf16 */
    int ldo;
    /* epilogue: v = acc * (alpha ? *alpha : 1) + bias[m] + res[b*res_bs + m*Ho*Wo + p]; act; y (+)= v */
    const float* alpha; const float* bias; const float* res; long res_bs;
    int act; int accumulate;
    int precision;
    /* output sub-lattice (0,0,0 or step 1 = every pixel): only outputs (sub_oy + a*sub_step, sub_ox + b*sub_step)
       are computed.  gd_conv2d uses it internally to split the data gradient of a STRIDED convolution into
       stride^2 launches, one per output parity class, each visiting only the taps that can reach that class
       (a stride-2 3x3 gradient then does 9/4 instead of 9 tap-GEMMs per pixel). */
    int sub_oy, sub_ox, sub_step;
} gd_conv_desc;
int gd_conv2d(const gd_conv_desc* d, void* stream);

/* 3x3 / stride 1 / pad 1 fast path (bf16 MFMA): the input patch of a TH x 32 pixel tile is staged once in LDS
 * and reused by all nine taps; weights are pre-packed into `ws` (gd_conv3x3_ws_bytes(M, Ck) bytes, caller
 * owned).  Same descriptor and epilogue contract as gd_conv2d; serves the forward conv (transposed = 0) and
 * the data gradient (transposed = 1).  gd_conv3x3_eligible() tells whether a descriptor qualifies. */
size_t gd_conv3x3_ws_bytes(int M, int Ck);
int gd_conv3x3_eligible(const gd_conv_desc* d);
int gd_conv3x3(const gd_conv_desc* d, void* ws, size_t ws_bytes, void* stream);
/* weight gradient of the same convolution (bf16 MFMA, fp32 atomics across the pixel splits):
 * dw (Cout, Cin, 3, 3) = sum_{b,p} dy[b][co][p] * relu?(x*in_scale + in_shift)[b][ci][p (+) tap]; dw is overwritten.
 * dy_bf16 (may be NULL): a dense bf16 copy (B, Cout, Ho, Wo) of dy made by the caller (gd_pack_bf16); it is then read
 * instead of dy -- worth it when Cin spans several 32-channel chunks, each of which re-reads every dY tile.
 * x_nhwc16 (may be NULL; needs in_scale == NULL): a dense PIXEL-MAJOR bf16 copy (B, H, W, x_ld) of x (the transposed
 * output of gd_pack_bf16, x_ld = Cin rounded up to 8); the patch staging is then 16-byte copies instead of strided
 * 4-byte gathers + converts -- worth it for wide convs (the 2C -> C fuse convs of DANetAttention).  x_ld may exceed Cin: the
 * rows of a wider pixel-major tensor (the split-bf16 pack's [hi | lo | hi] channels) are then read from a column offset.
 * accumulate != 0: dw is added to instead of overwritten (the three launches of a split-bf16 weight gradient). */
int gd_conv3x3_wgrad(const float* dy, long dy_bs, const void* dy_bf16, const float* x, long x_bs, const void* x_nhwc16,
                     int x_ld, const float* in_scale, const float* in_shift, int in_relu, int B, int Cout, int Cin, int H,
                     int W, int stride, int accumulate, float* dw, void* stream);   /* H, W = INPUT size; stride 1 or 2 (pad 1) */

/* ------------------------------------------------------------------------------------------
 * "NT" GEMM with the long reduction split over workgroups:
 *     C[b][m][n] (+)= alpha * sum_k A[b][m][k] * B~[b][n][k]        (k contiguous in both)
 * k runs over `kseg` segments of `klen` elements (segment = one image of a batch).
 * Serves the convolution weight gradient (B~ = im2col rows, k = output pixels), CAM's Gram
 * matrix X X^T (generator.py:133), nn.Linear forward (discriminator.py:76-77) and the
 * unfused fp32 PAM products (generator.py:117,120).
 * Partial sums of different k-splits are combined with fp32 atomics.
 * ---------------------------------------------------------------------------------------- */
typedef struct gd_gemm_nt_desc {
    int B, M, N;
    int kseg; long klen;
    /* A element (b, m, s, kk) at a[b*a_bs + s*a_ss + m*lda + kk] */
    const float* a; long a_bs, a_ss, lda;
    /* B plain (im2col == 0): element (b, n, s, kk) at bm[b*b_bs + s*b_ss + n*ldb + kk] */
    const float* bm; long b_bs, b_ss, ldb;
    /* B as im2col rows (im2col == 1): n = c*ks*ks + tap, kk = oy*Wo + ox of an Ho x Wo output;
       element = X~[s][c][oy*stride - pad + kh][ox*stride - pad + kw], X at bm[s*b_ss + c*Hi*Wi + ...] */
    int im2col, ks, stride, pad, Hi, Wi, Ho, Wo;
    /* optional X~ = relu?(X * in_scale[c] + in_shift[c]); the channel is c of the im2col row, or the B row n itself
       when im2col == 0 (1x1 weight gradient with a fused BatchNorm+ReLU input) */
    const float* in_scale; const float* in_shift; int in_relu;
    /* C element (b, m, n) at c[b*c_bs + m*ldc + n]; fp32 */
    float* c; long c_bs, ldc;
    const float* alpha;  /* device scalar or NULL */
    const float* bias;   /* per-n bias (added once) or NULL */
    int accumulate;      /* 0: C is overwritten (zeroed first when splits > 1);  1: C += */
    int splits;          /* k-splits (>=1); 0 = library picks */
    int precision;
} gd_gemm_nt_desc;
int gd_gemm_nt(const gd_gemm_nt_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d (generator.py:32,61,149,189,219,223).  x is (B, C, H, W) with batch stride x_bs.
 * ---------------------------------------------------------------------------------------- */
/* per-channel batch statistics -> mean[C], invstd[C]; if running_* != NULL they are blended
 * (momentum, unbiased variance) exactly like nn.BatchNorm2d in train mode.
 * ws: workspace of gd_bn_stats_ws_floats(...) floats. */
size_t gd_bn_stats_ws_floats(int B, int C, long HW);
int gd_bn_stats(const float* x, long x_bs, int B, int C, long HW, float eps, float momentum,
                float* mean, float* invstd, float* running_mean, float* running_var, float* ws, void* stream);
/* scale[c] = gamma[c]*invstd[c], shift[c] = beta[c] - mean[c]*scale[c]  (the folded affine) */
int gd_bn_fold(const float* gamma, const float* beta, const float* mean, const float* invstd, int C,
               float* scale, float* shift, void* stream);
/* SyncBN (optional; torch.nn.SyncBatchNorm semantics on the BatchNorm2d sites of generator.py:32,61,149,189,
 * 219,223): gd_bn_stats_local reduces this rank's shard to stats (C, 3) = (count, mean, M2) per channel; the caller
 * all-gathers the records of all ranks into stats_all (world, C, 3); gd_bn_stats_merge Chan-merges them (fp64) into the
 * GLOBAL batch's mean / invstd (+ running statistics, unbiased variance of the global count).  Backward: gd_bn_act_bwd
 * with dx = NULL gives this rank's dgamma / dbeta (they are also the parameter gradients); after their all-reduce (sum)
 * gd_bn_act_bwd_dx forms dx with inv_n = 1 / (global batch * HW). */
int gd_bn_stats_local(const float* x, long x_bs, int B, int C, long HW, float* stats, float* ws, void* stream);
int gd_bn_stats_merge(const float* stats_all, int world, int C, float eps, float momentum, float* mean, float* invstd,
                      float* running_mean, float* running_var, void* stream);
int gd_bn_act_bwd_dx(const float* dy, long dy_bs, const float* x, long x_bs, const float* scale, const float* shift,
                     const float* mean, const float* invstd, const float* dgamma_sum, const float* dbeta_sum, float inv_n,
                     int B, int C, long HW, int act, float* dx, long dx_bs, int accumulate_dx, void* stream);
/* eval mode: invstd from running_var (also written to invstd_out when non-NULL) */
int gd_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                    float eps, int C, float* scale, float* shift, float* invstd_out, void* stream);
/* out[c] (+)= sum over (B, HW) of x[b][c][:]  (bias gradients); ws: gd_bn_stats_ws_floats(B, C, HW) floats */
int gd_channel_sum(const float* x, long x_bs, int B, int C, long HW, float* out, int accumulate, float* ws,
                   void* stream);
/* y = act(x*scale[c] + shift[c]) */
int gd_affine_act(const float* x, long x_bs, const float* scale, const float* shift, int B, int C, long HW,
                  int act, float* y, long y_bs, void* stream);
/* backward of y = act(bn(x)): given dy (grad of y), x, the folded scale/shift, mean, invstd, gamma:
 * dgamma[C], dbeta[C] (overwritten) and dx (+)= ... ; ws as gd_bn_stats_ws_floats. */
int gd_bn_act_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* scale, const float* shift,
                  const float* mean, const float* invstd, const float* gamma, int B, int C, long HW, int act,
                  int train, float* dgamma, float* dbeta, float* dx, long dx_bs, int accumulate_dx, float* ws,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Resampling (nn.Upsample bicubic generator.py:221,225; F.interpolate bilinear generator.py:244;
 * bicubic down-sampling GAN_DANet_train.ipynb:L226,L231; VGG max-pool).
 * rscale_* = the coordinate scale ATen uses (1/scale_factor, or in/out when a size was given).
 * ---------------------------------------------------------------------------------------- */
int gd_bicubic_fwd(const float* x, int BC, int Hi, int Wi, float* y, int Ho, int Wo, float rscale_h,
                   float rscale_w, void* stream);
int gd_bicubic_bwd(const float* dy, int BC, int Hi, int Wi, float* dx, int Ho, int Wo, float rscale_h,
                   float rscale_w, void* stream);
/* last step of the generator tail (generator.py:242-247) in its collapsed form: final = conv3x3(C -> 1, pad 1) of
 * up2(c) + resize4(s) equals bias + sum over the nine taps of SHIFTED "tap planes" u (B, 9, H, W), each the channel
 * contraction sum_ch w[ch][tap] (.) done at low resolution and then resized (linear maps commute): only 9 planes instead
 * of C channels live at the output resolution.  y (B, 1, H, W); the backward scatters dy into the nine shifted planes. */
int gd_shift_sum9_fwd(const float* u, const float* bias, float* y, int B, int H, int W, void* stream);
int gd_shift_sum9_bwd(const float* dy, float* du, int B, int H, int W, void* stream);
/* input preamble of the train step (GAN_DANet_train.ipynb:L218-224), one launch, no cat pass:
 *   out (B, C1+C2, Ho, Wo) = cat([F.interpolate(lr (B,C1,H1,W1), scale_factor=1/rs1, mode='bicubic'),
 *                                 F.interpolate(aux (B,C2,H2,W2), scale_factor=1/rs2, mode='bicubic')], dim=1)
 * align_corners=False, antialias off; rs = input/output coordinate ratio (2.0 and 4.0 in the notebook). */
int gd_combine_inputs(const float* lr, int C1, int H1, int W1, float rs1, const float* aux, int C2, int H2, int W2,
                      float rs2, float* out, int B, int Ho, int Wo, void* stream); /* dx (BC,Hi,Wi) overwritten */
/* y = bilinear(x) (+ y if accumulate) ; or, when res != NULL, y = res + bilinear(x) in one pass: the skip
 * addition of generator.py:245 without a separate copy */
int gd_bilinear_fwd(const float* x, int BC, int Hi, int Wi, float* y, int Ho, int Wo, int accumulate,
                    const float* res, void* stream);
int gd_bilinear_bwd(const float* dy, int BC, int Hi, int Wi, float* dx, int Ho, int Wo, void* stream);
int gd_maxpool2_fwd(const float* x, int BC, int Hi, int Wi, float* y, void* stream);
int gd_maxpool2_bwd(const float* x, const float* dy, int BC, int Hi, int Wi, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pointwise / small reductions
 * ---------------------------------------------------------------------------------------- */
/* y = act(x) ; dx = dy * act'(y) computed from the OUTPUT y (valid for relu / leaky) */
int gd_act_fwd(const float* x, float* y, long n, int act, void* stream);
int gd_act_bwd(const float* y, const float* dy, float* dx, long n, int act, void* stream);
/* y = a*x + b*y  (a, b host scalars) */
int gd_axpby(const float* x, float a, float* y, float b, long n, void* stream);
/* y = (*s_dev) * x  (scale by a device scalar: loss weights / upstream scalar gradients, no host sync) */
int gd_scale_dev(const float* x, const float* s_dev, float* y, long n, int accumulate, void* stream);
/* dst[b][r][c] = src[b][r][c], r < R, c < Cc, independent batch strides and leading dimensions */
int gd_copy_rows(const float* src, long s_bs, long s_ld, float* dst, long d_bs, long d_ld, int B, int R, int Cc,
                 void* stream);
/* strided copy of a (B, C, HW) block: dst[b*d_bs + i] (+)= src[b*s_bs + i], i < C*HW */
int gd_copy_slab(const float* src, long s_bs, float* dst, long d_bs, int B, long chw, int accumulate, void* stream);
/* row softmax, in place capable: y[r][:] = softmax(sign * x[r][:]) over `cols` (CAM uses sign=-1:
 * softmax(max - E) == softmax(-E), generator.py:134-135) */
int gd_softmax_rows(const float* x, float* y, long rows, int cols, float sign, void* stream);
/* dx[r][:] = sign * p .* (dp - sum(dp .* p)) */
int gd_softmax_rows_bwd(const float* p, const float* dp, float* dx, long rows, int cols, float sign, void* stream);
/* out[0] (+)= sum_i a[i]*b[i]  (b == NULL: sum_i a[i]) -- gamma gradients ; ws >= 1024 floats */
int gd_dot(const float* a, const float* b, long n, float* out, int accumulate, float* ws, void* stream);
/* t[b][j][i] = s[b][i][j] : (B, R, Cc) -> (B, Cc, R) */
int gd_transpose(const float* s, float* t, int B, int R, int Cc, void* stream);
/* out = a + a^T for (B, n, n) */
int gd_add_transpose(const float* a, float* out, int B, int n, void* stream);

/* losses: value -> out[0] (fp32), gradient written when the pointer is non-NULL.  ws >= 2048 floats.
 * BCEWithLogits mean vs a constant label (GAN_DANet_train.ipynb:L190,L252-253,L261) */
int gd_bce_logits(const float* z, long n, float label, float* out, float* dz, float* ws, void* stream);
/* the same against a per-element target tensor t (n); dz / dt (either may be NULL): gradients of the mean */
int gd_bce_logits_target(const float* z, const float* t, long n, float* out, float* dz, float* dt, float* ws,
                         void* stream);
/* LeakyReLU with an arbitrary negative slope (the fused conv / linear epilogues implement the reference's 0.2 only);
 * backward takes the INPUT x. */
int gd_leaky_fwd(const float* x, float* y, long n, float slope, void* stream);
int gd_leaky_bwd(const float* x, const float* dy, float* dx, long n, float slope, void* stream);
/* MSELoss / L1 mean (L191,L262; losses.py:72) */
int gd_mse(const float* a, const float* b, long n, float* out, float* da, float* ws, void* stream);
int gd_l1(const float* a, const float* b, long n, float* out, float* da, float* ws, void* stream);
/* TVLoss.forward (losses.py:81-87) */
int gd_tv(const float* x, int B, int C, int H, int W, float weight, float* out, float* dx, float* ws, void* stream);
/* SSIM mean (losses.py:109-136), forward only (the train loop never differentiates it) */
int gd_ssim(const float* a, const float* b, int BC, int H, int W, int window, float* out, float* ws, void* stream);
/* per-sample SSIM means (SSIM(size_average=False), losses.py:136): out (B) */
int gd_ssim_samples(const float* a, const float* b, int B, int C, int H, int W, int window, float* out, float* ws,
                    void* stream);
/* SSIM backward (losses.py:118-136 under autograd): gscale (B) = upstream gradient of each sample's pixels (already
 * divided by the element count of the mean); coef_ws: caller-owned scratch of 4 * B*C*H*W floats; da / db (either may
 * be NULL): gradients w.r.t. img1 / img2, overwritten. */
int gd_ssim_bwd(const float* a, const float* b, const float* gscale, int B, int C, int H, int W, int window,
                float* coef_ws, float* da, float* db, void* stream);

/* AdamW step over one tensor (torch.optim.AdamW, GAN_DANet_train.ipynb:L182-183).
 * `step` is 1-based.  The gradient is read once, multiplied by grad_scale (1/world for DP). */
int gd_adamw(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2,
             float eps, float weight_decay, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pixel-major (NHWC) bf16 kernels for the frozen VGG19 feature stack of PerceptualLoss (losses.py:13-73;
 * torchvision vgg19.features[:21]).  Activations (B, H, W, C) bf16, C % 8 == 0.
 *   gd_conv3x3_nhwc_pack : w (Cout, Cin, 3, 3) fp32 -> packed bf16 operator in ws; transposed = 1 packs the stride-1
 *                          data-gradient operator (channels swapped, taps flipped), transposed = 2 the stride-2 one
 *                          (channels swapped, taps as stored: gd_conv3x3_nhwc_s2_dgrad).  ws needs
 *                          gd_conv3x3_ws_bytes(M, K) bytes, (M, K) = (Cout, Cin) or (Cin, Cout).
 *   gd_conv3x3_nhwc      : y = [mask > 0] * act(conv3x3_p1(x) + bias) + res   (x: K channels, y/mask/res: M channels;
 *                          bias, mask, res may be NULL; mask = the ReLU output whose backward is being applied)
 *   gd_nhwc_stem_fwd/bwd : first conv (features[0], Ci <= 4 input channels) from / to the fp32 NCHW image
 *   gd_nhwc_maxpool2_*   : nn.MaxPool2d(2); backward takes the pooled layer's input x (first maximum wins, ATen's
 *                          tie rule) and optionally gates by x > 0
 *   gd_nhwc_l1           : out (+)= mean |a - b| (losses.py:72), ws >= 1024 floats
 *   gd_nhwc_l1_grad      : g = (*upstream) / n * sign(a - b), optionally gated by a > 0
 *   split / split_c      : operand mode "x3": the pixel-major tensors hold 3 C bf16 per pixel, [hi | lo | hi] of the C fp32
 *                          values (see gd_disc_stem_fwd below); split_c = C (0 = plain), n = the LOGICAL element count
 * ---------------------------------------------------------------------------------------- */
int gd_conv3x3_nhwc_pack(const float* w, int Cout, int Cin, int transposed, void* ws, size_t ws_bytes, void* stream);
int gd_conv3x3_nhwc(const void* x, const void* wpack, const float* bias, const void* mask, const void* res, void* y, int B,
                    int H, int W, int K, int M, int relu, int split, void* stream);
/* the same kernel with an fp32 NCHW result (B, M, H, W), batch stride y_bs elements: wide 3x3 convs of the generator on a
 * pixel-major bf16 copy of their input (gd_pack_16 transposed output, K = its leading dimension, zero padded) */
int gd_conv3x3_nhwc_f32out(const void* x, const void* wpack, const float* bias, float* y32, long y_bs, int B, int H, int W, int K,
                           int M, int relu, void* stream);
int gd_nhwc_stem_fwd(const float* img, int B, int Ci, int H, int W, const float* w, const float* bias, int Co, int relu,
                     void* y, int split, void* stream);
int gd_nhwc_stem_bwd(const void* g, int B, int Ci, int H, int W, const float* w, int Co, float* dimg, int split, void* stream);
int gd_nhwc_maxpool2_fwd(const void* x, int B, int H, int W, int C, void* y, int split, void* stream);
int gd_nhwc_maxpool2_bwd(const void* x, const void* dy, int B, int H, int W, int C, int relu_mask, void* dx, int split, void* stream);
int gd_nhwc_l1(const void* a, const void* b, long n, float* out, int accumulate, float* ws, int split_c, void* stream);
int gd_nhwc_l1_grad(const void* a, const void* b, long n, const float* upstream, int relu_mask, void* g, int split_c, void* stream);

/* ------------------------------------------------------------------------------------------
 * Discriminator1 (discriminator.py:57-77: 4 x (conv3x3 stride 2 + LeakyReLU(0.2)) -> flatten -> fc1 -> fc2) on
 * pixel-major bf16 activations.  Ho = (H-1)/2+1, Wo = (W-1)/2+1 throughout.
 *   gd_disc_stem_fwd        : conv1 (discriminator.py:60) from the fp32 NCHW image (B, Ci <= 4, H, W):
 *                             y (B, Ho, Wo, Co) bf16 = LeakyReLU(conv3x3_s2_p1(img) + bias)
 *   gd_disc_stem_wgrad      : dw (Co, Ci, 3, 3), db (Co) fp32 (overwritten; db may be NULL) from the pre-activation
 *                             gradient g (B, Ho, Wo, Co) bf16 and the image.  Co == 64.
 *   gd_disc_stem_dgrad      : dimg (B, Ci, H, W) fp32 from g (the generator step differentiates through D)
 *   gd_conv3x3_nhwc_s2      : conv2..4 (discriminator.py:61-63): x (B, H, W, K) -> y (B, Ho, Wo, M), + bias,
 *                             act 0 none / 1 ReLU / 2 LeakyReLU(slope); wpack = gd_conv3x3_nhwc_pack(transposed = 0)
 *   gd_conv3x3_nhwc_s2_dgrad: its data gradient, split by input-pixel parity (nine tap passes, no zero-stuffing):
 *                             dx (B, H, W, K) = convT(dy) * LeakyReLU'(act_out), act_out (B, H, W, K) = the activation
 *                             output this conv consumed (NULL: no mask); wpack_t = gd_conv3x3_nhwc_pack(transposed = 2)
 *   gd_nhwc_flatten_fwd/bwd : x.flatten(1) (discriminator.py:72): y (B, HW, C) bf16 -> f (B, C*HW) fp32 in (c, h, w)
 *                             order; backward g (B, HW, C) bf16 = df * LeakyReLU'(y)
 *   gd_nhwc_to_nchw16       : g (B, HW, C) bf16 -> gt (B, C, HW) bf16 (the dy_bf16 operand of gd_conv3x3_wgrad) and,
 *                             when csum != NULL, csum (C) fp32 = per-channel sums (the bias gradient; overwritten)
 * ----------------------------------------------------------------------------------------  *   split (all eight entry points; operand mode "x3" of set_precision("mixed")): 1 = every pixel-major bf16 ACTIVATION or
 *   GRADIENT tensor named above holds 3 C channels per pixel, the C fp32 values split as [hi | lo | hi] with hi = bf16(v),
 *   lo = bf16(v - hi): the operand the conv kernels take unchanged against weights split [hi ; hi ; lo] along the contraction
 *   axis (gd_split3_weights), three bf16 MFMAs per product, ~2^-16 relative.  The conv entry points then take K (forward)
 *   resp. M (data gradient) = the 3 C PHYSICAL channel count of their input and the logical count of their output;
 *   gd_nhwc_to_nchw16 writes two (B, C, HW) images, hi then lo, and the sums of hi + lo.  0 = plain bf16 (C channels).
 */
int gd_disc_stem_fwd(const float* img, int B, int Ci, int H, int W, const float* w, const float* bias, int Co, float slope,
                     void* y, int split, void* stream);
int gd_disc_stem_wgrad(const void* g, const float* img, int B, int Ci, int H, int W, int Co, float* dw, float* db, int split, void* stream);
int gd_disc_stem_dgrad(const void* g, int B, int Ci, int H, int W, const float* w, int Co, float* dimg, int split, void* stream);
int gd_conv3x3_nhwc_s2(const void* x, const void* wpack, const float* bias, void* y, int B, int H, int W, int K, int M, int act,
                       float slope, int split, void* stream);
int gd_conv3x3_nhwc_s2_dgrad(const void* dy, const void* wpack_t, const void* act_out, float slope, void* dx, int B, int H,
                             int W, int K, int M, int split, void* stream);
int gd_nhwc_flatten_fwd(const void* y, int B, int HW, int C, float* f, int split, void* stream);
int gd_nhwc_flatten_bwd(const float* df, const void* y, float slope, int B, int HW, int C, void* g, int split, void* stream);
int gd_nhwc_to_nchw16(const void* g, int B, int HW, int C, void* gt, float* csum, int split, void* stream);

/* ------------------------------------------------------------------------------------------
 * PAM, fused (flash) form with 16-bit MFMA operands and fp32 softmax statistics (generator.py:115-122).
 * f16 = 0: bf16 operands (training default); f16 = 1: IEEE fp16 operands (BASELINE config 5) -- the packs below
 * must then have been made with gd_pack_16(..., f16 = 1).
 *   qt     : (B, Npad, 32), d zero-padded to 32, values PRE-SCALED by log2(e)  (gd_pack_bf16 scale_imm):
 *            the kernels work in the log2 domain and feed the softmax shift in as the MFMA accumulator input
 *   kt     : (B, Npad, 32), unscaled, d zero-padded to 31 and d = 31 set to 1.0 (gd_pack_bf16 ones_row = 31):
 *            the forward feeds its running row maximum through that k-slot (so r <= 31)
 *   v      : (B, Cp, Npad), Cp = C rounded up to 32   (channel-major, keys perm16-ordered: gd_pack_bf16)
 *   v_ones : != 0 when channel Cp-1 of v is a row of ones (gd_pack_bf16 ones_row; needs C < Cp): the softmax
 *            denominator then comes out of the O MFMAs instead of one VALU add per score
 *   x, out : (B, C, N) fp32 with batch strides; out = gamma * attn + x
 *   o_attn : (B, C, N) fp32 un-scaled attention output (kept for backward)
 *   lse    : (B, N) fp32 natural-log log-sum-exp of the unscaled energies.   Npad % 256 == 0.
 * ---------------------------------------------------------------------------------------- */
int gd_pam_flash_fwd(const void* qt, const void* kt, const void* v, int B, int N, int Npad, int C, int Cp,
                     int v_ones, int f16, const float* gamma, const float* x, long x_bs, float* out, long out_bs,
                     float* o_attn, float* lse, const float* k_sqnorm_max, void* stream);
/* gd_pam_flash_fwd (generator.py:115-122) with the max-free sweep for logits of ANY magnitude, bf16 operands: softmax is
 * invariant under a per-query shift, so a prepass takes m_i = max over `nsample` (128 | 256 | 512) strided keys of q_i . k_j
 * -- at most the true row maximum, hence row sums >= ~1 -- and the sweep runs exp2(s - m_i) without row maximum, test or
 * rescale.  Workgroups whose row sums leave (1e-30, 1e30) (the sample missed the true maximum by > ~100 log2 units) are
 * flagged and redone by the running-maximum sweep in a second launch in which every other workgroup exits at once.
 * ws: gd_pam_fwd_shift_ws_bytes(B, Npad) bytes of scratch.  Same outputs as gd_pam_flash_fwd. */
size_t gd_pam_fwd_shift_ws_bytes(int B, int Npad);
int gd_pam_flash_fwd_shift(const void* qt, const void* kt, const void* v, int B, int N, int Npad, int C, int Cp, int v_ones,
                           const float* gamma, const float* x, long x_bs, float* out, long out_bs, float* o_attn, float* lse,
                           int nsample, void* ws, size_t ws_bytes, void* stream);
/* k_sqnorm_max (B floats, or NULL): max_j |k_j|^2 of each image's packed keys.  Softmax is shift-invariant; when
 * |q_i| * max_j |k_j| (in log2 units, q is pre-scaled) stays inside the exponent range of the P operand type for every
 * query of a wave, that wave sweeps the keys without a running maximum (no per-tile max / test / rescale).  NULL: the
 * running maximum is always kept.  out (B floats) is overwritten. */
int gd_pam_key_sqnorm_max(const void* kt, int B, int N, int Npad, int f16, float* out, void* stream);
/* backward: 16-bit inputs qt, kt as above (B,Npad,32); kn (B,32,Npad) perm16-ordered (row 31 is don't-care);
 * vt (B,Npad,Cp); dot (B,Npad,Cp) = gamma*dOut; lse, delta (B,N) fp32 (delta = gamma*rowsum(dOut.*O)).  All packs
 * zero padded (gd_pack_bf16 does).  Outputs fp32, channel-major, overwritten: dqn, dkn (B,32,Npad), dv (B,Cp,Npad)
 * -- gradients w.r.t. the UNSCALED q, k, v.  Npad % 256 == 0.
 * form (GD_PAM_BWD_*):
 *   0 K64_ATOMIC  one key-parallel kernel, 4 waves x 64 keys per workgroup (one wave per SIMD, dK^T/dV^T in the
 *                 accumulation registers), dQ summed across key blocks with fp32 atomics (fastest; not bitwise
 *                 reproducible in dQ)
 *   1 K64_PARTS   the same kernel storing dQ as one bf16 part per 256-key block + a streaming sum (reproducible)
 *   2 K32_PARTS   8 waves x 32 keys per workgroup, bf16 dQ parts (the round-1 kernel; reproducible; bf16 only)
 *   3 TWO_KERNEL  dK/dV kernel, then a query-parallel kernel that recomputes S and dP for dQ (no scratch; bf16 only)
 * out_bs: 0, or (form 0 only) the batch stride in elements shared by dqn, dkn, dv when the caller passes them as row
 *   blocks of ONE (B, rows, Npad) buffer -- the three projection gradients can then run as one GEMM over that buffer.
 * scratch: caller-owned, >= gd_pam_bwd_scratch_bytes(Npad, form) (= one image's worth; more lets more images go per
 * launch); may be NULL for form 3. */
#define GD_PAM_BWD_K64_ATOMIC 0
#define GD_PAM_BWD_K64_PARTS 1
#define GD_PAM_BWD_K32_PARTS 2
#define GD_PAM_BWD_TWO_KERNEL 3
size_t gd_pam_bwd_scratch_bytes(int Npad, int form);
/* tuning hook (bench tooling; process-global, not thread safe): schedule variant of the K64 kernel (0 = production)
 * and the number of a wave's two key tiles whose V rows stay in registers (1 or 2; 0 = default). */
void gd_pam_k64_variant(int order, int vreg);
/* diagnostic variants 5 / 6 only (tools/pam_stamps.py): device buffer of images x key blocks x 4 waves x 12 uint32 that
 * receives the per-wave cycle sums of the tile loop's segments; NULL (default) = no output */
void gd_pam_k64_debug(void* buf);
int gd_pam_flash_bwd(const void* qt, const void* kt, const void* kn, const void* vt, const void* dot_,
                     const float* lse, const float* delta, int B, int N, int Npad, int Cp, int f16, int form,
                     float* dqn, float* dkn, float* dv, long out_bs, void* scratch, size_t scratch_bytes, void* stream);

/* test.ipynb c1:69-85 mild_histogram_matching, per sample of a batch: out[b] = (1 - weight) * src[b] + weight *
 * interp(cdf_src(src[b]), cdf_ref, sorted unique ref[b]) with numpy's np.unique / np.interp semantics (float64 result, as
 * the notebook produces).  src (B, ns), ref (B, nt) fp32; out (B, ns) fp64; ws: caller-owned scratch of
 * gd_hist_match_ws_bytes(ns, nt) bytes (sort buffers).  One radix sort pair per sample (rocPRIM via hipCUB). */
size_t gd_hist_match_ws_bytes(long ns, long nt);
int gd_hist_match(const float* src, const float* ref, int B, long ns, long nt, double weight, double* out, void* ws,
                  size_t ws_bytes, void* stream);
/* test.ipynb c1:87-101 smooth_blend: over the region rows [sr, er) x columns [sc, ec) of every (b, c) plane,
 * gen = gen * (1 - mask) + grace * mask, in place; mask (er-sr, ec-sc) fp32 is the feathered window the host builds. */
int gd_blend_region(float* gen, const float* grace, const float* mask, int BC, int H, int W, int sr, int er, int sc,
                    int ec, void* stream);
/* CustomDataset.apply_augmentation (datasets.py:181-208) for a batch of tiles as one gather: per-sample op word
 * ops[b] = hflip | vflip << 1 | quarter_turns << 2 | noise << 4 (flip W, flip H, torch.rot90 k, in that order; H == W
 * when a sample is turned an odd number of times -- checked by the host).  noise (same shape as dst) may be NULL;
 * it is added times noise_scale (0.05 in the reference) to the samples whose noise bit is set.  src != dst. */
int gd_augment_d4(const float* src, float* dst, int B, int C, int H, int W, const int* ops, const float* noise,
                  float noise_scale, void* stream);
/* attention gates of SqueezeExcitation / CBAMBlock (generator.py:70-101; exported by the reference, not on the
 * train path).  Dense (B, C, HW) fp32.
 *   gd_bcast_mul       : y = x * att; mode 0: att (B, C) channel gate (generator.py:84), mode 1: att (B, HW)
 *                        spatial gate (generator.py:101).  The data gradient is the same call on dy.
 *   gd_row_dot         : out[row] = sum_j a[row][j] * b[row][j]   (gate gradient of mode 0; mode 1 uses gd_chan_dot)
 *   gd_chan_maxmean_*  : y (B, 2, HW) = [max_c x, mean_c x] (generator.py:98-100), idx (B, HW) = arg max;
 *                        backward routes dy[:,0] to the arg-max channel and spreads dy[:,1] / C */
int gd_bcast_mul(const float* x, const float* att, float* y, int B, int C, long HW, int mode, void* stream);
int gd_row_dot(const float* a, const float* b, float* out, long rows, long n, void* stream);
int gd_chan_maxmean_fwd(const float* x, float* y, int* idx, int B, int C, long HW, void* stream);
int gd_chan_maxmean_bwd(const float* dy, const int* idx, float* dx, int B, int C, long HW, void* stream);
/* d_raw[b][i] = sum_c a[b][c][i]*o[b][c][i] (per-pixel channel dot), delta = (*gamma) * d_raw */
int gd_chan_dot(const float* a, long a_bs, const float* o, long o_bs, int B, int C, int N, const float* gamma,
                float* d_raw, float* delta, void* stream);
/* split-bf16 ("x3") operands of set_precision("mixed"): every conv operand v = hi + lo (hi = bf16(v), lo = bf16(v - hi)),
 * every product hi*hi + lo*hi + hi*lo on the bf16 matrix pipe with fp32 accumulation (~2^-16 relative, 5x the rate of the
 * exact f32 MFMA).  The split rides in the operand layout, so the 16-bit conv kernels above serve unchanged:
 *   gd_pack_16_split  : fp32 (B, R, Cc) planes (optionally max(0, row_scale * x + row_shift) first: the BatchNorm + ReLU
 *                       prologue of a dense layer, generator.py:36) -> up to three bf16 copies of the tile per output, copy j
 *                       at `cs` elements behind copy j - 1, holding the lo part when bit j of `pattern` is set, else the hi
 *                       part.  plain = (R, Cc) rows (channel-major, ldp == Cc), tr = the transpose (pixel-major rows of ldt
 *                       elements): [hi | lo | hi] as 3 R channels (cs = R, ldt = 3 R, pattern 0b010) is the forward /
 *                       data-gradient operand; separate hi / lo images (pattern 0b10) feed gd_conv3x3_wgrad's three
 *                       accumulating launches.  R % 8 == 0, Cc % 8 == 0.
 *   gd_split3_weights : w (A, Bn, Cn) fp32 -> (A, 3 Bn, Cn) fp32 [hi ; hi ; lo] along the middle axis (nn.Conv2d weights,
 *                       generator.py:34,148,218,222 and the VGG stack of losses.py:41: Bn = Cin for the forward, A = 1 and
 *                       Bn = Cout for the data gradient); gd_conv3x3_nhwc_pack then makes the 16-bit operator. */
int gd_pack_16_split(const float* s, long s_bs, int B, int R, int Cc, const float* row_scale, const float* row_shift, int relu,
                     void* plain, long p_bs, int ldp, long p_cs, int p_ncopy, int p_pattern, void* tr, long t_bs, int ldt,
                     long t_cs, int t_ncopy, int t_pattern, void* stream);
/* the same with the ReLU backward fused (autograd of nn.ReLU behind a conv, losses.py:41 / generator.py:219): `mask` (B, R, Cc)
 * fp32, batch stride m_bs, is the ReLU's output; elements whose mask value is not positive are packed as zero */
int gd_pack_16_split_masked(const float* s, long s_bs, int B, int R, int Cc, const float* row_scale, const float* row_shift,
                            int relu, void* plain, long p_bs, int ldp, long p_cs, int p_ncopy, int p_pattern, void* tr, long t_bs,
                            int ldt, long t_cs, int t_ncopy, int t_pattern, const float* mask, long m_bs, void* stream);
int gd_split3_weights(const float* w, long A, long Bn, long Cn, float* out, void* stream);
/* fp16 operand mode of gd_pam_flash_bwd (autograd of generator.py:115-122 under set_precision("fp16" | "mixed")): IEEE
 * fp16 loses everything below 6e-8, and gamma * dOut of a real training step sits below that.  All outputs of the backward
 * are linear in dOut, so it is packed as scales[0] * dOut with scales[0] = gamma * 2^k chosen so that the largest element has
 * magnitude in [0.5, 1); delta (B, N), computed by gd_chan_dot with gamma, is multiplied by 2^k in place; the consumers of
 * dQ / dK / dV multiply by scales[1] = 2^-k (the alpha of their GEMMs).  dout: (B, C, N) fp32, batch stride dout_bs;
 * scales: 2 floats; ws: 1024 floats. */
int gd_pam_f16_scale(const float* dout, long dout_bs, int B, int C, int N, const float* gamma, float* delta, float* scales,
                     float* ws, void* stream);
/* fp32 (B, R, Cc) planes (batch stride s_bs), times scale_imm and optionally times a device scalar -> bf16:
 *   plain      (B, Rp_plain, ld_plain)  zero padded copy          (NULL to skip)
 *   transposed (B, Ccp_t, ld_t)         zero padded transpose     (NULL to skip)
 * perm16 != 0: inside every 16 columns of `plain` the order is [0-3, 8-11, 4-7, 12-15] (the order an MFMA lane
 * half consumes an accumulator-row-ordered k-step: gd_pam_flash_fwd expects its V operand packed this way).
 * ones_row >= 0: that (padding) row of the source is taken as all ones in both outputs (-1: none). */
int gd_pack_bf16(const float* s, long s_bs, int B, int R, int Cc, const float* scale_dev, float scale_imm, void* plain,
                 int Rp_plain, int ld_plain, void* transposed, int Ccp_t, int ld_t, int perm16, int ones_row,
                 void* stream);
/* the same with the 16-bit output type selectable: f16 = 0 bf16 (== gd_pack_bf16), f16 = 1 IEEE fp16 */
int gd_pack_16(const float* s, long s_bs, int B, int R, int Cc, const float* scale_dev, float scale_imm, void* plain,
               int Rp_plain, int ld_plain, void* transposed, int Ccp_t, int ld_t, int perm16, int ones_row, int f16,
               void* stream);
/* gd_pack_16 with a per-row affine + ReLU applied first (row r -> max(0, row_scale[r] x + row_shift[r]) when relu): the
 * BatchNorm + ReLU prologue of a dense layer (generator.py:29-45), for the pixel-major bf16 copy of its input that
 * gd_conv3x3_wgrad reads (x_nhwc16).  Aligned shapes only: Cc % 4 == 0, ld % 8 == 0. */
int gd_pack_16_affine(const float* s, long s_bs, int B, int R, int Cc, const float* row_scale, const float* row_shift, int relu,
                      void* plain, int Rp_plain, int ld_plain, void* transposed, int Ccp_t, int ld_t, int f16, void* stream);

/* ------------------------------------------------------------------------------------------
 * RCCL communicator for hosts without torch.distributed (one process per GPU, one communicator per process; the
 * Python package here goes through torch.distributed's "nccl" backend, which IS RCCL, and does not call these).
 * librccl is dlopen()ed on first use (GD_RCCL_PATH overrides the search).  dtype: 0 fp32, 1 bf16, 2 fp16.  All
 * collectives are in-place-capable, sum-reducing, asynchronous on `stream`.
 *   gd_comm_unique_id : rank 0 creates the 128-byte id; the host distributes it (env, file, TCP store)
 *   gd_comm_init      : collective over `world` processes
 *   gd_allreduce      : buf (n elements) <- sum over ranks                     (gradient all-reduce, the G/D grads)
 *   gd_reduce_scatter : recv (recv_n) <- this rank's slice of the sum of send (world * recv_n)     (fc1 gradient)
 *   gd_allgather      : recv (world * send_n) <- every rank's send (send_n)                 (updated fc1 slices)
 * ---------------------------------------------------------------------------------------- */
int gd_comm_unique_id(char* id128);
int gd_comm_init(int rank, int world, const char* id128);
int gd_comm_world(void);
int gd_allreduce(void* buf, size_t n, int dtype, void* stream);
int gd_reduce_scatter(const void* send, void* recv, size_t recv_n, int dtype, void* stream);
int gd_allgather(const void* send, void* recv, size_t send_n, int dtype, void* stream);
int gd_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* GANDANET_H */
