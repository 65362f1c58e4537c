// 3x3 / stride 1 / pad 1 convolution for gfx950 with an LDS-resident input patch ("halo tile").
//
// The generic implicit-GEMM kernel (gemm_conv.hip) re-gathers every input element once per tap: 9x the load,
// convert and LDS-store work, which makes it VALU-bound.  Here a workgroup owns a TH x 32 pixel tile of one
// image and BM output channels.  Per 32-channel chunk it stages the (TH+2) x 34 input patch ONCE
// (fp32 -> optional BatchNorm affine + ReLU -> bf16, stored [pixel][channel]) and runs all nine taps from it:
// the B fragment of tap (ky,kx) for output pixel (y,x) is simply the 16-byte read at patch row (y+ky, x+kx),
// so every staged element feeds 9 MFMA k-steps.  Weights are pre-packed once per call into bf16
// [m-tile][chunk][tap][BM][32] (taps flipped / channels swapped for the data gradient), so the A tiles are
// plain 16-byte vector copies.  MFMA v_mfma_f32_32x32x16_bf16, fp32 accumulate; each MFMA n-tile is one image
// row segment of 32 pixels, so the epilogue stores 128-byte coalesced rows.
#include "common.h"
#include "tile_mma.h"
#include "../../include/gandanet.h"

namespace {

using gd::acc_row;
using gd::bf16x8_native_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int TW = 32;        // tile width in pixels (= MFMA N)
constexpr int PW = TW + 2;    // patch width
constexpr int CK = 32;        // channels per chunk
constexpr int LD = CK + 8;    // LDS row: 32 channels + 8 pad (80 B): conflict-free 16-B fragment reads

// ---- weight pre-pack: A[m][c][tap] (generic strides) -> bf16 wpack[mt][chunk][tap][BM][32] ----------------
__global__ void pack_w_kernel(const float* __restrict__ a, long a_sm, long a_sc, long a_st, int M, int Ck, int flip,
                              int BM, int nchunks, unsigned short* __restrict__ wp, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % CK);
        long t = i / CK;
        const int mm = (int)(t % BM);
        t /= BM;
        const int tap = (int)(t % 9);
        t /= 9;
        const int chunk = (int)(t % nchunks);
        const int mt = (int)(t / nchunks);
        const int m = mt * BM + mm, c = chunk * CK + cc;
        float v = 0.f;
        if (m < M && c < Ck) v = a[(long)m * a_sm + (long)c * a_sc + (long)(flip ? 8 - tap : tap) * a_st];
        wp[i] = gd_f2bf(v);
    }
}

// S = convolution stride (1, or 2 for the down-sampling convs of the discriminator: the patch then covers
// (2TH+1) x 65 input pixels and consecutive output pixels read patch rows two apart)
template <int BM, int TH, int S>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const gd_conv_desc d, const unsigned short* __restrict__ wp,
                                                             int tiles_x, int nchunks) {
    constexpr int WAVES_M = 2, WAVES_N = 2;
    constexpr int TM = BM / (32 * WAVES_M);       // 32-row m-tiles per wave
    constexpr int TN = TH / WAVES_N;              // image rows (n-tiles) per wave
    constexpr int PH = S * (TH - 1) + 3;
    constexpr int PWk = S * (TW - 1) + 3;
    constexpr int NPIX = PH * PWk;
    constexpr int WCHUNKS = 3 * BM * CK / 8;      // 16-byte vectors of one kernel row of weights
    constexpr int WPT = WCHUNKS / 256;            // per thread
    static_assert(WCHUNKS % 256 == 0, "weight stage must divide evenly");

    __shared__ __attribute__((aligned(16))) unsigned short patch[NPIX * LD];
    __shared__ __attribute__((aligned(16))) unsigned short wts[3 * BM * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z;
    const int mt = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = d.Hi, W = d.Wi;            // input size
    const int Ho = d.Ho, Wo = d.Wo;          // output size
    const long HW = (long)H * W, HWo = (long)Ho * Wo;
    const float* ximg = d.x + (long)b * d.x_bs;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const unsigned short* wbase = wp + (long)mt * nchunks * 9 * BM * CK;
    u32x4_t wreg[WPT];
    auto load_w = [&](int chunk, int ky) {
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(wbase + ((long)chunk * 9 + ky * 3) * BM * CK);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = src[tid + i * 256];
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int v = tid + i * 256;          // vector index inside [3][BM][32] (4 vectors per row)
            const int row = v >> 2, q = v & 3;    // row = kx*BM + m
            *reinterpret_cast<u32x4_t*>(wts + row * LD + q * 8) = wreg[i];
        }
    };

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * CK;
        load_w(chunk, 0);
        // ---- stage the input patch of this chunk: [pixel][channel] ----
        // work item = (patch pixel, channel half): bounds, address and LDS slot are computed once per item, then 8
        // channel pairs are walked with a constant stride (lanes = consecutive pixels: coalesced 34-pixel rows)
        for (int w = tid; w < 2 * NPIX; w += 256) {
            const int half = w >= NPIX ? 1 : 0;
            const int pix = w - half * NPIX;
            const int py = pix / PWk, px = pix - py * PWk;
            const int iy = S * y0 - 1 + py, ix = S * x0 - 1 + px;
            const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cb = c0 + half * 16;
            const float* p = ximg + (long)cb * HW + (long)iy * W + ix;
            unsigned int* dst = reinterpret_cast<unsigned int*>(patch + pix * LD + half * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cb + 2 * j;
                float v0 = 0.f, v1 = 0.f;
                if (inside) {
                    if (c < d.Ck) v0 = p[(long)(2 * j) * HW];
                    if (c + 1 < d.Ck) v1 = p[(long)(2 * j + 1) * HW];
                    if (d.in_scale) {
                        if (c < d.Ck) {
                            v0 = fmaf(v0, d.in_scale[c], d.in_shift[c]);
                            if (d.in_relu) v0 = fmaxf(v0, 0.f);
                        }
                        if (c + 1 < d.Ck) {
                            v1 = fmaf(v1, d.in_scale[c + 1], d.in_shift[c + 1]);
                            if (d.in_relu) v1 = fmaxf(v1, 0.f);
                        }
                    }
                }
                dst[j] = gd_pack_bf2(v0, v1);
            }
        }
        for (int ky = 0; ky < 3; ++ky) {
            store_w();
            __syncthreads();                       // weights of this kernel row (and, for ky = 0, the patch) visible
            if (ky < 2) load_w(chunk, ky + 1);     // next row's weights fly under the MFMAs
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int ks = 0; ks < CK / 16; ++ks) {
                    bf16x8_t fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[i] = *reinterpret_cast<const bf16x8_t*>(wts + (kx * BM + wm * TM * 32 + i * 32 + r) * LD +
                                                                  ks * 16 + 8 * h);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[j] = *reinterpret_cast<const bf16x8_t*>(patch + (((wn * TN + j) * S + ky) * PWk + r * S + kx) * LD +
                                                                  ks * 16 + 8 * h);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8_native_t, fa[i]), __builtin_bit_cast(bf16x8_native_t, fb[j]),
                                acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();                       // all reads of wts (and after ky = 2 of the patch) are done
        }
    }

    // ---- epilogue (same contract as gd_conv2d: alpha, bias, residual, activation, accumulate) ----
    const float alpha = d.alpha ? *d.alpha : 1.f;
    const int ox = x0 + r;
    if (ox < Wo) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int oy = y0 + wn * TN + j;
            if (oy >= Ho) continue;
            const long pn = (long)oy * Wo + ox;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mt * BM + wm * TM * 32 + i * 32 + acc_row(e, h);
                    if (m >= d.M) continue;
                    float v = acc[i][j][e] * alpha;
                    if (d.bias) v += d.bias[m];
                    if (d.res) v += d.res[(long)b * d.res_bs + (long)m * HWo + pn];
                    if (d.act == GD_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (d.act == GD_ACT_LEAKY02) v = v >= 0.f ? v : 0.2f * v;
                    float* yp = reinterpret_cast<float*>(d.y) + (long)b * d.y_bs + (long)m * HWo + pn;
                    if (d.accumulate) v += *yp;
                    *yp = v;
                }
        }
    }
}

}  // namespace

// output-channel tile: 64 or 128, whichever pads M least; with a single 32-channel chunk of input (dense-layer data
// gradients, K = 24) the kernel is staging / store bound, not MFMA bound, and the wide tile stages the patch fewer times
static inline int pick_bm(int M, int Ck) {
    if (Ck <= CK && M > 64) return 128;
    return (M <= 64 || (M % 128 != 0 && M % 128 <= 64)) ? 64 : 128;
}

extern "C" size_t gd_conv3x3_ws_bytes(int M, int Ck) {
    const int bm = pick_bm(M, Ck);
    const long mtiles = (M + bm - 1) / bm, nchunks = (Ck + CK - 1) / CK;
    return (size_t)(mtiles * bm * nchunks * CK * 9 * 2);
}

// returns 1 when the descriptor is served by this kernel, 0 when the caller should use the generic one
extern "C" int gd_conv3x3_eligible(const gd_conv_desc* d) {
    if (!(d && d->ks == 3 && d->pad == 1 && d->out_layout == 0 && !d->out_bf16 && d->precision == GD_PREC_BF16 &&
          d->a_bs == 0 && (d->Mstore == 0 || d->Mstore == d->M) && d->sub_step <= 1))
        return 0;
    if (d->stride == 1) return d->Hi == d->Ho && d->Wi == d->Wo;
    // stride 2: forward gather only (its data gradient is split by output parity on the generic kernel)
    return d->stride == 2 && !d->transposed && (d->Hi - 1) / 2 + 1 == d->Ho && (d->Wi - 1) / 2 + 1 == d->Wo;
}

extern "C" int gd_conv3x3(const gd_conv_desc* dp, void* ws, size_t ws_bytes, void* stream) {
    GD_CHECK_ARG(dp && ws, "gd_conv3x3: null argument");
    GD_CHECK_ARG(gd_conv3x3_eligible(dp), "gd_conv3x3: descriptor not eligible (needs 3x3, pad 1, stride 1 or forward stride 2, bf16, fp32 NCHW out)");
    const gd_conv_desc& d = *dp;
    GD_CHECK_ARG(d.B > 0 && d.B <= 65535 && d.M > 0 && d.Ck > 0 && d.Hi > 0 && d.Wi > 0, "gd_conv3x3: bad sizes");
    GD_CHECK_ARG(d.a && d.x && d.y, "gd_conv3x3: null tensor");
    GD_CHECK_ARG((d.in_scale == nullptr) == (d.in_shift == nullptr), "gd_conv3x3: in_scale/in_shift must come together");
    GD_CHECK_ARG(ws_bytes >= gd_conv3x3_ws_bytes(d.M, d.Ck), "gd_conv3x3: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int bm = pick_bm(d.M, d.Ck);
    const int mtiles = (d.M + bm - 1) / bm, nchunks = (d.Ck + CK - 1) / CK;
    const long total = (long)mtiles * nchunks * 9 * bm * CK;
    {
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(pack_w_kernel, dim3(blocks), dim3(256), 0, s, d.a, d.a_sm, d.a_sc, d.a_st, d.M, d.Ck,
                           d.transposed ? 1 : 0, bm, nchunks, (unsigned short*)ws, total);
    }
    const int th = d.stride == 1 ? 8 : 4;   // stride 2 patches are 4x larger per output row: 4-row tiles keep 2 blocks/CU
    const int tiles_x = (d.Wo + TW - 1) / TW, tiles_y = (d.Ho + th - 1) / th;
    GD_CHECK_ARG((long)tiles_x * tiles_y < (1L << 31), "gd_conv3x3: image too large");
    dim3 grid(tiles_x * tiles_y, mtiles, d.B);
    const unsigned short* wpk = (const unsigned short*)ws;
    if (d.stride == 1) {
        if (bm == 64) hipLaunchKernelGGL((conv3x3_halo_kernel<64, 8, 1>), grid, dim3(256), 0, s, d, wpk, tiles_x, nchunks);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<128, 8, 1>), grid, dim3(256), 0, s, d, wpk, tiles_x, nchunks);
    } else {
        if (bm == 64) hipLaunchKernelGGL((conv3x3_halo_kernel<64, 4, 2>), grid, dim3(256), 0, s, d, wpk, tiles_x, nchunks);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<128, 4, 2>), grid, dim3(256), 0, s, d, wpk, tiles_x, nchunks);
    }
    GD_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================
// 3x3 / stride 1 / pad 1 convolution on NHWC bf16 activations (the frozen VGG19 feature stack of PerceptualLoss,
// losses.py:13-73: forward AND data gradient, which is the same operator with transposed / flipped weights).
// With pixel-major bf16 storage a patch pixel's 32-channel chunk is 64 contiguous bytes: staging is four 16-byte
// copies per pixel (no gather, no convert), prefetched into registers under the previous chunk's MFMAs, and the
// tensor is read and written at 2 bytes per element.  Same tile / fragment scheme as conv3x3_halo_kernel.
// Epilogue: (+ bias) (ReLU) (* [mask > 0]: backward of the ReLU that produced `mask`) (+ res) -> bf16 NHWC.
// =====================================================================================================
namespace {

struct NhwcConvArgs {
    const unsigned short* x;      // (B, H, W, K) bf16
    const unsigned short* wp;     // packed weights [m-tile][chunk][tap][BM][32]
    const float* bias;            // (M) or null
    const unsigned short* mask;   // (B, Ho, Wo, M) bf16 or null
    const unsigned short* res;    // (B, Ho, Wo, M) bf16 or null
    unsigned short* y;            // (B, Ho, Wo, M) bf16
    float* y32;                   // alternative output: fp32 NCHW (B, M, Ho, Wo) with batch stride y32_bs (y unused then)
    long y32_bs;
    int H, W, Ho, Wo, K, M, act, tiles_x, nchunks;   // act: 0 none, 1 ReLU, 2 LeakyReLU(slope)
    float slope;
    int osplit;                   // 1: y holds 3 M channels per pixel, the result split as [hi | lo | hi] (operand mode "x3")
    int epi_lds;                  // 1: 16-bit pixel-major output through the LDS-transposed epilogue (16-byte stores)
};

// S = stride (1: VGG stack; 2: the down-sampling convs of Discriminator1, discriminator.py:60-63 -- the patch of a
// 4 x 32 output tile is then 9 x 65 input pixels and consecutive output pixels read patch pixels two apart)
// EPI = 1: the LDS-transposed 16-bit pixel-major epilogue (its own instantiation, so that the fp32 NCHW / direct form keeps
// its registers)
template <int BM, int S, int EPI = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_nhwc_kernel(const NhwcConvArgs a) {
    constexpr int TH = S == 1 ? 8 : 4;
    constexpr int WAVES_M = 2, WAVES_N = 2;
    constexpr int TM = BM / (32 * WAVES_M);
    constexpr int TN = TH / WAVES_N;
    constexpr int PH = S * (TH - 1) + 3, PWk = S * (TW - 1) + 3, NPIX = PH * PWk;
    constexpr int WCHUNKS = 3 * BM * CK / 8;
    constexpr int WPT = WCHUNKS / 256;
    constexpr int NIT = (4 * NPIX + 255) / 256;       // 16-byte staging items (pixel, channel octet) per thread
    static_assert(WCHUNKS % 256 == 0, "weight stage must divide evenly");

    __shared__ __attribute__((aligned(16))) unsigned short patch[NPIX * LD];
    __shared__ __attribute__((aligned(16))) unsigned short wts[3 * BM * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z, mt = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = a.H, W = a.W, K = a.K;
    const unsigned short* ximg = a.x + (long)b * H * W * K;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const unsigned short* wbase = a.wp + (long)mt * a.nchunks * 9 * BM * CK;
    u32x4_t wreg[WPT];
    auto load_w = [&](int chunk, int ky) {
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(wbase + ((long)chunk * 9 + ky * 3) * BM * CK);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = src[tid + i * 256];
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int v = tid + i * 256;
            const int row = v >> 2, q = v & 3;
            *reinterpret_cast<u32x4_t*>(wts + row * LD + q * 8) = wreg[i];
        }
    };

    // staging plan, fixed per thread: element offset of the item inside the image (channel 0 of the chunk) and its
    // LDS slot; bit `it` of inmask: the pixel lies inside the image (else zero padding)
    unsigned int goff[NIT], inmask = 0, slotmask = 0;
    int loff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int w = tid + it * 256;
        const int pix = w >> 2, q = w & 3;
        const int py = pix / PWk, px = pix - py * PWk;
        const int iy = S * y0 - 1 + py, ix = S * x0 - 1 + px;
        const bool slot = w < 4 * NPIX;
        const bool inside = slot && iy >= 0 && iy < H && ix >= 0 && ix < W;
        loff[it] = pix * LD + q * 8;
        goff[it] = inside ? (unsigned int)(((long)iy * W + ix) * K + q * 8) : 0u;
        inmask |= inside ? (1u << it) : 0u;
        slotmask |= slot ? (1u << it) : 0u;
    }
    u32x4_t raw[NIT];
    auto load_patch = [&](int chunk) {
        const unsigned short* base = ximg + chunk * CK;       // wave-uniform
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = (tid + it * 256) & 3;
            const u32x4_t z = {0u, 0u, 0u, 0u};
            raw[it] = (((inmask >> it) & 1u) && chunk * CK + q * 8 < K) ? *reinterpret_cast<const u32x4_t*>(base + goff[it]) : z;
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if ((slotmask >> it) & 1u) *reinterpret_cast<u32x4_t*>(patch + loff[it]) = raw[it];
    };

    load_patch(0);
    store_patch();
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        load_w(chunk, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            store_w();
            __syncthreads();
            if (ky == 0 && chunk + 1 < a.nchunks) load_patch(chunk + 1);   // lands under this chunk's MFMAs
            if (ky < 2) load_w(chunk, ky + 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int ks = 0; ks < CK / 16; ++ks) {
                    bf16x8_t fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[i] = *reinterpret_cast<const bf16x8_t*>(wts + (kx * BM + wm * TM * 32 + i * 32 + r) * LD +
                                                                  ks * 16 + 8 * h);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[j] = *reinterpret_cast<const bf16x8_t*>(patch + (((wn * TN + j) * S + ky) * PWk + r * S + kx) * LD +
                                                                  ks * 16 + 8 * h);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8_native_t, fa[i]), __builtin_bit_cast(bf16x8_native_t, fb[j]),
                                acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        if (chunk + 1 < a.nchunks) store_patch();       // published by the next chunk's first barrier
    }

    // ---- pixel-major 16-bit output: every 32 x 32 accumulator tile goes through a per-wave LDS scratch (the patch image is
    // dead behind the loop's last barrier) and leaves as 16-byte stores, four lanes covering the 64 contiguous bytes of a
    // pixel's 32 channels -- the direct form below writes 8 bytes per lane 2 M (6 M) bytes apart.  GD_NHWC_EPI_LDS=0 (host)
    // selects the direct form for A/B.
    if constexpr (EPI == 1) {
        constexpr int SLD = LD;                              // 80-byte scratch rows: 16-byte aligned reads
        static_assert(4 * 2 * 32 * SLD <= NPIX * LD, "per-wave scratch must fit in the patch image");
        unsigned short* scr = patch + wave * (2 * 32 * SLD);
        const int RS = a.osplit ? 3 * a.M : a.M;
        const int ox = x0 + r;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int oy = y0 + wn * TN + j;
            if (oy >= a.Ho) continue;                            // wave-uniform
            const long prow = ((long)b * a.Ho + oy) * a.Wo;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mbase = mt * BM + wm * TM * 32 + i * 32;
                if (mbase >= a.M) continue;                      // wave-uniform
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = mbase + 8 * g + 4 * h;
                    const bool ok = ox < a.Wo && m < a.M;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = acc[i][j][4 * g + k];
                        if (a.bias && ok) v[k] += a.bias[m + k];
                        if (a.act == 1) v[k] = fmaxf(v[k], 0.f);
                        else if (a.act == 2) v[k] = v[k] > 0.f ? v[k] : v[k] * a.slope;
                    }
                    if (ok) {
                        const long pbase = (prow + ox) * RS;
                        if (a.mask) {
                            const uint2 mk = *reinterpret_cast<const uint2*>(a.mask + pbase + m);
                            if (!((mk.x & 0x7FFFu) && !(mk.x & 0x8000u))) v[0] = 0.f;
                            if (!((mk.x >> 16) & 0x7FFFu) || (mk.x >> 31)) v[1] = 0.f;
                            if (!((mk.y & 0x7FFFu) && !(mk.y & 0x8000u))) v[2] = 0.f;
                            if (!((mk.y >> 16) & 0x7FFFu) || (mk.y >> 31)) v[3] = 0.f;
                        }
                        if (a.res) {
#pragma unroll
                            for (int part = 0; part < 2; ++part) {
                                if (part && !a.osplit) break;
                                const uint2 rr = *reinterpret_cast<const uint2*>(a.res + pbase + part * a.M + m);
                                v[0] += gd_bf2f((unsigned short)(rr.x & 0xFFFFu));
                                v[1] += gd_bf2f((unsigned short)(rr.x >> 16));
                                v[2] += gd_bf2f((unsigned short)(rr.y & 0xFFFFu));
                                v[3] += gd_bf2f((unsigned short)(rr.y >> 16));
                            }
                        }
                    }
                    uint2 hi, lo = make_uint2(0u, 0u);
                    if (a.osplit) {
                        gd_split_bf2(v[0], v[1], hi.x, lo.x);
                        gd_split_bf2(v[2], v[3], hi.y, lo.y);
                        *reinterpret_cast<uint2*>(scr + 32 * SLD + r * SLD + 8 * g + 4 * h) = lo;
                    } else {
                        hi.x = gd_pack_bf2(v[0], v[1]);
                        hi.y = gd_pack_bf2(v[2], v[3]);
                    }
                    *reinterpret_cast<uint2*>(scr + r * SLD + 8 * g + 4 * h) = hi;
                }
                // the wave's own LDS writes above are ordered before these reads (one wave: in-order DS queue)
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int px = pass * 16 + (lane >> 2), c8 = (lane & 3) * 8;
                    const int oxp = x0 + px, m = mbase + c8;
                    if (oxp < a.Wo && m < a.M) {
                        unsigned short* dst = a.y + (prow + oxp) * RS + m;
                        const u32x4_t vh = *reinterpret_cast<const u32x4_t*>(scr + px * SLD + c8);
                        *reinterpret_cast<u32x4_t*>(dst) = vh;
                        if (a.osplit) {
                            *reinterpret_cast<u32x4_t*>(dst + a.M) = *reinterpret_cast<const u32x4_t*>(scr + 32 * SLD + px * SLD + c8);
                            *reinterpret_cast<u32x4_t*>(dst + 2 * a.M) = vh;
                        }
                    }
                }
            }
        }
        return;
    }

    // ---- epilogue: lane = pixel, accumulator registers 4g..4g+3 = four consecutive output channels (8-byte stores) ----
    const int ox = x0 + r;
    if (ox < a.Wo) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int oy = y0 + wn * TN + j;
            if (oy >= a.Ho) continue;
            const long pbase = (((long)b * a.Ho + oy) * a.Wo + ox) * (a.osplit ? 3 * a.M : a.M);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = mt * BM + wm * TM * 32 + i * 32 + 8 * g + 4 * h;
                    if (m >= a.M) continue;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = acc[i][j][4 * g + k];
                        if (a.bias) v[k] += a.bias[m + k];
                        if (a.act == 1) v[k] = fmaxf(v[k], 0.f);
                        else if (a.act == 2) v[k] = v[k] > 0.f ? v[k] : v[k] * a.slope;
                    }
                    if (a.mask) {
                        const uint2 mk = *reinterpret_cast<const uint2*>(a.mask + pbase + m);
                        // bf16 > 0  <=>  sign clear and magnitude non-zero
                        if (!((mk.x & 0x7FFFu) && !(mk.x & 0x8000u))) v[0] = 0.f;
                        if (!((mk.x >> 16) & 0x7FFFu) || (mk.x >> 31)) v[1] = 0.f;
                        if (!((mk.y & 0x7FFFu) && !(mk.y & 0x8000u))) v[2] = 0.f;
                        if (!((mk.y >> 16) & 0x7FFFu) || (mk.y >> 31)) v[3] = 0.f;
                    }
                    if (a.res) {
#pragma unroll
                        for (int part = 0; part < 2; ++part) {          // split: res = hi + lo
                            if (part && !a.osplit) break;
                            const uint2 rr = *reinterpret_cast<const uint2*>(a.res + pbase + part * a.M + m);
                            v[0] += gd_bf2f((unsigned short)(rr.x & 0xFFFFu));
                            v[1] += gd_bf2f((unsigned short)(rr.x >> 16));
                            v[2] += gd_bf2f((unsigned short)(rr.y & 0xFFFFu));
                            v[3] += gd_bf2f((unsigned short)(rr.y >> 16));
                        }
                    }
                    if (a.y32) {
                        // fp32 NCHW: for one channel the 32 lanes of a half-wave are 32 consecutive pixels (128-byte rows)
                        float* yp = a.y32 + (long)b * a.y32_bs + (long)m * a.Ho * a.Wo + (long)oy * a.Wo + ox;
#pragma unroll
                        for (int k = 0; k < 4; ++k) yp[(long)k * a.Ho * a.Wo] = v[k];
                        continue;
                    }
                    uint2 o;
                    if (a.osplit) {
                        uint2 lo;
                        gd_split_bf2(v[0], v[1], o.x, lo.x);
                        gd_split_bf2(v[2], v[3], o.y, lo.y);
                        *reinterpret_cast<uint2*>(a.y + pbase + a.M + m) = lo;
                        *reinterpret_cast<uint2*>(a.y + pbase + 2 * a.M + m) = o;
                    } else {
                        o.x = gd_pack_bf2(v[0], v[1]);
                        o.y = gd_pack_bf2(v[2], v[3]);
                    }
                    *reinterpret_cast<uint2*>(a.y + pbase + m) = o;
                }
        }
    }
}

// Data gradient of the stride-2 convolution on NHWC bf16, split by the PARITY of the input pixel (Y, X) = (2u+py, 2v+px):
//   dX[Y][X][ci] = sum over taps (ky, kx) with ky = py+1 (mod 2), kx = px+1 (mod 2) and co of
//                  W[co][ci][ky][kx] * dY[(Y+1-ky)/2][(X+1-kx)/2][co]
// so class (0,0) has one tap, (0,1)/(1,0) two and (1,1) four: nine MFMA tap passes in all, none of them multiplying
// the zeros a dilated (stride-1 over an up-sampled dY) formulation would.  A workgroup owns a 4 x 32 tile of (u, v)
// = 8 x 64 input pixels and 64 input channels; it stages the 5 x 33 dY patch per 32-channel chunk of co once and
// keeps the four parity classes in 4 x TN accumulators.  Operator image: wp[m-tile][chunk][tap][64][32] with m = ci,
// c = co, taps NOT flipped (gd_conv3x3_nhwc_pack(transposed = 2)).
// Epilogue: * LeakyReLU'(act) where act (B, H, W, K) is the OUTPUT of the activation that produced this conv's input.
struct NhwcDgrad2Args {
    const unsigned short* dy;     // (B, Ho, Wo, M) bf16
    const unsigned short* wp;
    const unsigned short* act;    // (B, H, W, K) bf16 or null
    unsigned short* dx;           // (B, H, W, K) bf16
    int H, W, Ho, Wo, K, M, tiles_x, nchunks;
    float slope;
    int split;                    // 1: act and dx hold 3 K channels per pixel [hi | lo | hi] (dy / wp carry the split in M)
};

__global__ __launch_bounds__(256, 2) void conv3x3_nhwc_dgrad2_kernel(const NhwcDgrad2Args a) {
    constexpr int BM = 64, TH = 4;
    constexpr int TN = TH / 2;                     // dY rows per wave
    constexpr int PH = TH + 1, PWd = TW + 1, NPIX = PH * PWd;
    constexpr int WPT = 3 * BM * CK / 8 / 256;     // 3
    constexpr int NIT = (4 * NPIX + 255) / 256;    // 3

    __shared__ __attribute__((aligned(16))) unsigned short patch[NPIX * LD];
    __shared__ __attribute__((aligned(16))) unsigned short wts[3 * BM * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z, mt = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int u0 = ty * TH, v0 = tx * TW;
    const int Ho = a.Ho, Wo = a.Wo, M = a.M;
    const unsigned short* dyimg = a.dy + (long)b * Ho * Wo * M;

    f32x16_t acc[4][TN];                           // [parity class 2 py + px][dY row of this wave]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][j][e] = 0.f;

    const unsigned short* wbase = a.wp + (long)mt * a.nchunks * 9 * BM * CK;
    u32x4_t wreg[WPT];
    auto load_w = [&](int chunk, int ky) {
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(wbase + ((long)chunk * 9 + ky * 3) * BM * CK);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = src[tid + i * 256];
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int v = tid + i * 256;
            *reinterpret_cast<u32x4_t*>(wts + (v >> 2) * LD + (v & 3) * 8) = wreg[i];
        }
    };

    unsigned int goff[NIT], inmask = 0, slotmask = 0;
    int loff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int w = tid + it * 256;
        const int pix = w >> 2, q = w & 3;
        const int py = pix / PWd, px = pix - py * PWd;
        const int iy = u0 + py, ix = v0 + px;
        const bool slot = w < 4 * NPIX;
        const bool inside = slot && iy < Ho && ix < Wo;
        loff[it] = pix * LD + q * 8;
        goff[it] = inside ? (unsigned int)(((long)iy * Wo + ix) * M + q * 8) : 0u;
        inmask |= inside ? (1u << it) : 0u;
        slotmask |= slot ? (1u << it) : 0u;
    }
    u32x4_t raw[NIT];
    auto load_patch = [&](int chunk) {
        const unsigned short* base = dyimg + chunk * CK;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = (tid + it * 256) & 3;
            const u32x4_t z = {0u, 0u, 0u, 0u};
            raw[it] = (((inmask >> it) & 1u) && chunk * CK + q * 8 < M) ? *reinterpret_cast<const u32x4_t*>(base + goff[it]) : z;
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if ((slotmask >> it) & 1u) *reinterpret_cast<u32x4_t*>(patch + loff[it]) = raw[it];
    };

    load_patch(0);
    store_patch();
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        load_w(chunk, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            store_w();
            __syncthreads();
            if (ky == 0 && chunk + 1 < a.nchunks) load_patch(chunk + 1);
            if (ky < 2) load_w(chunk, ky + 1);
            // tap row ky serves input rows of parity py = (ky + 1) & 1 from dY row u + oy, oy = (ky == 0)
            const int py = (ky + 1) & 1, oy = ky == 0 ? 1 : 0;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int px = (kx + 1) & 1, ox = kx == 0 ? 1 : 0;
#pragma unroll
                for (int ks = 0; ks < CK / 16; ++ks) {
                    const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(wts + (kx * BM + wm * 32 + r) * LD + ks * 16 + 8 * h);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(
                            patch + ((wn * TN + j + oy) * PWd + r + ox) * LD + ks * 16 + 8 * h);
                        acc[2 * py + px][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_native_t, fa), __builtin_bit_cast(bf16x8_native_t, fb),
                            acc[2 * py + px][j], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        if (chunk + 1 < a.nchunks) store_patch();
    }

    // ---- epilogue: lane = dY column v0 + r -> input pixels X = 2 (v0 + r) + px; 8-byte stores of 4 channels ----
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const int X = 2 * (v0 + r) + px;
        if (X >= a.W) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int Y = 2 * (u0 + wn * TN + j) + py;
            if (Y >= a.H) continue;
            const long pbase = (((long)b * a.H + Y) * a.W + X) * (a.split ? 3 * a.K : a.K);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = mt * BM + wm * 32 + 8 * g + 4 * h;
                if (m >= a.K) continue;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[c][j][4 * g + k];
                if (a.act) {
                    const uint2 mk = *reinterpret_cast<const uint2*>(a.act + pbase + m);
                    if (!((mk.x & 0x7FFFu) && !(mk.x & 0x8000u))) v[0] *= a.slope;
                    if (!((mk.x >> 16) & 0x7FFFu) || (mk.x >> 31)) v[1] *= a.slope;
                    if (!((mk.y & 0x7FFFu) && !(mk.y & 0x8000u))) v[2] *= a.slope;
                    if (!((mk.y >> 16) & 0x7FFFu) || (mk.y >> 31)) v[3] *= a.slope;
                }
                uint2 o;
                if (a.split) {
                    uint2 lo;
                    gd_split_bf2(v[0], v[1], o.x, lo.x);
                    gd_split_bf2(v[2], v[3], o.y, lo.y);
                    *reinterpret_cast<uint2*>(a.dx + pbase + a.K + m) = lo;
                    *reinterpret_cast<uint2*>(a.dx + pbase + 2 * a.K + m) = o;
                } else {
                    o.x = gd_pack_bf2(v[0], v[1]);
                    o.y = gd_pack_bf2(v[2], v[3]);
                }
                *reinterpret_cast<uint2*>(a.dx + pbase + m) = o;
            }
        }
    }
}

}  // namespace

// w (Cout, Cin, 3, 3) fp32 -> packed bf16 operator image in ws (gd_conv3x3_ws_bytes(M, K) bytes with
// (M, K) = (Cout, Cin) for the forward operator (transposed = 0, stride 1 or 2), (Cin, Cout) for transposed = 1 (the
// stride-1 data-gradient operator: taps flipped) and transposed = 2 (the stride-2 data-gradient operator of
// gd_conv3x3_nhwc_s2_dgrad: taps as stored, 64-row tiles)
extern "C" int gd_conv3x3_nhwc_pack(const float* w, int Cout, int Cin, int transposed, void* ws, size_t ws_bytes,
                                    void* stream) {
    GD_CHECK_ARG(w && ws && Cout > 0 && Cin > 0 && transposed >= 0 && transposed <= 2, "gd_conv3x3_nhwc_pack: bad arguments");
    const int M = transposed ? Cin : Cout, K = transposed ? Cout : Cin;
    GD_CHECK_ARG(ws_bytes >= gd_conv3x3_ws_bytes(M, K), "gd_conv3x3_nhwc_pack: workspace too small");
    const int bm = transposed == 2 ? 64 : pick_bm(M, K);
    const int mtiles = (M + bm - 1) / bm, nchunks = (K + CK - 1) / CK;
    const long total = (long)mtiles * nchunks * 9 * bm * CK;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    const long sm = transposed ? 9 : (long)Cin * 9, sc = transposed ? (long)Cin * 9 : 9;
    hipLaunchKernelGGL(pack_w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, sm, sc, 1L, M, K,
                       transposed == 1 ? 1 : 0, bm, nchunks, (unsigned short*)ws, total);
    GD_LAUNCH_CHECK();
    return 0;
}

static int nhwc_conv_launch(const void* x, const void* wpack, const float* bias, const void* mask, const void* res,
                            void* y, int B, int H, int W, int K, int M, int stride, int act, float slope, void* stream,
                            float* y32 = nullptr, long y32_bs = 0, int osplit = 0) {
    NhwcConvArgs a;
    static const int epi_env = getenv("GD_NHWC_EPI_LDS") ? atoi(getenv("GD_NHWC_EPI_LDS")) : 1;
    a.y32 = y32; a.y32_bs = y32_bs; a.osplit = osplit; a.epi_lds = epi_env;
    a.x = (const unsigned short*)x; a.wp = (const unsigned short*)wpack; a.bias = bias;
    a.mask = (const unsigned short*)mask; a.res = (const unsigned short*)res; a.y = (unsigned short*)y;
    a.H = H; a.W = W; a.K = K; a.M = M; a.act = act; a.slope = slope;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    const int bm = pick_bm(M, K);
    const int mtiles = (M + bm - 1) / bm;
    a.nchunks = (K + CK - 1) / CK;
    a.tiles_x = (a.Wo + TW - 1) / TW;
    const int th = stride == 1 ? 8 : 4;
    const int tiles_y = (a.Ho + th - 1) / th;
    dim3 grid(a.tiles_x * tiles_y, mtiles, B);
    hipStream_t s = (hipStream_t)stream;
    if (!a.y32 && a.epi_lds) {
        if (stride == 1) {
            if (bm == 64) hipLaunchKernelGGL((conv3x3_nhwc_kernel<64, 1, 1>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv3x3_nhwc_kernel<128, 1, 1>), grid, dim3(256), 0, s, a);
        } else {
            if (bm == 64) hipLaunchKernelGGL((conv3x3_nhwc_kernel<64, 2, 1>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((conv3x3_nhwc_kernel<128, 2, 1>), grid, dim3(256), 0, s, a);
        }
    } else if (stride == 1) {
        if (bm == 64) hipLaunchKernelGGL((conv3x3_nhwc_kernel<64, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_nhwc_kernel<128, 1>), grid, dim3(256), 0, s, a);
    } else {
        if (bm == 64) hipLaunchKernelGGL((conv3x3_nhwc_kernel<64, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_nhwc_kernel<128, 2>), grid, dim3(256), 0, s, a);
    }
    GD_LAUNCH_CHECK();
    return 0;
}

// split = 1 (operand mode "x3"): x holds K = 3 Cin physical channels [hi | lo | hi], wpack comes from weights split [hi ; hi ; lo]
// along the contraction axis, and y / mask / res hold 3 M channels per pixel (mask: the sign of the hi part; res: hi + lo)
extern "C" int gd_conv3x3_nhwc(const void* x, const void* wpack, const float* bias, const void* mask, const void* res,
                               void* y, int B, int H, int W, int K, int M, int relu, int split, void* stream) {
    GD_CHECK_ARG(x && wpack && y, "gd_conv3x3_nhwc: null pointer");
    GD_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && W > 0 && K > 0 && M > 0 && K % 8 == 0 && M % 8 == 0,
                 "gd_conv3x3_nhwc: channel counts must be multiples of 8");
    GD_CHECK_ARG((long)H * W * K < (1L << 31) && (long)H * W * M * (split ? 3 : 1) < (1L << 31), "gd_conv3x3_nhwc: image too large");
    return nhwc_conv_launch(x, wpack, bias, mask, res, y, B, H, W, K, M, 1, relu ? 1 : 0, 0.f, stream, nullptr, 0, split ? 1 : 0);
}

// stride 1 with an fp32 NCHW result: x (B, H, W, K) bf16 pixel-major -> y32 (B, M, H, W) fp32 (batch stride y_bs elements),
// + bias, act 0 / 1.  For the wide 3x3 convs of the generator (the 2C -> C fuse conv of DANetAttention, generator.py:108):
// the caller packs the fp32 NCHW input pixel-major once (gd_pack_16, shared with gd_conv3x3_wgrad's x_nhwc16) and gets
// the NHWC kernel's 16-byte patch staging; the data gradient is the same call on the packed dY with the transposed operator.
extern "C" int gd_conv3x3_nhwc_f32out(const void* x, const void* wpack, const float* bias, float* y32, long y_bs, int B, int H,
                                      int W, int K, int M, int relu, void* stream) {
    GD_CHECK_ARG(x && wpack && y32, "gd_conv3x3_nhwc_f32out: null pointer");
    GD_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && W > 0 && K > 0 && M > 0 && K % 8 == 0 && M % 4 == 0 && y_bs >= (long)M * H * W,
                 "gd_conv3x3_nhwc_f32out: K must be a multiple of 8, M of 4");
    GD_CHECK_ARG((long)H * W * K < (1L << 31), "gd_conv3x3_nhwc_f32out: image too large");
    return nhwc_conv_launch(x, wpack, bias, nullptr, nullptr, nullptr, B, H, W, K, M, 1, relu ? 1 : 0, 0.f, stream, y32, y_bs);
}

// stride 2 / pad 1 forward: x (B, H, W, K) bf16 -> y (B, (H-1)/2+1, (W-1)/2+1, M) bf16, + bias, act 0 none / 1 ReLU /
// 2 LeakyReLU(slope).  wpack = gd_conv3x3_nhwc_pack(transposed = 0).
// split = 1 (operand mode "x3"): x holds K = 3 Cin channels per pixel [hi | lo | hi], wpack comes from the weights split
// [hi ; hi ; lo] along Cin (gd_split3_weights), and y receives 3 M channels per pixel, the result split the same way.
extern "C" int gd_conv3x3_nhwc_s2(const void* x, const void* wpack, const float* bias, void* y, int B, int H, int W, int K,
                                  int M, int act, float slope, int split, void* stream) {
    GD_CHECK_ARG(x && wpack && y, "gd_conv3x3_nhwc_s2: null pointer");
    GD_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && W > 0 && K > 0 && M > 0 && K % 8 == 0 && M % 8 == 0 && act >= 0 && act <= 2,
                 "gd_conv3x3_nhwc_s2: channel counts must be multiples of 8, act in 0..2");
    GD_CHECK_ARG((long)H * W * K < (1L << 31) && (long)H * W * M * (split ? 3 : 1) < (1L << 31), "gd_conv3x3_nhwc_s2: image too large");
    return nhwc_conv_launch(x, wpack, bias, nullptr, nullptr, y, B, H, W, K, M, 2, act, slope, stream, nullptr, 0, split ? 1 : 0);
}

// data gradient of gd_conv3x3_nhwc_s2: dy (B, Ho, Wo, M) bf16 -> dx (B, H, W, K) bf16, optionally times
// LeakyReLU'(act_out) with act_out (B, H, W, K) the activation output this conv consumed.
// wpack_t = gd_conv3x3_nhwc_pack(transposed = 2).
// split = 1: dy holds M = 3 Cout channels per pixel [hi | lo | hi], wpack_t comes from the weights split [hi ; hi ; lo] along
// Cout, act_out and dx hold 3 K channels per pixel (the sign of a split value is the sign of its hi part).
extern "C" int gd_conv3x3_nhwc_s2_dgrad(const void* dy, const void* wpack_t, const void* act_out, float slope, void* dx,
                                        int B, int H, int W, int K, int M, int split, void* stream) {
    GD_CHECK_ARG(dy && wpack_t && dx, "gd_conv3x3_nhwc_s2_dgrad: null pointer");
    GD_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && W > 0 && K > 0 && M > 0 && K % 8 == 0 && M % 8 == 0,
                 "gd_conv3x3_nhwc_s2_dgrad: channel counts must be multiples of 8");
    GD_CHECK_ARG((long)H * W * K * (split ? 3 : 1) < (1L << 31) && (long)H * W * M < (1L << 31), "gd_conv3x3_nhwc_s2_dgrad: image too large");
    NhwcDgrad2Args a;
    a.split = split ? 1 : 0;
    a.dy = (const unsigned short*)dy; a.wp = (const unsigned short*)wpack_t; a.act = (const unsigned short*)act_out;
    a.dx = (unsigned short*)dx; a.H = H; a.W = W; a.K = K; a.M = M; a.slope = slope;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.nchunks = (M + CK - 1) / CK;
    a.tiles_x = (a.Wo + TW - 1) / TW;
    const int tiles_y = (a.Ho + 3) / 4;
    dim3 grid(a.tiles_x * tiles_y, (K + 63) / 64, B);
    hipLaunchKernelGGL(conv3x3_nhwc_dgrad2_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    GD_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================
// weight gradient of the 3x3 / stride 1 / pad 1 convolution
//   dW[co][ci][tap] = sum_{b,y,x} dY[b][co][y][x] * X~[b][ci][y+ky-1][x+kx-1]
// GEMM view per tap: M = co, N = ci, K = pixels.  A workgroup owns BM = 32*NW output channels x one 32-channel
// chunk of ci x all nine taps (wave w holds the nine 32x32 accumulators of m-tile w) and walks a share of the
// (image, 4x32-pixel tile) list.  Per tile it stages dY as [co][pixel] (pixel-contiguous like NCHW: the A
// fragment is one 16-byte read) and the haloed input patch as [pixel][ci] -- the same image the forward kernel
// uses -- and takes the B fragments (k = pixel, n = ci) with ds_read_b64_tr_b16, the LDS transpose read, so the
// nine taps are nine shifted reads of one staged patch.  Partial sums of the splits are combined with fp32
// atomics into the (small) weight gradient.
// =====================================================================================================
namespace {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
constexpr int WTH = 4;                 // tile rows
constexpr int WPH = WTH + 2;
constexpr int WNPIX = WPH * PW;        // 204 patch pixels
constexpr int DYLD = WTH * TW + 8;     // dY rows: 128 pixels + 8 pad (272 B): conflict-free 16-byte reads

struct WgradArgs {
    const unsigned short* dy16;      // optional bf16 copy of dy, dense (B, M, Ho, Wo): read instead of dy
    const float* dy; long dy_bs;
    const float* x; long x_bs;
    const unsigned short* x16; int x16_ld;   // optional pixel-major bf16 copy of x, dense (B, H, W, x16_ld): read instead of x
    const float* in_scale; const float* in_shift; int in_relu;
    float* dw;
    int B, M, Ck, H, W, Ho, Wo;
    int tiles_x, tiles_y, tiles_per_split;
};

__device__ __forceinline__ s16x4_t lds_tr_read(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

// CS = false: the NW waves are NW output-channel tiles (BM = 32 NW) over ONE 32-channel chunk of ci.
// CS = true (Cout <= 32, the 24-channel dense layers): the waves are NW ci-CHUNKS sharing one 32-row dY tile -- the
//   dY tile is staged once for 32 NW input channels, no wave multiplies an all-zero m-tile, and the workgroup has
//   NW x 64 threads to keep loads in flight (with waves = m-tiles a Cout of 24 left 128 threads per workgroup).
#ifndef GD_WGRAD_PIPED_MINW
#define GD_WGRAD_PIPED_MINW 1
#endif
template <int NW, int S, bool CS, bool PIPED = false>
__global__ __launch_bounds__(NW * 64, PIPED ? GD_WGRAD_PIPED_MINW : 2) void conv3x3_wgrad_kernel(const WgradArgs a) {
    constexpr int BM = CS ? 32 : 32 * NW, NT = 64 * NW, NCH = CS ? NW : 1;
    constexpr int PHk = S * (WTH - 1) + 3, PWk = S * (TW - 1) + 3, NPIXk = PHk * PWk;   // input patch of the tile
    __shared__ __attribute__((aligned(16))) unsigned short dys[BM * DYLD];
    __shared__ __attribute__((aligned(16))) unsigned short patch[NCH * NPIXk * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c0 = blockIdx.x * CK * NCH;
    const int m0 = blockIdx.y * BM;
    const long HW = (long)a.H * a.W;            // input plane
    const long HWo = (long)a.Ho * a.Wo;         // output (dY) plane
    const int ntiles = a.B * a.tiles_y * a.tiles_x;
    const int t_begin = blockIdx.z * a.tiles_per_split;
    const int t_end = min(ntiles, t_begin + a.tiles_per_split);

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // transpose-read lane roles: group of 16 lanes = one 4(k) x 16(n) block; lane 4q+pp supplies row q, cols 4pp..
    const int li = lane & 15, tq = li >> 2, tp = li & 3, tg = (lane >> 4) & 1;
    const bool vec_ok = (a.Wo % 4) == 0;

    // Both operands 16-bit (dy16 channel-major, x16 pixel-major): the tile loop is software-pipelined -- the global loads
    // of tile t + 1 are issued into registers before the MFMAs of tile t and stored to LDS after them, so the memory
    // round trip no longer sits between two barriers with the matrix pipe idle
    // (PIPED instantiations only: the host picks them when both 16-bit copies are given)
    constexpr bool piped = PIPED;
    constexpr int NDY = PIPED ? (BM * 32 + NT - 1) / NT : 1;     // 8-byte dY items per thread
    constexpr int NXI = PIPED ? (4 * NCH * NPIXk + NT - 1) / NT : 1;   // 16-byte patch items per thread
    uint2 rdy[NDY];
    u32x4_t rxp[NXI];
    auto load_regs = [&](int t) {
        const int b = t / (a.tiles_y * a.tiles_x);
        const int rem = t - b * (a.tiles_y * a.tiles_x);
        const int ty_ = rem / a.tiles_x, tx_ = rem - ty_ * a.tiles_x;
        const int y0 = ty_ * WTH, x0 = tx_ * TW;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int idx = tid + k * NT;
            const int m = idx >> 5, v = idx & 31;
            const int yy = y0 + (v >> 3), xx = x0 + (v & 7) * 4;
            uint2 w = make_uint2(0u, 0u);
            if (idx < BM * 32 && m0 + m < a.M && yy < a.Ho) {
                const unsigned short* p = a.dy16 + ((long)b * a.M + m0 + m) * HWo + (long)yy * a.Wo + xx;
                if (vec_ok && xx + 3 < a.Wo) {
                    w = *reinterpret_cast<const uint2*>(p);
                } else {
                    unsigned int e0 = xx + 0 < a.Wo ? p[0] : 0u, e1 = xx + 1 < a.Wo ? p[1] : 0u;
                    unsigned int e2 = xx + 2 < a.Wo ? p[2] : 0u, e3 = xx + 3 < a.Wo ? p[3] : 0u;
                    w.x = e0 | (e1 << 16);
                    w.y = e2 | (e3 << 16);
                }
            }
            rdy[k] = w;
        }
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int w = tid + k * NT;
            const int oc = w / NPIXk;                   // (chunk, octet)
            const int q = oc & 3, chk = oc >> 2;
            const int pix = w - oc * NPIXk;
            const int py = pix / PWk, px = pix - py * PWk;
            const int iy = S * y0 - 1 + py, ix = S * x0 - 1 + px;
            const int cb = c0 + chk * CK + q * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (w < 4 * NCH * NPIXk && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && cb < a.x16_ld)
                v = *reinterpret_cast<const u32x4_t*>(a.x16 + (((long)b * a.H + iy) * a.W + ix) * a.x16_ld + cb);
            rxp[k] = v;
        }
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int idx = tid + k * NT;
            if (idx < BM * 32) *reinterpret_cast<uint2*>(dys + (idx >> 5) * DYLD + (idx & 31) * 4) = rdy[k];
        }
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int w = tid + k * NT;
            if (w < 4 * NCH * NPIXk) {
                const int oc = w / NPIXk;
                const int pix = w - oc * NPIXk;
                *reinterpret_cast<u32x4_t*>(patch + ((oc >> 2) * NPIXk + pix) * LD + (oc & 3) * 8) = rxp[k];
            }
        }
    };
    if constexpr (piped) {
        if (t_begin < t_end) load_regs(t_begin);
    }

    for (int t = t_begin; t < t_end; ++t) {
        const int b = t / (a.tiles_y * a.tiles_x);
        const int rem = t - b * (a.tiles_y * a.tiles_x);
        const int ty_ = rem / a.tiles_x, tx_ = rem - ty_ * a.tiles_x;
        const int y0 = ty_ * WTH, x0 = tx_ * TW;
        const float* dyb = a.dy + (long)b * a.dy_bs;
        const float* xb = a.x + (long)b * a.x_bs;

        if constexpr (piped) {
            store_regs();
        } else {
        // ---- dY tile -> [co][pixel] bf16 : 4-pixel vectors, BM*32 of them ----
#pragma unroll 4
        for (int k = 0; k < 16 / (CS ? NW : 1) + (CS && (16 % NW) ? 1 : 0); ++k) {
            const int idx = tid + k * NT;
            if (CS && idx >= 32 * 32) break;
            const int m = idx >> 5, v = idx & 31;
            const int yy = y0 + (v >> 3), xx = x0 + (v & 7) * 4;
            if (a.dy16) {       // pre-converted dY: 8-byte copies, no convert (the tile is re-read once per ci chunk)
                uint2 w = make_uint2(0u, 0u);
                if (m0 + m < a.M && yy < a.Ho) {
                    const unsigned short* p = a.dy16 + ((long)b * a.M + m0 + m) * HWo + (long)yy * a.Wo + xx;
                    if (vec_ok && xx + 3 < a.Wo) {
                        w = *reinterpret_cast<const uint2*>(p);
                    } else {
                        unsigned int e0 = xx + 0 < a.Wo ? p[0] : 0u, e1 = xx + 1 < a.Wo ? p[1] : 0u;
                        unsigned int e2 = xx + 2 < a.Wo ? p[2] : 0u, e3 = xx + 3 < a.Wo ? p[3] : 0u;
                        w.x = e0 | (e1 << 16);
                        w.y = e2 | (e3 << 16);
                    }
                }
                *reinterpret_cast<uint2*>(dys + m * DYLD + v * 4) = w;
                continue;
            }
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + m < a.M && yy < a.Ho) {
                const float* p = dyb + (long)(m0 + m) * HWo + (long)yy * a.Wo + xx;
                if (vec_ok && xx + 3 < a.Wo) {
                    f = *reinterpret_cast<const float4*>(p);
                } else {
                    if (xx + 0 < a.Wo) f.x = p[0];
                    if (xx + 1 < a.Wo) f.y = p[1];
                    if (xx + 2 < a.Wo) f.z = p[2];
                    if (xx + 3 < a.Wo) f.w = p[3];
                }
            }
            uint2 w;
            w.x = gd_pack_bf2(f.x, f.y);
            w.y = gd_pack_bf2(f.z, f.w);
            *reinterpret_cast<uint2*>(dys + m * DYLD + v * 4) = w;
        }
        // ---- input patch -> [pixel][ci] bf16 ----
        // pixel-major bf16 source (gd_pack_16 transposed output): a patch pixel's 32-channel chunk is 64 contiguous
        // bytes, so an item = (patch pixel, channel octet) is ONE 16-byte load and ONE 16-byte LDS store -- against 16
        // strided 4-byte loads, 8 converts and 8 stores per item of the fp32 NCHW path below
        if (a.x16) {
            for (int w = tid; w < 4 * NCH * NPIXk; w += NT) {
                const int oc = w / NPIXk;                   // (chunk, octet)
                const int q = oc & 3, chk = oc >> 2;
                const int pix = w - oc * NPIXk;
                const int py = pix / PWk, px = pix - py * PWk;
                const int iy = S * y0 - 1 + py, ix = S * x0 - 1 + px;
                const int cb = c0 + chk * CK + q * 8;
                u32x4_t v = {0u, 0u, 0u, 0u};
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && cb < a.x16_ld)
                    v = *reinterpret_cast<const u32x4_t*>(a.x16 + (((long)b * a.H + iy) * a.W + ix) * a.x16_ld + cb);
                *reinterpret_cast<u32x4_t*>(patch + (chk * NPIXk + pix) * LD + q * 8) = v;
            }
        } else
        // fp32 NCHW source (fused BN affine + ReLU): item = (patch pixel, channel half)
        for (int w = tid; w < 2 * NCH * NPIXk; w += NT) {
            const int hc = w / NPIXk;                   // (chunk, channel half)
            const int half = hc & 1, chk = hc >> 1;
            const int pix = w - hc * NPIXk;
            const int py = pix / PWk, px = pix - py * PWk;
            const int iy = S * y0 - 1 + py, ix = S * x0 - 1 + px;
            const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const int cb = c0 + chk * CK + half * 16;
            const float* p = xb + (long)cb * HW + (long)iy * a.W + ix;
            unsigned int* dst = reinterpret_cast<unsigned int*>(patch + (chk * NPIXk + pix) * LD + half * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cb + 2 * j;
                float v0 = 0.f, v1 = 0.f;
                if (inside) {
                    if (c < a.Ck) v0 = p[(long)(2 * j) * HW];
                    if (c + 1 < a.Ck) v1 = p[(long)(2 * j + 1) * HW];
                    if (a.in_scale) {
                        if (c < a.Ck) {
                            v0 = fmaf(v0, a.in_scale[c], a.in_shift[c]);
                            if (a.in_relu) v0 = fmaxf(v0, 0.f);
                        }
                        if (c + 1 < a.Ck) {
                            v1 = fmaf(v1, a.in_scale[c + 1], a.in_shift[c + 1]);
                            if (a.in_relu) v1 = fmaxf(v1, 0.f);
                        }
                    }
                }
                dst[j] = gd_pack_bf2(v0, v1);
            }
        }
        }
        __syncthreads();
        if constexpr (piped) {
            if (t + 1 < t_end) load_regs(t + 1);
        }

        // ---- 8 k-steps of 16 pixels x 9 taps ----
#pragma unroll 2
        for (int s = 0; s < 8; ++s) {
            const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(dys + ((CS ? 0 : wave * 32) + r) * DYLD + 16 * s + 8 * h);
            const int prow = s >> 1, pcol = (s & 1) * 16 + 8 * h + tq;   // tile-local OUTPUT pixel of this lane's block row
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const unsigned short* p0 = patch + ((CS ? wave * NPIXk : 0) + (prow * S + ky) * PWk + pcol * S + kx) * LD +
                                               16 * tg + 4 * tp;
                    const s16x4_t lo = lds_tr_read(p0);                // output pixels +0..3
                    const s16x4_t hi = lds_tr_read(p0 + 4 * S * LD);   // output pixels +4..7
                    const bf16x8_t fb = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8_native_t, fa), __builtin_bit_cast(bf16x8_native_t, fb),
                        acc[ky * 3 + kx], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---- combine: dW[co][ci][tap] += acc ----
    const int ci = c0 + (CS ? wave * CK : 0) + r;
    if (ci < a.Ck) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = m0 + (CS ? 0 : wave * 32) + acc_row(e, h);
                if (co < a.M) atomicAdd(a.dw + ((long)co * a.Ck + ci) * 9 + t, acc[t][e]);
            }
    }
}

}  // namespace

static const bool g_wgrad_cs_piped = getenv("GD_WGRAD_CS_PIPED") ? atoi(getenv("GD_WGRAD_CS_PIPED")) != 0 : true;   // A/B switch
static const bool g_wgrad_s2_piped = getenv("GD_WGRAD_S2_PIPED") ? atoi(getenv("GD_WGRAD_S2_PIPED")) != 0 : true;   // A/B switch

// dw (Cout, Cin, 3, 3) fp32 is overwritten (accumulate = 0) or added to.  Same input-transform contract as gd_conv2d.
extern "C" int gd_conv3x3_wgrad(const float* dy, long dy_bs, const void* dy_bf16, const float* x, long x_bs,
                                const void* x_nhwc16, int x_ld, const float* in_scale, const float* in_shift, int in_relu,
                                int B, int Cout, int Cin, int H, int W, int stride, int accumulate, float* dw, void* stream) {
    GD_CHECK_ARG((dy || dy_bf16) && (x || x_nhwc16) && dw, "gd_conv3x3_wgrad: null pointer");
    GD_CHECK_ARG(!x_nhwc16 || (!in_scale && x_ld >= Cin && x_ld % 8 == 0),
                 "gd_conv3x3_wgrad: the pixel-major bf16 x needs x_ld >= Cin, x_ld % 8 == 0 and no input transform");
    GD_CHECK_ARG(B > 0 && Cout > 0 && Cin > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "gd_conv3x3_wgrad: bad sizes");
    GD_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "gd_conv3x3_wgrad: in_scale/in_shift must come together");
    hipStream_t s = (hipStream_t)stream;
    // the pixel splits add into dw with fp32 atomics: it starts from zero, or (accumulate) from what it holds
    if (!accumulate)
        GD_CHECK_ARG(hipMemsetAsync(dw, 0, (size_t)Cout * Cin * 9 * sizeof(float), s) == hipSuccess, "gd_conv3x3_wgrad: memset failed");
    WgradArgs a;
    a.dy16 = (const unsigned short*)dy_bf16;
    a.dy = dy; a.dy_bs = dy_bs; a.x = x; a.x_bs = x_bs;
    a.x16 = (const unsigned short*)x_nhwc16; a.x16_ld = x_ld;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
    a.dw = dw; a.B = B; a.M = Cout; a.Ck = Cin; a.H = H; a.W = W;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;     // 3x3, pad 1
    a.tiles_x = (a.Wo + TW - 1) / TW;
    a.tiles_y = (a.Ho + WTH - 1) / WTH;
    const long ntiles = (long)B * a.tiles_x * a.tiles_y;
    GD_CHECK_ARG(ntiles < (1L << 31), "gd_conv3x3_wgrad: too many tiles");
    // m-tiles per workgroup: 2, 4 or 6 waves, whichever pads Cout least (ties -> the larger block).  A 1-wave
    // variant for Cout <= 32 was measured 2x SLOWER (64 threads stage the whole patch) and is not offered.
    int best_nw = 2;
    long best_pad = -1;
    for (int nw : {2, 4, 6}) {
        const long bm = 32L * nw, pad = ((Cout + bm - 1) / bm) * bm;
        if (best_pad < 0 || pad <= best_pad) { best_pad = pad; best_nw = nw; }
    }
    const int chunks = (Cin + CK - 1) / CK;
    // Cout <= 32 at stride 1: waves = ci chunks (3..4 per workgroup, spread evenly over the groups).  With only two
    // chunks it is slower than two independent 2-wave workgroups (9.2 vs 5 ms on the 64 -> 1 conv at 1024 x 1024).
    const bool cs = Cout <= 32 && stride == 1 && chunks >= 3;
    const int groups = cs ? (chunks + 3) / 4 : chunks;
    const int cs_nw = cs ? (chunks + groups - 1) / groups : 0;
    const int bm = cs ? 32 : 32 * best_nw;
    const int mblocks = (Cout + bm - 1) / bm;
    long splits = 1024 / ((long)mblocks * groups);
    if (splits < 1 || gd_get_deterministic()) splits = 1;    // deterministic mode: one adder per dW element
    if (splits > ntiles) splits = ntiles;
    if (splits > 65535) splits = 65535;
    a.tiles_per_split = (int)((ntiles + splits - 1) / splits);
    splits = (ntiles + a.tiles_per_split - 1) / a.tiles_per_split;
    dim3 grid(groups, mblocks, (unsigned)splits);
    if (cs && a.dy16 && a.x16 && g_wgrad_cs_piped) {
        // both operands 16-bit (the dense layers' packs): the same software pipeline as the wide convs below
        if (cs_nw == 3) hipLaunchKernelGGL((conv3x3_wgrad_kernel<3, 1, true, true>), grid, dim3(192), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 1, true, true>), grid, dim3(256), 0, s, a);
    } else if (cs) {
        if (cs_nw == 3) hipLaunchKernelGGL((conv3x3_wgrad_kernel<3, 1, true>), grid, dim3(192), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 1, true>), grid, dim3(256), 0, s, a);
    } else if (stride == 1 && a.dy16 && a.x16 && best_nw >= 4) {
        // both operands 16-bit: software-pipelined staging (368->184: 8.4 -> 6.4 ms; the 2-wave block spills under it
        // and stays on the plain loop: 1.85 vs 2.20 ms at 184->64)
        if (best_nw == 4) hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 1, false, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wgrad_kernel<6, 1, false, true>), grid, dim3(384), 0, s, a);
    } else if (stride == 1) {
        if (best_nw == 2) hipLaunchKernelGGL((conv3x3_wgrad_kernel<2, 1, false>), grid, dim3(128), 0, s, a);
        else if (best_nw == 4) hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 1, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wgrad_kernel<6, 1, false>), grid, dim3(384), 0, s, a);
    } else if (a.dy16 && a.x16 && best_nw == 4 && g_wgrad_s2_piped) {
        // stride 2, both operands 16-bit (Discriminator1's trunk): the same software pipeline
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 2, false, true>), grid, dim3(256), 0, s, a);
    } else {
        if (best_nw == 2) hipLaunchKernelGGL((conv3x3_wgrad_kernel<2, 2, false>), grid, dim3(128), 0, s, a);
        else if (best_nw == 4) hipLaunchKernelGGL((conv3x3_wgrad_kernel<4, 2, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wgrad_kernel<6, 2, false>), grid, dim3(384), 0, s, a);
    }
    GD_LAUNCH_CHECK();
    return 0;
}
