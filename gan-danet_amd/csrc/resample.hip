// Resampling kernels for gfx950: bicubic (A = -0.75, half-pixel centres, border-clamped taps) and
// bilinear (align_corners = False) resize forward + gather-form backward, 2x2 max-pool.
// HBM-bound; one thread per output element, planes walked contiguously along x (coalesced).
// The backward kernels are GATHERS (each input pixel sums the output pixels that read it, found by
// re-running the forward index computation over a conservative window) -- no atomics, deterministic.
#include "common.h"
#include "../../include/gandanet.h"

namespace {

constexpr int WMAX = 14;  // widest per-axis window the gather backward supports

__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// forward taps of one output coordinate o: indices idx[4] (clamped) and weights w[4]
__device__ __forceinline__ void cubic_taps(int o, float rs, int n_in, int (&idx)[4], float (&w)[4]) {
    const float src = rs * ((float)o + 0.5f) - 0.5f;
    const float fl = floorf(src);
    cubic_w(src - fl, w);
    const int i0 = (int)fl;
#pragma unroll
    for (int t = 0; t < 4; ++t) idx[t] = clampi(i0 - 1 + t, 0, n_in - 1);
}

__device__ __forceinline__ void linear_taps(int o, float rs, int n_in, int& i0, int& i1, float& lam) {
    float src = rs * ((float)o + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
    lam = src - (float)i0;
}

__global__ __launch_bounds__(256) void bicubic_fwd_kernel(const float* __restrict__ x, int Hi, int Wi,
                                                         float* __restrict__ y, int Ho, int Wo, float rsh, float rsw) {
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int oy = blockIdx.y;
    const long bc = blockIdx.z;
    if (ox >= Wo) return;
    int iy[4], ix[4];
    float wy[4], wx[4];
    cubic_taps(oy, rsh, Hi, iy, wy);
    cubic_taps(ox, rsw, Wi, ix, wx);
    const float* p = x + bc * (long)Hi * Wi;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float* row = p + (long)iy[a] * Wi;
        float r = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) r = fmaf(row[ix[c]], wx[c], r);
        acc = fmaf(r, wy[a], acc);
    }
    y[bc * (long)Ho * Wo + (long)oy * Wo + ox] = acc;
}

// Input preamble of the train step (GAN_DANet_train.ipynb:L218-224): out = cat([bicubic(lr, 1/s1), bicubic(aux, 1/s2)], 1)
// in ONE launch, written straight into the (B, C1+C2, Ho, Wo) generator input (no separate cat pass).
// rs1 / rs2 = input/output coordinate ratios (2 and 4 for scale_factor 0.5 / 0.25), antialias off like the reference.
__global__ __launch_bounds__(256) void combine_inputs_kernel(const float* __restrict__ lr, int C1, int H1, int W1, float rs1,
                                                            const float* __restrict__ aux, int C2, int H2, int W2,
                                                            float rs2, float* __restrict__ out, int Ho, int Wo) {
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int oy = blockIdx.y;
    const int z = blockIdx.z, Ct = C1 + C2;
    const int b = z / Ct, c = z - b * Ct;
    if (ox >= Wo) return;
    const bool first = c < C1;
    const int Hi = first ? H1 : H2, Wi = first ? W1 : W2;
    const float rs = first ? rs1 : rs2;
    const float* p = first ? lr + ((long)b * C1 + c) * H1 * W1 : aux + ((long)b * C2 + (c - C1)) * H2 * W2;
    int iy[4], ix[4];
    float wy[4], wx[4];
    cubic_taps(oy, rs, Hi, iy, wy);
    cubic_taps(ox, rs, Wi, ix, wx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float* row = p + (long)iy[a] * Wi;
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) r = fmaf(row[ix[k]], wx[k], r);
        acc = fmaf(r, wy[a], acc);
    }
    out[((long)z * Ho + oy) * Wo + ox] = acc;
}

// Tail of the generator (generator.py:242-247): final = conv3x3(64 -> 1, pad 1) applied to up2(c) + resize4(s) is linear
// in c and s and the resizes act per channel, so  final(z)[p] = b + sum_tap u_tap[p + off_tap]  with the nine "tap planes"
// u_tap = up2(sum_ch w[ch][tap] c_ch) + resize4(sum_ch w[ch][tap] s_ch): the channel contraction runs at LOW resolution and
// only 9 planes (not 64 channels) exist at 4H x 4W.  This kernel is the last step: shifted sum of the planes (+ bias).
__global__ __launch_bounds__(256) void shift_sum9_fwd_kernel(const float* __restrict__ u, const float* __restrict__ bias,
                                                            float* __restrict__ y, int H, int W) {
    const int x = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
    const long b = blockIdx.z;
    if (x >= W) return;
    const long HW = (long)H * W;
    const float* ub = u + b * 9 * HW;
    float acc = bias ? bias[0] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int sy = yy + t / 3 - 1, sx = x + t % 3 - 1;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) acc += ub[t * HW + (long)sy * W + sx];
    }
    y[b * HW + (long)yy * W + x] = acc;
}
// du[b][tap][q] = dy[b][q - off_tap] (zero outside the image)
__global__ __launch_bounds__(256) void shift_sum9_bwd_kernel(const float* __restrict__ dy, float* __restrict__ du, int H, int W) {
    const int x = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
    const long b = blockIdx.z;
    if (x >= W) return;
    const long HW = (long)H * W;
    const float* g = dy + b * HW;
    float* ub = du + b * 9 * HW;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int sy = yy - (t / 3 - 1), sx = x - (t % 3 - 1);
        ub[t * HW + (long)yy * W + x] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? g[(long)sy * W + sx] : 0.f;
    }
}

// conservative window of output coordinates that may read input coordinate i (cubic: |src - i| < 2 + clamp)
__device__ __forceinline__ void cubic_window(int i, float rs, int n_in, int n_out, int& lo, int& hi) {
    const float inv = 1.f / rs;
    lo = (i == 0) ? 0 : (int)floorf(((float)i - 2.f + 0.5f) * inv - 0.5f) - 1;
    hi = (i == n_in - 1) ? n_out - 1 : (int)ceilf(((float)i + 2.f + 0.5f) * inv - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}
__device__ __forceinline__ void linear_window(int i, float rs, int n_in, int n_out, int& lo, int& hi) {
    const float inv = 1.f / rs;
    lo = (i == 0) ? 0 : (int)floorf(((float)i - 1.f + 0.5f) * inv - 0.5f) - 1;
    hi = (i == n_in - 1) ? n_out - 1 : (int)ceilf(((float)i + 1.f + 0.5f) * inv - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}

template <bool CUBIC>
__global__ __launch_bounds__(256) void resize_bwd_gather_kernel(const float* __restrict__ dy, int Hi, int Wi,
                                                               float* __restrict__ dx, int Ho, int Wo, float rsh,
                                                               float rsw) {
    // (Hi, Wi) = forward INPUT size = size of dx; (Ho, Wo) = forward output size = size of dy
    const int ix = blockIdx.x * 256 + threadIdx.x;
    const int iy = blockIdx.y;
    const long bc = blockIdx.z;
    if (ix >= Wi) return;
    int xlo, xhi, ylo, yhi;
    if (CUBIC) {
        cubic_window(ix, rsw, Wi, Wo, xlo, xhi);
        cubic_window(iy, rsh, Hi, Ho, ylo, yhi);
    } else {
        linear_window(ix, rsw, Wi, Wo, xlo, xhi);
        linear_window(iy, rsh, Hi, Ho, ylo, yhi);
    }
    // x weights of every candidate column, once
    float wxs[WMAX];
#pragma unroll
    for (int k = 0; k < WMAX; ++k) {
        const int ox = xlo + k;
        float w = 0.f;
        if (ox <= xhi) {
            if (CUBIC) {
                int idx[4];
                float ww[4];
                cubic_taps(ox, rsw, Wi, idx, ww);
#pragma unroll
                for (int t = 0; t < 4; ++t) w += (idx[t] == ix) ? ww[t] : 0.f;
            } else {
                int i0, i1;
                float lam;
                linear_taps(ox, rsw, Wi, i0, i1, lam);
                w += (i0 == ix) ? (1.f - lam) : 0.f;
                w += (i1 == ix) ? lam : 0.f;
            }
        }
        wxs[k] = w;
    }
    const float* g = dy + bc * (long)Ho * Wo;
    float acc = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
        float wy = 0.f;
        if (CUBIC) {
            int idx[4];
            float ww[4];
            cubic_taps(oy, rsh, Hi, idx, ww);
#pragma unroll
            for (int t = 0; t < 4; ++t) wy += (idx[t] == iy) ? ww[t] : 0.f;
        } else {
            int i0, i1;
            float lam;
            linear_taps(oy, rsh, Hi, i0, i1, lam);
            wy += (i0 == iy) ? (1.f - lam) : 0.f;
            wy += (i1 == iy) ? lam : 0.f;
        }
        if (wy == 0.f) continue;
        const float* row = g + (long)oy * Wo;
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < WMAX; ++k) {
            const int ox = xlo + k;
            if (ox <= xhi) r = fmaf(row[ox], wxs[k], r);
        }
        acc = fmaf(r, wy, acc);
    }
    dx[bc * (long)Hi * Wi + (long)iy * Wi + ix] = acc;
}

// Backward of the exact x2 bicubic upsample (the only bicubic on the training path, generator.py:221,225).
// For scale 2 the fractional offsets are only 0.25 / 0.75, so an interior input pixel i receives from the eight
// output rows 2i-3 .. 2i+4 with eight FIXED weights (even rows 2(i+2-k) carry cubic_w(0.75)[k], odd rows
// 2(i+1-k)+1 carry cubic_w(0.25)[k]).  Border pixels (index 0 or n-1, where clamped taps pile up) take the
// general window routine.
__device__ __forceinline__ void up2_weights(int i, int n_in, int n_out, float rs, int& lo, float (&w)[WMAX]) {
    if (i >= 1 && i <= n_in - 2) {
        float w75[4], w25[4];
        cubic_w(0.75f, w75);
        cubic_w(0.25f, w25);
        lo = 2 * i - 3;
        // offsets 0..7 <-> rows 2i-3 .. 2i+4 : odd rows (v_k, k = 3..0) interleaved with even rows (w_k, k = 3..0)
        const float t[8] = {w25[3], w75[3], w25[2], w75[2], w25[1], w75[1], w25[0], w75[0]};
#pragma unroll
        for (int k = 0; k < WMAX; ++k) {
            const int o = lo + k;
            w[k] = (k < 8 && o >= 0 && o < n_out) ? t[k < 8 ? k : 0] : 0.f;
        }
    } else {
        int hi;
        cubic_window(i, rs, n_in, n_out, lo, hi);
#pragma unroll
        for (int k = 0; k < WMAX; ++k) {
            const int o = lo + k;
            float ww = 0.f;
            if (o <= hi) {
                int idx[4];
                float cw[4];
                cubic_taps(o, rs, n_in, idx, cw);
#pragma unroll
                for (int q = 0; q < 4; ++q) ww += (idx[q] == i) ? cw[q] : 0.f;
            }
            w[k] = ww;
        }
    }
}

__global__ __launch_bounds__(256) void bicubic_up2_bwd_kernel(const float* __restrict__ dy, int Hi, int Wi,
                                                             float* __restrict__ dx) {
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int ix = blockIdx.x * 256 + threadIdx.x;
    const int iy = blockIdx.y;
    const long bc = blockIdx.z;
    if (ix >= Wi) return;
    const float* g = dy + bc * (long)Ho * Wo;
    float acc = 0.f;
    if (iy >= 2 && iy <= Hi - 3 && ix >= 2 && ix <= Wi - 3) {
        // interior: all 8 x 8 contributing outputs exist; fixed weights, fully unrolled, two float4 loads per row
        float w75[4], w25[4];
        cubic_w(0.75f, w75);
        cubic_w(0.25f, w25);
        const float t[8] = {w25[3], w75[3], w25[2], w75[2], w25[1], w75[1], w25[0], w75[0]};
        const float* p = g + (long)(2 * iy - 3) * Wo + (2 * ix - 3);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const float* row = p + (long)a * Wo;
            float rsum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) rsum = fmaf(row[k], t[k], rsum);
            acc = fmaf(rsum, t[a], acc);
        }
    } else {
        int xlo, ylo;
        float wx[WMAX], wy[WMAX];
        up2_weights(ix, Wi, Wo, 0.5f, xlo, wx);
        up2_weights(iy, Hi, Ho, 0.5f, ylo, wy);
#pragma unroll
        for (int a = 0; a < WMAX; ++a) {
            const int oy = ylo + a;
            if (wy[a] == 0.f || oy < 0 || oy >= Ho) continue;
            const float* row = g + (long)oy * Wo;
            float rsum = 0.f;
#pragma unroll
            for (int k = 0; k < WMAX; ++k) {
                const int ox = xlo + k;
                if (wx[k] != 0.f) rsum = fmaf(row[ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox)], wx[k], rsum);
            }
            acc = fmaf(rsum, wy[a], acc);
        }
    }
    dx[bc * (long)Hi * Wi + (long)iy * Wi + ix] = acc;
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int Hi, int Wi,
                                                          float* __restrict__ y, int Ho, int Wo, float rsh, float rsw,
                                                          int accumulate, const float* __restrict__ res) {
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int oy = blockIdx.y;
    const long bc = blockIdx.z;
    if (ox >= Wo) return;
    int y0, y1, x0, x1;
    float ly, lx;
    linear_taps(oy, rsh, Hi, y0, y1, ly);
    linear_taps(ox, rsw, Wi, x0, x1, lx);
    const float* p = x + bc * (long)Hi * Wi;
    const float* r0 = p + (long)y0 * Wi;
    const float* r1 = p + (long)y1 * Wi;
    const float top = r0[x0] * (1.f - lx) + r0[x1] * lx;
    const float bot = r1[x0] * (1.f - lx) + r1[x1] * lx;
    float v = top * (1.f - ly) + bot * ly;
    const long oidx = bc * (long)Ho * Wo + (long)oy * Wo + ox;
    if (res) v += res[oidx];          // y = res + bilinear(x) in one pass (skip addition, generator.py:245)
    else if (accumulate) v += y[oidx];
    y[oidx] = v;
}

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, int Hi, int Wi,
                                                          float* __restrict__ y) {
    const int Ho = Hi / 2, Wo = Wi / 2;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int oy = blockIdx.y;
    const long bc = blockIdx.z;
    if (ox >= Wo) return;
    const float* p = x + bc * (long)Hi * Wi + (long)(2 * oy) * Wi + 2 * ox;
    const float2 a = *reinterpret_cast<const float2*>(p);
    const float2 b = *reinterpret_cast<const float2*>(p + Wi);
    y[bc * (long)Ho * Wo + (long)oy * Wo + ox] = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          int Hi, int Wi, float* __restrict__ dx) {
    const int Ho = Hi / 2, Wo = Wi / 2;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const int oy = blockIdx.y;
    const long bc = blockIdx.z;
    if (ox >= Wo) return;
    const long base = bc * (long)Hi * Wi + (long)(2 * oy) * Wi + 2 * ox;
    const float2 a = *reinterpret_cast<const float2*>(x + base);
    const float2 b = *reinterpret_cast<const float2*>(x + base + Wi);
    const float g = dy[bc * (long)Ho * Wo + (long)oy * Wo + ox];
    // first maximum in row-major order takes the gradient (ATen's max_pool2d_with_indices tie rule)
    int arg = 0;
    float m = a.x;
    if (a.y > m) { m = a.y; arg = 1; }
    if (b.x > m) { m = b.x; arg = 2; }
    if (b.y > m) { m = b.y; arg = 3; }
    float2 oa, ob;
    oa.x = arg == 0 ? g : 0.f;
    oa.y = arg == 1 ? g : 0.f;
    ob.x = arg == 2 ? g : 0.f;
    ob.y = arg == 3 ? g : 0.f;
    *reinterpret_cast<float2*>(dx + base) = oa;
    *reinterpret_cast<float2*>(dx + base + Wi) = ob;
}

int check_plane(int BC, int Hi, int Wi, int Ho, int Wo, const char* who) {
    if (!(BC > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Ho <= 65535 && Hi <= 65535 && BC <= 65535 * 32)) {
        gd_set_error(who);
        return -1;
    }
    return 0;
}

}  // namespace

// blockIdx.z is limited to 65535: split BC over several launches when needed
#define GD_FOR_BC_SLICES(BC, body)                        \
    for (long z0_ = 0; z0_ < (BC); z0_ += 65535) {        \
        const int nz_ = (int)(((BC) - z0_) < 65535 ? ((BC) - z0_) : 65535); \
        body                                              \
    }

extern "C" int gd_bicubic_fwd(const float* x, int BC, int Hi, int Wi, float* y, int Ho, int Wo, float rsh, float rsw,
                              void* stream) {
    GD_CHECK_ARG(x && y, "gd_bicubic_fwd: null pointer");
    if (check_plane(BC, Hi, Wi, Ho, Wo, "gd_bicubic_fwd: bad sizes")) return -1;
    GD_FOR_BC_SLICES(BC, hipLaunchKernelGGL(bicubic_fwd_kernel, dim3(gd_cdiv(Wo, 256), Ho, nz_), dim3(256), 0,
                                            (hipStream_t)stream, x + z0_ * (long)Hi * Wi, Hi, Wi,
                                            y + z0_ * (long)Ho * Wo, Ho, Wo, rsh, rsw);)
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_shift_sum9_fwd(const float* u, const float* bias, float* y, int B, int H, int W, void* stream) {
    GD_CHECK_ARG(u && y && B > 0 && B <= 65535 && H > 0 && H <= 65535 && W > 0, "gd_shift_sum9_fwd: bad arguments");
    hipLaunchKernelGGL(shift_sum9_fwd_kernel, dim3(gd_cdiv(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, u, bias, y, H, W);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_shift_sum9_bwd(const float* dy, float* du, int B, int H, int W, void* stream) {
    GD_CHECK_ARG(dy && du && B > 0 && B <= 65535 && H > 0 && H <= 65535 && W > 0, "gd_shift_sum9_bwd: bad arguments");
    hipLaunchKernelGGL(shift_sum9_bwd_kernel, dim3(gd_cdiv(W, 256), H, B), dim3(256), 0, (hipStream_t)stream, dy, du, H, W);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_combine_inputs(const float* lr, int C1, int H1, int W1, float rs1, const float* aux, int C2, int H2,
                                 int W2, float rs2, float* out, int B, int Ho, int Wo, void* stream) {
    GD_CHECK_ARG(lr && aux && out, "gd_combine_inputs: null pointer");
    GD_CHECK_ARG(B > 0 && C1 > 0 && C2 > 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && Ho > 0 && Wo > 0 && Ho <= 65535 &&
                     (long)B * (C1 + C2) <= 65535 && rs1 > 0.f && rs2 > 0.f,
                 "gd_combine_inputs: bad sizes");
    hipLaunchKernelGGL(combine_inputs_kernel, dim3(gd_cdiv(Wo, 256), Ho, B * (C1 + C2)), dim3(256), 0, (hipStream_t)stream,
                       lr, C1, H1, W1, rs1, aux, C2, H2, W2, rs2, out, Ho, Wo);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_bicubic_bwd(const float* dy, int BC, int Hi, int Wi, float* dx, int Ho, int Wo, float rsh, float rsw,
                              void* stream) {
    GD_CHECK_ARG(dy && dx, "gd_bicubic_bwd: null pointer");
    if (check_plane(BC, Hi, Wi, Ho, Wo, "gd_bicubic_bwd: bad sizes")) return -1;
    if (Ho == 2 * Hi && Wo == 2 * Wi && rsh == 0.5f && rsw == 0.5f && Hi >= 2 && Wi >= 2) {   // exact x2: fixed weights
        GD_FOR_BC_SLICES(BC, hipLaunchKernelGGL(bicubic_up2_bwd_kernel, dim3(gd_cdiv(Wi, 256), Hi, nz_), dim3(256), 0,
                                                (hipStream_t)stream, dy + z0_ * (long)Ho * Wo, Hi, Wi,
                                                dx + z0_ * (long)Hi * Wi);)
        GD_LAUNCH_CHECK();
        return 0;
    }
    GD_CHECK_ARG(4.f / rsw + 5.f <= (float)WMAX, "gd_bicubic_bwd: scale factor too large for the gather window");
    GD_FOR_BC_SLICES(BC, hipLaunchKernelGGL((resize_bwd_gather_kernel<true>), dim3(gd_cdiv(Wi, 256), Hi, nz_), dim3(256),
                                            0, (hipStream_t)stream, dy + z0_ * (long)Ho * Wo, Hi, Wi,
                                            dx + z0_ * (long)Hi * Wi, Ho, Wo, rsh, rsw);)
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_bilinear_fwd(const float* x, int BC, int Hi, int Wi, float* y, int Ho, int Wo, int accumulate,
                               const float* res, void* stream) {
    GD_CHECK_ARG(x && y, "gd_bilinear_fwd: null pointer");
    if (check_plane(BC, Hi, Wi, Ho, Wo, "gd_bilinear_fwd: bad sizes")) return -1;
    const float rsh = (float)Hi / (float)Ho, rsw = (float)Wi / (float)Wo;
    GD_FOR_BC_SLICES(BC, hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(gd_cdiv(Wo, 256), Ho, nz_), dim3(256), 0,
                                            (hipStream_t)stream, x + z0_ * (long)Hi * Wi, Hi, Wi,
                                            y + z0_ * (long)Ho * Wo, Ho, Wo, rsh, rsw, accumulate,
                                            res ? res + z0_ * (long)Ho * Wo : (const float*)nullptr);)
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_bilinear_bwd(const float* dy, int BC, int Hi, int Wi, float* dx, int Ho, int Wo, void* stream) {
    GD_CHECK_ARG(dy && dx, "gd_bilinear_bwd: null pointer");
    if (check_plane(BC, Hi, Wi, Ho, Wo, "gd_bilinear_bwd: bad sizes")) return -1;
    const float rsh = (float)Hi / (float)Ho, rsw = (float)Wi / (float)Wo;
    GD_CHECK_ARG(2.f / rsw + 5.f <= (float)WMAX, "gd_bilinear_bwd: scale factor too large for the gather window");
    GD_FOR_BC_SLICES(BC, hipLaunchKernelGGL((resize_bwd_gather_kernel<false>), dim3(gd_cdiv(Wi, 256), Hi, nz_), dim3(256),
                                            0, (hipStream_t)stream, dy + z0_ * (long)Ho * Wo, Hi, Wi,
                                            dx + z0_ * (long)Hi * Wi, Ho, Wo, rsh, rsw);)
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_maxpool2_fwd(const float* x, int BC, int Hi, int Wi, float* y, void* stream) {
    GD_CHECK_ARG(x && y, "gd_maxpool2_fwd: null pointer");
    GD_CHECK_ARG(Hi % 2 == 0 && Wi % 2 == 0, "gd_maxpool2_fwd: odd sizes unsupported");
    if (check_plane(BC, Hi, Wi, Hi / 2, Wi / 2, "gd_maxpool2_fwd: bad sizes")) return -1;
    GD_FOR_BC_SLICES(BC, hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(gd_cdiv(Wi / 2, 256), Hi / 2, nz_), dim3(256), 0,
                                            (hipStream_t)stream, x + z0_ * (long)Hi * Wi, Hi, Wi,
                                            y + z0_ * (long)(Hi / 2) * (Wi / 2));)
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_maxpool2_bwd(const float* x, const float* dy, int BC, int Hi, int Wi, float* dx, void* stream) {
    GD_CHECK_ARG(x && dy && dx, "gd_maxpool2_bwd: null pointer");
    GD_CHECK_ARG(Hi % 2 == 0 && Wi % 2 == 0, "gd_maxpool2_bwd: odd sizes unsupported");
    if (check_plane(BC, Hi, Wi, Hi / 2, Wi / 2, "gd_maxpool2_bwd: bad sizes")) return -1;
    GD_FOR_BC_SLICES(BC, hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(gd_cdiv(Wi / 2, 256), Hi / 2, nz_), dim3(256), 0,
                                            (hipStream_t)stream, x + z0_ * (long)Hi * Wi,
                                            dy + z0_ * (long)(Hi / 2) * (Wi / 2), Hi, Wi, dx + z0_ * (long)Hi * Wi);)
    GD_LAUNCH_CHECK();
    return 0;
}
