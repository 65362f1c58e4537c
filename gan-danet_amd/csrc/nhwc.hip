// Pixel-major (NHWC) bf16 activation kernels for the frozen VGG19 feature stack of PerceptualLoss
// (losses.py:13-73; torchvision vgg19.features[:21]): the 3-channel stem conv and its data gradient, 2x2 max
// pooling forward / backward, and the L1 feature distance with its (ReLU-masked) gradient.  The 3x3 convs
// between them are gd_conv3x3_nhwc (conv3x3.hip).  All HBM-bound: 16-byte accesses (8 channels) per lane,
// lanes = consecutive channel octets of consecutive pixels.
#include "common.h"
#include "../../include/gandanet.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = gd_bf2f((unsigned short)(v[k] & 0xFFFFu));
        f[2 * k + 1] = gd_bf2f((unsigned short)(v[k] >> 16));
    }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
    const u32x4_t v = {gd_pack_bf2(f[0], f[1]), gd_pack_bf2(f[2], f[3]), gd_pack_bf2(f[4], f[5]), gd_pack_bf2(f[6], f[7])};
    return v;
}

// Split activations (operand mode "x3" of set_precision("mixed")): a pixel's C logical channels stored as 3 C bf16,
// [hi | lo | hi] with hi = bf16(v), lo = bf16(v - hi) (see disc_nhwc.hip / gd_pack_16_split).  row = the pixel's first element.
__device__ __forceinline__ void store8(unsigned short* row, int C, int c8, const float* v, int split) {
    const u32x4_t hi = pack8(v);
    *reinterpret_cast<u32x4_t*>(row + c8) = hi;
    if (split) {
        float r[8];
        unpack8(hi, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = v[k] - r[k];
        *reinterpret_cast<u32x4_t*>(row + C + c8) = pack8(r);
        *reinterpret_cast<u32x4_t*>(row + 2 * C + c8) = hi;
    }
}
__device__ __forceinline__ void load8(const unsigned short* row, int C, int c8, float* v, int split) {
    unpack8(*reinterpret_cast<const u32x4_t*>(row + c8), v);
    if (split) {
        float l[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(row + C + c8), l);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += l[k];
    }
}

// ---- stem: fp32 NCHW image (B, Ci <= 4, H, W) -> conv3x3 p1 (+bias, ReLU) -> NHWC bf16 (B, H, W, Co) --------------
// thread = (pixel, output-channel octet); weights in LDS as [ci][tap][co]
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ img, int Ci, int H, int W,
                                                      const float* __restrict__ w, const float* __restrict__ bias, int Co,
                                                      int relu, unsigned short* __restrict__ y, long npix_total, int split) {
    extern __shared__ float wl[];                       // Ci * 9 * Co
    const int RS = split ? 3 * Co : Co;
    for (int i = threadIdx.x; i < Ci * 9 * Co; i += 256) {
        const int co = i % Co, t = (i / Co) % 9, ci = i / (9 * Co);
        wl[i] = w[((long)co * Ci + ci) * 9 + t];
    }
    __syncthreads();
    const int oct = Co / 8;
    const long HW = (long)H * W;
    if (Ci == 1 && Co == 64) {
        // the common case (single-channel image, 64 filters): a thread's octet is fixed over the grid stride
        // (256 * gridDim.x is a multiple of 8), so its 9 x 8 weights and 8 biases live in registers -- the LDS-resident
        // form below spends 18 16-byte LDS reads per 16 bytes stored
        const int o8 = (threadIdx.x & 7) * 8;
        float wr[9][8], br[8];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) wr[t][k] = wl[t * 64 + o8 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) br[k] = bias ? bias[o8 + k] : 0.f;
        for (long p = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; p < npix_total; p += ((long)gridDim.x * 256) >> 3) {
            const long b = p / HW;
            const int rem = (int)(p - b * HW);
            const int py = rem / W, px = rem - py * W;
            const float* plane = img + b * HW;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = br[k];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
                const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? plane[(long)iy * W + ix] : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(v, wr[t][k], acc[k]);
            }
            if (relu)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaxf(acc[k], 0.f);
            store8(y + p * RS, 64, o8, acc, split);
        }
        return;
    }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < npix_total * oct; idx += (long)gridDim.x * 256) {
        const long p = idx / oct;
        const int o8 = (int)(idx - p * oct) * 8;
        const long b = p / HW;
        const int rem = (int)(p - b * HW);
        const int py = rem / W, px = rem - py * W;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = bias ? bias[o8 + k] : 0.f;
        for (int ci = 0; ci < Ci; ++ci) {
            const float* plane = img + (b * Ci + ci) * HW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float v = plane[(long)iy * W + ix];
                const float* wp = wl + (ci * 9 + t) * Co + o8;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(v, wp[k], acc[k]);
            }
        }
        if (relu)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = fmaxf(acc[k], 0.f);
        store8(y + p * RS, Co, o8, acc, split);
    }
}

// data gradient of the stem: g (B, H, W, Co) bf16 -> dimg (B, Ci, H, W) fp32.
// A workgroup owns a 16 x 16 pixel tile.  Phase 1: every pixel of the 18 x 18 haloed tile reads its Co gradient
// values ONCE and reduces them against the 9*Ci weight vectors (t[ci][tap] = sum_co g[co] w[co][ci][tap]) into LDS;
// phase 2: an image pixel sums the nine tap planes at its shifted neighbours.  (One thread per pixel gathering
// 9 x Co values from its neighbours directly re-read g nine times: 38.8 ms vs this at B=32, 1024 x 1024, Co=64.)
constexpr int SB_T = 16, SB_HT = SB_T + 2, SB_NH = SB_HT * SB_HT;
__global__ __launch_bounds__(256) void stem_bwd_kernel(const unsigned short* __restrict__ g, int Ci, int H, int W,
                                                      const float* __restrict__ w, int Co, float* __restrict__ dimg,
                                                      int tiles_x, int split) {
    extern __shared__ float smem[];
    const int RS = split ? 3 * Co : Co;
    float* wl = smem;                                   // [ci][tap][co]
    float* T = smem + Ci * 9 * Co;                      // [ci*9 + tap][haloed pixel]
    for (int i = threadIdx.x; i < Ci * 9 * Co; i += 256) {
        const int co = i % Co, t = (i / Co) % 9, ci = i / (9 * Co);
        wl[i] = w[((long)co * Ci + ci) * 9 + t];
    }
    __syncthreads();
    const int b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * SB_T, x0 = tx * SB_T;
    const long HW = (long)H * W;
    for (int hp = threadIdx.x; hp < SB_NH; hp += 256) {
        const int hy = hp / SB_HT, hx = hp - hy * SB_HT;
        const int qy = y0 - 1 + hy, qx = x0 - 1 + hx;
        const bool inside = qy >= 0 && qy < H && qx >= 0 && qx < W;
        for (int ci = 0; ci < Ci; ++ci) {
            float t[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) t[k] = 0.f;
            if (inside) {
                const unsigned short* gp = g + (((long)b * H + qy) * W + qx) * RS;
                for (int o8 = 0; o8 < Co; o8 += 8) {
                    float f[8];
                    load8(gp, Co, o8, f, split);
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float* wp = wl + (ci * 9 + k) * Co + o8;      // uniform address: LDS broadcast
#pragma unroll
                        for (int j = 0; j < 8; ++j) t[k] = fmaf(f[j], wp[j], t[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) T[(ci * 9 + k) * SB_NH + hp] = t[k];
        }
    }
    __syncthreads();
    const int py = threadIdx.x / SB_T, px = threadIdx.x - py * SB_T;
    const int oy = y0 + py, ox = x0 + px;
    if (oy < H && ox < W) {
        for (int ci = 0; ci < Ci; ++ci) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k)     // output pixel q = p - (tap offset) read input pixel p through tap k
                acc += T[(ci * 9 + k) * SB_NH + (py + 2 - k / 3) * SB_HT + (px + 2 - k % 3)];
            dimg[((long)b * Ci + ci) * HW + (long)oy * W + ox] = acc;
        }
    }
}

// ---- 2x2 max pooling, NHWC bf16; thread = (output pixel, channel octet) --------------------------------------------
__global__ __launch_bounds__(256) void pool_fwd_kernel(const unsigned short* __restrict__ x, int H, int W, int C,
                                                      unsigned short* __restrict__ y, long total, int split) {
    const int oct = C / 8, Ho = H / 2, Wo = W / 2, RS = split ? 3 * C : C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long p = idx / oct;
        const int o8 = (int)(idx - p * oct) * 8;
        const long b = p / ((long)Ho * Wo);
        const int rem = (int)(p - b * (long)Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const unsigned short* xp = x + ((b * H + 2 * oy) * (long)W + 2 * ox) * RS;
        float m[8], f[8];
        load8(xp, C, o8, m, split);
        const long offs[3] = {(long)RS, (long)W * RS, (long)W * RS + RS};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            load8(xp + offs[q], C, o8, f, split);
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], f[k]);
        }
        store8(y + p * RS, C, o8, m, split);          // a maximum of values hi + lo splits back into the same hi, lo
    }
}
// dx = dy at the FIRST maximum of each window in row-major order (ATen's tie rule), else 0; with relu_mask the
// result is also gated by x > 0 (backward of the ReLU that produced x)
__global__ __launch_bounds__(256) void pool_bwd_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                      int H, int W, int C, int relu_mask, unsigned short* __restrict__ dx,
                                                      long total, int split) {
    const int oct = C / 8, Ho = H / 2, Wo = W / 2, RS = split ? 3 * C : C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long p = idx / oct;
        const int o8 = (int)(idx - p * oct) * 8;
        const long b = p / ((long)Ho * Wo);
        const int rem = (int)(p - b * (long)Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const long base = ((b * H + 2 * oy) * (long)W + 2 * ox) * RS;
        const long offs[4] = {0L, (long)RS, (long)W * RS, (long)W * RS + RS};
        float v[4][8], g[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) load8(x + base + offs[q], C, o8, v[q], split);
        load8(dy + p * RS, C, o8, g, split);
        float out[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int arg = 0;
            float m = v[0][k];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (v[q][k] > m) { m = v[q][k]; arg = q; }
            const float gk = (relu_mask && !(m > 0.f)) ? 0.f : g[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) out[q][k] = q == arg ? gk : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) store8(dx + base + offs[q], C, o8, out[q], split);
    }
}

// ---- L1 feature distance: partial sums of |a - b| (two-stage, deterministic) and its gradient ----------------------
// split_c: 0 = plain tensors of n8 octets; C > 0 = split tensors of C logical channels per pixel (n8 LOGICAL octets)
__global__ __launch_bounds__(256) void l1_sum_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                    long n8, float* __restrict__ ws, int split_c) {
    __shared__ float red[4];
    float s = 0.f;
    const int oct = split_c ? split_c / 8 : 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float fa[8], fb[8];
        if (split_c) {
            const long px = i / oct;
            const int o8 = (int)(i - px * oct) * 8;
            load8(a + px * 3 * split_c, split_c, o8, fa, 1);
            load8(b + px * 3 * split_c, split_c, o8, fb, 1);
        } else {
            unpack8(reinterpret_cast<const u32x4_t*>(a)[i], fa);
            unpack8(reinterpret_cast<const u32x4_t*>(b)[i], fb);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += fabsf(fa[k] - fb[k]);
    }
    s = gd_block_sum(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ ws, int nparts, float scale,
                                                      float* __restrict__ out, int accumulate) {
    __shared__ double redd[4];
    double acc = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += ws[i];
    acc = gd_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)((redd[0] + redd[1] + redd[2] + redd[3]) * (double)scale);
        out[0] = accumulate ? out[0] + v : v;
    }
}
// g = (*upstream) * inv_n * sign(a - b), gated by a > 0 when relu_mask (a is a ReLU output)
__global__ __launch_bounds__(256) void l1_grad_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                     long n8, const float* __restrict__ upstream, float inv_n, int relu_mask,
                                                     unsigned short* __restrict__ g, int split_c) {
    const float k = (*upstream) * inv_n;
    const int oct = split_c ? split_c / 8 : 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float fa[8], fb[8], o[8];
        const long px = i / oct;
        const int o8 = (int)(i - px * oct) * 8;
        if (split_c) {
            load8(a + px * 3 * split_c, split_c, o8, fa, 1);
            load8(b + px * 3 * split_c, split_c, o8, fb, 1);
        } else {
            unpack8(reinterpret_cast<const u32x4_t*>(a)[i], fa);
            unpack8(reinterpret_cast<const u32x4_t*>(b)[i], fb);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dv = fa[j] - fb[j];
            float v = dv > 0.f ? k : (dv < 0.f ? -k : 0.f);
            if (relu_mask && !(fa[j] > 0.f)) v = 0.f;
            o[j] = v;
        }
        if (split_c) store8(g + px * 3 * split_c, split_c, o8, o, 1);
        else reinterpret_cast<u32x4_t*>(g)[i] = pack8(o);
    }
}

static inline int grid_n(long n, int cap = 16384) {
    long g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

#define NS(s) ((hipStream_t)(s))

extern "C" int gd_nhwc_stem_fwd(const float* img, int B, int Ci, int H, int W, const float* w, const float* bias, int Co,
                                int relu, void* y, int split, void* stream) {
    GD_CHECK_ARG(img && w && y && B > 0 && Ci > 0 && Ci <= 4 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0 && Ci * 9 * Co * 4 <= 65536,
                 "gd_nhwc_stem_fwd: needs Ci <= 4, Co % 8 == 0");
    const long npix = (long)B * H * W;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid_n(npix * (Co / 8))), dim3(256), (size_t)Ci * 9 * Co * 4, NS(stream), img, Ci,
                       H, W, w, bias, Co, relu, (unsigned short*)y, npix, split);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_nhwc_stem_bwd(const void* g, int B, int Ci, int H, int W, const float* w, int Co, float* dimg, int split,
                                void* stream) {
    GD_CHECK_ARG(g && w && dimg && B > 0 && B <= 65535 && Ci > 0 && Ci <= 4 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0,
                 "gd_nhwc_stem_bwd: needs Ci <= 4, Co % 8 == 0");
    const size_t lds = ((size_t)Ci * 9 * Co + (size_t)Ci * 9 * SB_NH) * sizeof(float);
    GD_CHECK_ARG(lds <= 64 * 1024, "gd_nhwc_stem_bwd: Ci * Co too large for the LDS tile");
    const int tiles_x = (W + SB_T - 1) / SB_T, tiles_y = (H + SB_T - 1) / SB_T;
    hipLaunchKernelGGL(stem_bwd_kernel, dim3(tiles_x * tiles_y, B), dim3(256), lds, NS(stream), (const unsigned short*)g, Ci,
                       H, W, w, Co, dimg, tiles_x, split);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_nhwc_maxpool2_fwd(const void* x, int B, int H, int W, int C, void* y, int split, void* stream) {
    GD_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
                 "gd_nhwc_maxpool2_fwd: needs even H, W and C % 8 == 0");
    const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(grid_n(total)), dim3(256), 0, NS(stream), (const unsigned short*)x, H, W, C,
                       (unsigned short*)y, total, split);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_nhwc_maxpool2_bwd(const void* x, const void* dy, int B, int H, int W, int C, int relu_mask, void* dx,
                                    int split, void* stream) {
    GD_CHECK_ARG(x && dy && dx && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
                 "gd_nhwc_maxpool2_bwd: needs even H, W and C % 8 == 0");
    const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_n(total)), dim3(256), 0, NS(stream), (const unsigned short*)x,
                       (const unsigned short*)dy, H, W, C, relu_mask, (unsigned short*)dx, total, split);
    GD_LAUNCH_CHECK();
    return 0;
}
// split_c (here and in gd_nhwc_l1_grad): 0 = plain tensors of n elements; C = split tensors ([hi | lo | hi], 3 C bf16 per pixel) of
// n LOGICAL elements with C channels per pixel
extern "C" int gd_nhwc_l1(const void* a, const void* b, long n, float* out, int accumulate, float* ws, int split_c, void* stream) {
    GD_CHECK_ARG(a && b && out && ws && n > 0 && n % 8 == 0, "gd_nhwc_l1: n must be a positive multiple of 8");
    GD_CHECK_ARG(split_c >= 0 && split_c % 8 == 0 && (split_c == 0 || n % split_c == 0), "gd_nhwc_l1: bad split channel count");
    const int g = grid_n(n / 8, 1024);
    hipLaunchKernelGGL(l1_sum_kernel, dim3(g), dim3(256), 0, NS(stream), (const unsigned short*)a, (const unsigned short*)b,
                       n / 8, ws, split_c);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, NS(stream), ws, g, (float)(1.0 / (double)n), out, accumulate);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_nhwc_l1_grad(const void* a, const void* b, long n, const float* upstream, int relu_mask, void* g, int split_c,
                               void* stream) {
    GD_CHECK_ARG(a && b && upstream && g && n > 0 && n % 8 == 0, "gd_nhwc_l1_grad: n must be a positive multiple of 8");
    GD_CHECK_ARG(split_c >= 0 && split_c % 8 == 0 && (split_c == 0 || n % split_c == 0), "gd_nhwc_l1_grad: bad split channel count");
    hipLaunchKernelGGL(l1_grad_kernel, dim3(grid_n(n / 8)), dim3(256), 0, NS(stream), (const unsigned short*)a,
                       (const unsigned short*)b, n / 8, upstream, (float)(1.0 / (double)n), relu_mask, (unsigned short*)g, split_c);
    GD_LAUNCH_CHECK();
    return 0;
}
