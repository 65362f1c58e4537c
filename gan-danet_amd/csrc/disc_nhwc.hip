// Discriminator1 (discriminator.py:57-77) on pixel-major (NHWC) bf16 activations: the pieces around the stride-2
// 3x3 convolutions of conv3x3.hip (gd_conv3x3_nhwc_s2 / gd_conv3x3_nhwc_s2_dgrad / gd_conv3x3_wgrad):
//   * the 1 -> 64 stem conv (stride 2, + bias + LeakyReLU) from the fp32 NCHW image, its weight/bias gradient and its
//     data gradient (the generator step differentiates through D into the fake image);
//   * x.flatten(1) (discriminator.py:72): NHWC bf16 -> fp32 rows in (c, h, w) order for fc1, and its backward fused
//     with the LeakyReLU mask of conv4;
//   * the pixel-major -> channel-major bf16 transposer that hands a gradient to the weight-gradient kernel as its
//     [co][pixel] operand, fused with the per-channel sums (= the bias gradient).
// All HBM-bound byte shuffling: 16-byte accesses, no MFMA.
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
// bf16 > 0  <=>  sign clear and magnitude non-zero
__device__ __forceinline__ bool bf_pos(unsigned int h) { return (h & 0x7FFFu) && !(h & 0x8000u); }

// Split activations (set_precision("mixed"), operand mode "x3"): a pixel's C logical channels are stored as the 3 C bf16
// [hi | lo | hi] with hi = bf16(v), lo = bf16(v - hi) -- the operand the pixel-major conv kernels take against weights
// split [hi ; hi ; lo] along the contraction axis.  `row` = the pixel's first element, c8 = the channel octet.
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

// ---- stem forward: thread = (output pixel, output-channel octet); weights in LDS as [ci][tap][co] ---------------------
__global__ __launch_bounds__(256) void disc_stem_fwd_kernel(const float* __restrict__ img, int Ci, int H, int W, int Ho, int Wo,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int Co, float slope, unsigned short* __restrict__ y, long npix_total, int split) {
    extern __shared__ float wl[];
    const int RS = split ? 3 * Co : Co;                 // elements per output pixel
    for (int i = threadIdx.x; i < Ci * 9 * Co; i += 256) {
        const int co = i % Co, t = (i / Co) % 9, ci = i / (9 * Co);
        wl[i] = w[((long)co * Ci + ci) * 9 + t];
    }
    __syncthreads();
    const int oct = Co / 8;
    const long HW = (long)H * W, HWo = (long)Ho * Wo;
    if (Ci == 1 && Co == 64) {
        // single-channel image, 64 filters: the thread's octet is fixed over the grid stride, weights and biases in registers
        const int o8 = (threadIdx.x & 7) * 8;
        float wr[9][8], br[8];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) wr[t][k] = wl[t * 64 + o8 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) br[k] = bias ? bias[o8 + k] : 0.f;
        for (long p = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; p < npix_total; p += ((long)gridDim.x * 256) >> 3) {
            const long b = p / HWo;
            const int rem = (int)(p - b * HWo);
            const int py = rem / Wo, px = rem - py * Wo;
            const float* plane = img + b * HW;
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = br[k];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = 2 * py + t / 3 - 1, ix = 2 * px + t % 3 - 1;
                const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? plane[(long)iy * W + ix] : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(v, wr[t][k], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = acc[k] > 0.f ? acc[k] : acc[k] * slope;
            store8(y + p * RS, 64, o8, acc, split);
        }
        return;
    }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < npix_total * oct; idx += (long)gridDim.x * 256) {
        const long p = idx / oct;
        const int o8 = (int)(idx - p * oct) * 8;
        const long b = p / HWo;
        const int rem = (int)(p - b * HWo);
        const int py = rem / Wo, px = rem - py * Wo;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = bias ? bias[o8 + k] : 0.f;
        for (int ci = 0; ci < Ci; ++ci) {
            const float* plane = img + (b * Ci + ci) * HW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = 2 * py + t / 3 - 1, ix = 2 * px + t % 3 - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float v = plane[(long)iy * W + ix];
                const float* wp = wl + (ci * 9 + t) * Co + o8;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(v, wp[k], acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] > 0.f ? acc[k] : acc[k] * slope;
        store8(y + p * RS, Co, o8, acc, split);
    }
}

// ---- stem weight / bias gradient: dw[co][ci][tap] = sum_p g[p][co] x[ci][2p + tap - 1], db[co] = sum_p g[p][co] -------
// thread = (output pixel, channel octet), grid-strided; blockIdx.y = ci.  Ten accumulators per channel (nine taps + the
// plain sum), reduced over the lanes of equal octet by xor shuffles, over the waves through LDS, then one atomic per
// value and workgroup.  Requires Co == 64 (octet = tid & 7).
__global__ __launch_bounds__(256) void disc_stem_wgrad_kernel(const unsigned short* __restrict__ g, const float* __restrict__ img,
                                                             int Ci, int H, int W, int Ho, int Wo, float* __restrict__ dw,
                                                             float* __restrict__ db, long npix_total, int split) {
    constexpr int Co = 64, OCT = 8;
    const int RS = split ? 3 * Co : Co;
    __shared__ float red[4][OCT][80];
    const int ci = blockIdx.y;
    const long HW = (long)H * W, HWo = (long)Ho * Wo;
    float acc[8][10];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int t = 0; t < 10; ++t) acc[k][t] = 0.f;
    const int o8 = (threadIdx.x & 7) * 8;
    for (long p = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; p < npix_total; p += ((long)gridDim.x * 256) >> 3) {
        const long b = p / HWo;
        const int rem = (int)(p - b * HWo);
        const int py = rem / Wo, px = rem - py * Wo;
        float gv[8], xv[9];
        load8(g + p * RS, Co, o8, gv, split);
        const float* plane = img + (b * Ci + ci) * HW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = 2 * py + t / 3 - 1, ix = 2 * px + t % 3 - 1;
            xv[t] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? plane[(long)iy * W + ix] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[k][t] = fmaf(gv[k], xv[t], acc[k][t]);
            acc[k][9] += gv[k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            float v = acc[k][t];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 8) red[wave][lane][k * 10 + t] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < OCT * 80; i += 256) {
        const int o = i / 80, kt = i - o * 80, k = kt / 10, t = kt - k * 10;
        const float v = red[0][o][kt] + red[1][o][kt] + red[2][o][kt] + red[3][o][kt];
        const int co = o * 8 + k;
        if (t < 9) atomicAdd(dw + ((long)co * Ci + ci) * 9 + t, v);
        else if (ci == 0 && db) atomicAdd(db + co, v);
    }
}

// ---- stem data gradient: g (B, Ho, Wo, Co) bf16 -> dimg (B, Ci, H, W) fp32 ---------------------------------------------
// dimg[Y][X] = sum over taps with ky = Y+1 (mod 2), kx = X+1 (mod 2) of T[tap][(Y+1-ky)/2][(X+1-kx)/2],
// T[tap][p] = sum_co g[p][co] w[co][ci][tap].  A workgroup owns 16 x 16 gradient pixels (= 32 x 32 image pixels) plus the
// one-pixel high-side halo the odd image rows / columns reach into; phase 1 computes T once per gradient pixel into LDS.
constexpr int SD_T = 16, SD_HT = SD_T + 1, SD_NH = SD_HT * SD_HT;
__global__ __launch_bounds__(256) void disc_stem_dgrad_kernel(const unsigned short* __restrict__ g, int Ci, int H, int W, int Ho,
                                                             int Wo, const float* __restrict__ w, int Co,
                                                             float* __restrict__ dimg, int tiles_x, int split) {
    extern __shared__ float smem[];
    const int RS = split ? 3 * Co : Co;
    float* wl = smem;                                   // [ci][tap][co]
    float* T = smem + Ci * 9 * Co;                      // [ci*9 + tap][haloed gradient pixel]
    for (int i = threadIdx.x; i < Ci * 9 * Co; i += 256) {
        const int co = i % Co, t = (i / Co) % 9, ci = i / (9 * Co);
        wl[i] = w[((long)co * Ci + ci) * 9 + t];
    }
    __syncthreads();
    const int b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int u0 = ty * SD_T, v0 = tx * SD_T;
    for (int hp = threadIdx.x; hp < SD_NH; hp += 256) {
        const int hy = hp / SD_HT, hx = hp - hy * SD_HT;
        const int qy = u0 + hy, qx = v0 + hx;
        const bool inside = qy < Ho && qx < Wo;
        for (int ci = 0; ci < Ci; ++ci) {
            float t[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) t[k] = 0.f;
            if (inside) {
                const unsigned short* gp = g + (((long)b * Ho + qy) * Wo + qx) * RS;
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
            for (int k = 0; k < 9; ++k) T[(ci * 9 + k) * SD_NH + hp] = t[k];
        }
    }
    __syncthreads();
    const long HW = (long)H * W;
    for (int q = threadIdx.x; q < 4 * SD_T * SD_T; q += 256) {
        const int ly = q / (2 * SD_T), lx = q - ly * (2 * SD_T);
        const int Y = 2 * u0 + ly, X = 2 * v0 + lx;
        if (Y >= H || X >= W) continue;
        const int py = ly & 1, px = lx & 1, uy = ly >> 1, ux = lx >> 1;
        for (int ci = 0; ci < Ci; ++ci) {
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (((ky + 1) & 1) != py) continue;
                const int ry = uy + (ky == 0 ? 1 : 0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (((kx + 1) & 1) != px) continue;
                    const int rx = ux + (kx == 0 ? 1 : 0);
                    acc += T[(ci * 9 + ky * 3 + kx) * SD_NH + ry * SD_HT + rx];
                }
            }
            dimg[((long)b * Ci + ci) * HW + (long)Y * W + X] = acc;
        }
    }
}

// ---- flatten: y (B, HW, C) bf16 -> f (B, C * HW) fp32, (c, h, w) order; lanes run along the pixel index ---------------
__global__ __launch_bounds__(256) void flatten_fwd_kernel(const unsigned short* __restrict__ y, int HW, int C,
                                                         float* __restrict__ f, long total, int split) {
    const int oct = C / 8, RS = split ? 3 * C : C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int p = (int)(idx % HW);
        const long t = idx / HW;
        const int o8 = (int)(t % oct) * 8;
        const long b = t / oct;
        float v[8];
        load8(y + (b * HW + p) * RS, C, o8, v, split);
        float* dst = f + (b * C + o8) * HW + p;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[(long)k * HW] = v[k];
    }
}
// backward: g (B, HW, C) bf16 = df (B, C * HW) * LeakyReLU'(y)
__global__ __launch_bounds__(256) void flatten_bwd_kernel(const float* __restrict__ df, const unsigned short* __restrict__ y,
                                                         float slope, int HW, int C, unsigned short* __restrict__ g, long total,
                                                         int split) {
    const int oct = C / 8, RS = split ? 3 * C : C;      // the sign of a split value is the sign of its hi part
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int p = (int)(idx % HW);
        const long t = idx / HW;
        const int o8 = (int)(t % oct) * 8;
        const long b = t / oct;
        const u32x4_t yv = *reinterpret_cast<const u32x4_t*>(y + (b * HW + p) * RS + o8);
        const float* src = df + (b * C + o8) * HW + p;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned int h = (k & 1) ? (yv[k >> 1] >> 16) : (yv[k >> 1] & 0xFFFFu);
            v[k] = src[(long)k * HW] * (bf_pos(h) ? 1.f : slope);
        }
        store8(g + (b * HW + p) * RS, C, o8, v, split);
    }
}

// ---- g (B, HW, C) bf16 -> gt (B, C, HW) bf16 + channel sums --------------------------------------------------------------
// blockIdx.x = 64-channel group, blockIdx.y strides over the (image, 64-pixel tile) list; a 64 x 64 tile goes through LDS
// ([channel][pixel], rows padded to 72), the channel sums stay in registers until the end (one atomic per channel and
// workgroup: per-tile atomics would pile millions of adds onto C addresses)
constexpr int TT = 64, TLD = TT + 8;
// ldg = elements per pixel of g (C, or 3 C for a split gradient whose hi / lo parts are transposed by one launch each: g then
// points at the part's first channel and the channel sums of the two launches add up to the sums of hi + lo)
__global__ __launch_bounds__(256) void nhwc_to_nchw16_kernel(const unsigned short* __restrict__ g, int B, int HW, int C, int ldg,
                                                            unsigned short* __restrict__ gt, float* __restrict__ csum) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[TT * TLD];
    __shared__ float part[4][TT];
    const int c0 = blockIdx.x * TT;
    const int tiles_img = (HW + TT - 1) / TT;
    const long ntiles = (long)B * tiles_img;
    const int tid = threadIdx.x;
    const bool vec = (HW % 8) == 0;
    float sum = 0.f;                                     // thread (c = tid & 63, quarter = tid >> 6): 16 pixels of channel c
    for (long t = blockIdx.y; t < ntiles; t += gridDim.y) {
        const long b = t / tiles_img;
        const int p0 = (int)(t - b * tiles_img) * TT;
        // read: item = (pixel, octet), 512 items
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int it = tid + k * 256;
            const int pl = it >> 3, o8 = (it & 7) * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (p0 + pl < HW && c0 + o8 < C) v = *reinterpret_cast<const u32x4_t*>(g + (b * HW + p0 + pl) * ldg + c0 + o8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tile[(o8 + 2 * j) * TLD + pl] = (unsigned short)(v[j] & 0xFFFFu);
                tile[(o8 + 2 * j + 1) * TLD + pl] = (unsigned short)(v[j] >> 16);
            }
        }
        __syncthreads();
        // write: item = (channel, pixel octet)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int it = tid + k * 256;
            const int cl = it >> 3, q8 = (it & 7) * 8;
            if (c0 + cl < C && p0 + q8 < HW) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(tile + cl * TLD + q8);
                unsigned short* dst = gt + (b * C + c0 + cl) * HW + p0 + q8;
                if (vec) {
                    *reinterpret_cast<u32x4_t*>(dst) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (p0 + q8 + j < HW) dst[j] = (unsigned short)((j & 1) ? (v[j >> 1] >> 16) : (v[j >> 1] & 0xFFFFu));
                }
            }
        }
        if (csum) {
            const int cl = tid & 63, q = tid >> 6;
#pragma unroll
            for (int j = 0; j < 16; ++j) sum += gd_bf2f(tile[cl * TLD + q * 16 + j]);     // out-of-range pixels are zeros
        }
        __syncthreads();
    }
    if (csum) {
        part[tid >> 6][tid & 63] = sum;
        __syncthreads();
        if (tid < TT && c0 + tid < C) atomicAdd(csum + c0 + tid, part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
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

extern "C" int gd_disc_stem_fwd(const float* img, int B, int Ci, int H, int W, const float* w, const float* bias, int Co,
                                float slope, void* y, int split, void* stream) {
    GD_CHECK_ARG(img && w && y && B > 0 && Ci > 0 && Ci <= 4 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0 && Ci * 9 * Co * 4 <= 65536,
                 "gd_disc_stem_fwd: needs Ci <= 4, Co % 8 == 0");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    hipLaunchKernelGGL(disc_stem_fwd_kernel, dim3(grid_n(npix * (Co / 8))), dim3(256), (size_t)Ci * 9 * Co * 4, NS(stream), img,
                       Ci, H, W, Ho, Wo, w, bias, Co, slope, (unsigned short*)y, npix, split);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_disc_stem_wgrad(const void* g, const float* img, int B, int Ci, int H, int W, int Co, float* dw, float* db,
                                  int split, void* stream) {
    GD_CHECK_ARG(g && img && dw && B > 0 && Ci > 0 && Ci <= 4 && H > 0 && W > 0 && Co == 64,
                 "gd_disc_stem_wgrad: needs Ci <= 4, Co == 64");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    GD_CHECK_ARG(hipMemsetAsync(dw, 0, (size_t)Co * Ci * 9 * sizeof(float), NS(stream)) == hipSuccess, "gd_disc_stem_wgrad: memset failed");
    if (db) GD_CHECK_ARG(hipMemsetAsync(db, 0, (size_t)Co * sizeof(float), NS(stream)) == hipSuccess, "gd_disc_stem_wgrad: memset failed");
    hipLaunchKernelGGL(disc_stem_wgrad_kernel, dim3(grid_n(npix * 8, 1024), Ci), dim3(256), 0, NS(stream), (const unsigned short*)g,
                       img, Ci, H, W, Ho, Wo, dw, db, npix, split);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_disc_stem_dgrad(const void* g, int B, int Ci, int H, int W, const float* w, int Co, float* dimg, int split,
                                  void* stream) {
    GD_CHECK_ARG(g && w && dimg && B > 0 && B <= 65535 && Ci > 0 && Ci <= 4 && H > 0 && W > 0 && Co > 0 && Co % 8 == 0,
                 "gd_disc_stem_dgrad: needs Ci <= 4, Co % 8 == 0");
    const size_t lds = ((size_t)Ci * 9 * Co + (size_t)Ci * 9 * SD_NH) * sizeof(float);
    GD_CHECK_ARG(lds <= 64 * 1024, "gd_disc_stem_dgrad: Ci * Co too large for the LDS tile");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    // tiles over the GRADIENT grid; a tile also writes the odd image row / column past its last gradient pixel
    const int tiles_x = (Wo + SD_T - 1) / SD_T, tiles_y = (Ho + SD_T - 1) / SD_T;
    hipLaunchKernelGGL(disc_stem_dgrad_kernel, dim3(tiles_x * tiles_y, B), dim3(256), lds, NS(stream), (const unsigned short*)g,
                       Ci, H, W, Ho, Wo, w, Co, dimg, tiles_x, split);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_nhwc_flatten_fwd(const void* y, int B, int HW, int C, float* f, int split, void* stream) {
    GD_CHECK_ARG(y && f && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "gd_nhwc_flatten_fwd: C must be a multiple of 8");
    const long total = (long)B * HW * (C / 8);
    hipLaunchKernelGGL(flatten_fwd_kernel, dim3(grid_n(total)), dim3(256), 0, NS(stream), (const unsigned short*)y, HW, C, f, total, split);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_nhwc_flatten_bwd(const float* df, const void* y, float slope, int B, int HW, int C, void* g, int split,
                                   void* stream) {
    GD_CHECK_ARG(df && y && g && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "gd_nhwc_flatten_bwd: C must be a multiple of 8");
    const long total = (long)B * HW * (C / 8);
    hipLaunchKernelGGL(flatten_bwd_kernel, dim3(grid_n(total)), dim3(256), 0, NS(stream), df, (const unsigned short*)y, slope, HW,
                       C, (unsigned short*)g, total, split);
    GD_LAUNCH_CHECK();
    return 0;
}

// split = 1: g holds 3 C channels per pixel [hi | lo | hi]; gt receives TWO (B, C, HW) images, hi then lo (the dY operands of
// the weight gradient's three accumulating launches), csum the sums of hi + lo
extern "C" int gd_nhwc_to_nchw16(const void* g, int B, int HW, int C, void* gt, float* csum, int split, void* stream) {
    GD_CHECK_ARG(g && gt && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "gd_nhwc_to_nchw16: C must be a multiple of 8");
    if (csum) GD_CHECK_ARG(hipMemsetAsync(csum, 0, (size_t)C * sizeof(float), NS(stream)) == hipSuccess, "gd_nhwc_to_nchw16: memset failed");
    const long ntiles = (long)B * ((HW + TT - 1) / TT);
    const int cg = (C + TT - 1) / TT;
    long gy = 2048 / cg;
    if (gy > ntiles) gy = ntiles;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(nhwc_to_nchw16_kernel, dim3(cg, (unsigned)gy), dim3(256), 0, NS(stream), (const unsigned short*)g, B, HW, C,
                       split ? 3 * C : C, (unsigned short*)gt, csum);
    if (split)
        hipLaunchKernelGGL(nhwc_to_nchw16_kernel, dim3(cg, (unsigned)gy), dim3(256), 0, NS(stream), (const unsigned short*)g + C, B, HW,
                           C, 3 * C, (unsigned short*)gt + (long)B * C * HW, csum);
    GD_LAUNCH_CHECK();
    return 0;
}
