// Pointwise / small-reduction kernels for gfx950: activations, axpby, slab copies, row softmax (fwd/bwd),
// dot/sum reductions, transposes, the losses of the G+D step (BCE-with-logits, MSE, L1, TV, SSIM),
// AdamW, and the fp32 -> bf16 pack/transpose feeding the fused PAM kernels.  All HBM-bound: grid-stride,
// 16-byte accesses where alignment allows, two-stage deterministic reductions (no float atomics).
#include "common.h"
#include "../../include/gandanet.h"

#include <math.h>

namespace {

constexpr int RED_BLOCKS = 1024;  // stage-1 workgroups of the scalar reductions

static inline int grid_for(long n, int per_block = 256 * 4) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == GD_ACT_RELU) return fmaxf(v, 0.f);
    if (act == GD_ACT_LEAKY02) return v >= 0.f ? v : 0.2f * v;
    if (act == GD_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = act_f(x[i], act);
}
__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long n,
                               int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float o = y[i], g = dy[i];
        float v = g;
        if (act == GD_ACT_RELU) v = o > 0.f ? g : 0.f;
        else if (act == GD_ACT_LEAKY02) v = o >= 0.f ? g : 0.2f * g;  // sign(y) == sign(x) for leaky
        else if (act == GD_ACT_SIGMOID) v = g * o * (1.f - o);
        dx[i] = v;
    }
}
__global__ void axpby_kernel(const float* __restrict__ x, float a, float* __restrict__ y, float b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = b == 0.f ? a * x[i] : fmaf(a, x[i], b * y[i]);
}
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, long n,
                                 int accumulate) {
    const float k = *s;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = accumulate ? fmaf(k, x[i], y[i]) : k * x[i];
}
// dst[b][r][c] = src[b][r][c] for r < R, c < Cc with independent leading dimensions (compaction of padded planes)
__global__ void copy_rows_kernel(const float* __restrict__ src, long s_bs, long s_ld, float* __restrict__ dst, long d_bs,
                                 long d_ld, int R, int Cc) {
    const int b = blockIdx.y;
    const long total = (long)R * Cc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / Cc), c = (int)(i - (long)r * Cc);
        dst[(long)b * d_bs + (long)r * d_ld + c] = src[(long)b * s_bs + (long)r * s_ld + c];
    }
}
__global__ void copy_slab_kernel(const float* __restrict__ src, long s_bs, float* __restrict__ dst, long d_bs, long chw,
                                 int accumulate) {
    const int b = blockIdx.y;
    const float* s = src + (long)b * s_bs;
    float* d = dst + (long)b * d_bs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += (long)gridDim.x * blockDim.x)
        d[i] = accumulate ? d[i] + s[i] : s[i];
}
// 16 bytes per lane (the scalar form above moved 1.6 TB/s on the 1 GB slab copies of the dense blocks): chw, both batch strides
// multiples of 4 floats and both bases 16-byte aligned (checked by the host)
__global__ __launch_bounds__(256) void copy_slab_vec_kernel(const float4* __restrict__ src, long s_bs4, float4* __restrict__ dst,
                                                           long d_bs4, long n4, int accumulate) {
    const int b = blockIdx.y;
    const float4* s = src + (long)b * s_bs4;
    float4* d = dst + (long)b * d_bs4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = s[i];
        if (accumulate) {
            const float4 o = d[i];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        d[i] = v;
    }
}

// ---- row softmax ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int cols,
                                                          float sign) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const float* p = x + row * cols;
    float* q = y + row * cols;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, sign * p[i]);
    m = gd_block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) s += __expf(sign * p[i] - m);
    s = gd_block_sum(s, red);
    const float inv = 1.f / s;
    for (int i = threadIdx.x; i < cols; i += 256) q[i] = __expf(sign * p[i] - m) * inv;
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                              float* __restrict__ dx, int cols, float sign) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const float* pp = p + row * cols;
    const float* pd = dp + row * cols;
    float* po = dx + row * cols;
    float s = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) s = fmaf(pp[i], pd[i], s);
    s = gd_block_sum(s, red);
    for (int i = threadIdx.x; i < cols; i += 256) po[i] = sign * pp[i] * (pd[i] - s);
}

// ---- generic two-stage scalar reductions -----------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ ws, int nparts, float scale,
                                                          float* __restrict__ out, int accumulate) {
    __shared__ double redd[4];
    double a = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += ws[i];
    a = gd_wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)((redd[0] + redd[1] + redd[2] + redd[3]) * (double)scale);
        out[0] = accumulate ? out[0] + v : v;
    }
}

__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                 float* __restrict__ ws) {
    __shared__ float red[8];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        s = b ? fmaf(a[i], b[i], s) : s + a[i];
    s = gd_block_sum(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

// ---- transposes -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ s, float* __restrict__ t, int R, int Cc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float* sp = s + (long)b * R * Cc;
    float* tp = t + (long)b * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? sp[(long)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < R) tp[(long)c * R + r] = tile[tx][i];
    }
}
__global__ void add_transpose_kernel(const float* __restrict__ a, float* __restrict__ out, int n) {
    const int b = blockIdx.y;
    const float* ap = a + (long)b * n * n;
    float* op = out + (long)b * n * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n * n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / n), c = (int)(i - (long)r * n);
        op[i] = ap[i] + ap[(long)c * n + r];
    }
}

// fp32 (R, Cc) plane -> bf16 plain copy (zero padded to Rp x ldp) and/or bf16 transpose (zero padded Ccp x ldt)
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ s, long s_bs, int R, int Cc,
                                                       const float* __restrict__ scale, float scale_imm,
                                                       unsigned short* __restrict__ plain, int Rp, int ldp,
                                                       unsigned short* __restrict__ tr, int Ccp, int ldt, int perm16,
                                                       int ones_row, int f16) {
    __shared__ float tile[32][33];
    // 16-bit output type: bf16 (round to nearest even) or IEEE fp16 (BASELINE config 5)
    auto cvt = [f16](float v) -> unsigned short {
        if (f16) return __builtin_bit_cast(unsigned short, (_Float16)v);
        return gd_f2bf(v);
    };
    const int b = blockIdx.z;
    const float* sp = s + (long)b * s_bs;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float k = scale ? scale_imm * (*scale) : scale_imm;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = r == ones_row ? 1.f : (r < R && c < Cc) ? k * sp[(long)r * Cc + c] : 0.f;
        tile[i][tx] = v;
        if (plain && r < Rp && c < ldp) {
            // perm16: inside every group of 16 columns store [0-3, 8-11, 4-7, 12-15], the order in which a lane half
            // consumes an accumulator-row-ordered k-step -- its fragment becomes ONE 16-byte read
            const int cc = perm16 ? ((c & ~15) | ((c & 3) | ((c & 4) << 1) | ((c & 8) >> 1))) : c;
            plain[(long)b * Rp * ldp + (long)r * ldp + cc] = cvt(v);
        }
    }
    if (tr) {
        __syncthreads();
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, r = r0 + tx;
            if (c < Ccp && r < ldt) tr[(long)b * Ccp * ldt + (long)c * ldt + r] = cvt(tile[tx][i]);
        }
    }
}

// The same on 64 x 64 tiles with 16-byte accesses (float4 reads, 8 x 16-bit stores) for the aligned case (Cc % 4 == 0,
// output leading dimensions % 8 == 0): the 32 x 32 kernel above moves 2-byte elements in 64-byte runs.  Optional per-row
// affine + ReLU (row_scale / row_shift: the BatchNorm+ReLU prologue of a dense layer, applied while packing its input).
__global__ __launch_bounds__(256) void pack16_tile64_kernel(const float* __restrict__ s, long s_bs, int R, int Cc,
                                                           const float* __restrict__ scale, float scale_imm,
                                                           const float* __restrict__ row_scale,
                                                           const float* __restrict__ row_shift, int relu,
                                                           unsigned short* __restrict__ plain, int Rp, int ldp,
                                                           unsigned short* __restrict__ tr, int Ccp, int ldt, int perm16,
                                                           int ones_row, int f16) {
    constexpr int TLD = 72;                               // 144-byte rows: 16-byte aligned
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * TLD];
    const int b = blockIdx.z;
    const float* sp = s + (long)b * s_bs;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const float k = scale ? scale_imm * (*scale) : scale_imm;
    auto cvt2 = [f16](float a, float bb) -> unsigned int {
        if (f16) {
            const unsigned int lo = __builtin_bit_cast(unsigned short, (_Float16)a), hi = __builtin_bit_cast(unsigned short, (_Float16)bb);
            return lo | (hi << 16);
        }
        return gd_pack_bf2(a, bb);
    };
    {
        const int cg = (tid & 15) * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int rl = (tid >> 4) + 16 * kk;
            const int r = r0 + rl, c = c0 + cg;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r == ones_row) {
                v = make_float4(1.f, 1.f, 1.f, 1.f);
            } else if (r < R && c < Cc) {                 // Cc % 4 == 0: a float4 is all in or all out
                v = *reinterpret_cast<const float4*>(sp + (long)r * Cc + c);
                float a = k, sh = 0.f;
                if (row_scale) { a *= row_scale[r]; sh = row_shift[r]; }
                v.x = fmaf(v.x, a, sh); v.y = fmaf(v.y, a, sh); v.z = fmaf(v.z, a, sh); v.w = fmaf(v.w, a, sh);
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            uint2 w;
            w.x = cvt2(v.x, v.y);
            w.y = cvt2(v.z, v.w);
            *reinterpret_cast<uint2*>(tile + rl * TLD + cg) = w;
        }
    }
    __syncthreads();
    if (plain) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int it = tid + 256 * kk;
            const int rl = it >> 3, g = it & 7;
            const int r = r0 + rl, c = c0 + 8 * g;
            if (r < Rp && c < ldp) {
                uint4 o;
                if (perm16) {       // output positions [0-3, 4-7 | 8-11, 12-15] of a 16-group hold source columns [0-3, 8-11 | 4-7, 12-15]
                    const int base = 16 * (g >> 1) + 4 * (g & 1);
                    const uint2 lo = *reinterpret_cast<const uint2*>(tile + rl * TLD + base);
                    const uint2 hi = *reinterpret_cast<const uint2*>(tile + rl * TLD + base + 8);
                    o = make_uint4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    o = *reinterpret_cast<const uint4*>(tile + rl * TLD + 8 * g);
                }
                *reinterpret_cast<uint4*>(plain + (long)b * Rp * ldp + (long)r * ldp + c) = o;
            }
        }
    }
    if (tr) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int it = tid + 256 * kk;
            const int cl = it >> 3, rg = (it & 7) * 8;
            const int c = c0 + cl, r = r0 + rg;
            if (c < Ccp && r < ldt) {
                unsigned int w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w[j] = (unsigned int)tile[(rg + 2 * j) * TLD + cl] | ((unsigned int)tile[(rg + 2 * j + 1) * TLD + cl] << 16);
                *reinterpret_cast<uint4*>(tr + (long)b * Ccp * ldt + (long)c * ldt + r) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
}

// ---- split-bf16 ("x3") operands --------------------------------------------------------------------------------------
// set_precision("mixed") keeps the convolutions at ~2^-16 relative error on the bf16 matrix pipe: every operand is split
// v = hi + lo (hi = bf16(v), lo = bf16(v - hi)) and a product runs as hi*hi + lo*hi + hi*lo -- three bf16 MFMAs with fp32
// accumulation, 5x the rate of the exact f32 MFMA.  The split rides in the operand LAYOUT, so the conv kernels are the
// plain 16-bit ones: the activation pack below writes up to three copies of the tile, each holding the hi or the lo part
// (copy j at `cs` elements behind copy j - 1; bit j of `pattern` set = lo), e.g. pixel-major [hi | lo | hi] as 3 C
// channels for the forward / data gradient against the weights [hi ; hi ; lo] (gd_split3_weights), or separate hi / lo
// images for the weight gradient's three accumulating launches.  Same 64 x 64 tiling, per-row affine + ReLU and
// alignment contract as pack16_tile64_kernel (Cc % 4 == 0, leading dimensions % 8 == 0); bf16 only.
__global__ __launch_bounds__(256) void pack16_split_kernel(const float* __restrict__ s, long s_bs, int R, int Cc,
                                                          const float* __restrict__ row_scale,
                                                          const float* __restrict__ row_shift, int relu,
                                                          unsigned short* __restrict__ plain, long p_bs, int ldp, long p_cs,
                                                          int p_n, int p_pat, unsigned short* __restrict__ tr, long t_bs,
                                                          int ldt, long t_cs, int t_n, int t_pat,
                                                          const float* __restrict__ mask, long m_bs) {
    constexpr int TLD = 72;
    __shared__ __attribute__((aligned(16))) unsigned short tile[2][64 * TLD];      // [0] hi, [1] lo
    const int b = blockIdx.z;
    const float* sp = s + (long)b * s_bs;
    const float* mp = mask ? mask + (long)b * m_bs : nullptr;       // ReLU backward fused: v <- v where mask > 0, else 0
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    {
        const int cg = (tid & 15) * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int rl = (tid >> 4) + 16 * kk;
            const int r = r0 + rl, c = c0 + cg;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R && c < Cc) {
                v = *reinterpret_cast<const float4*>(sp + (long)r * Cc + c);
                if (row_scale) {
                    const float a = row_scale[r], sh = row_shift[r];
                    v.x = fmaf(v.x, a, sh); v.y = fmaf(v.y, a, sh); v.z = fmaf(v.z, a, sh); v.w = fmaf(v.w, a, sh);
                }
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (mp) {
                    const float4 m = *reinterpret_cast<const float4*>(mp + (long)r * Cc + c);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                }
            }
            uint2 hi;
            hi.x = gd_pack_bf2(v.x, v.y);
            hi.y = gd_pack_bf2(v.z, v.w);
            uint2 lo;
            lo.x = gd_pack_bf2(v.x - gd_bf2f((unsigned short)(hi.x & 0xFFFFu)), v.y - gd_bf2f((unsigned short)(hi.x >> 16)));
            lo.y = gd_pack_bf2(v.z - gd_bf2f((unsigned short)(hi.y & 0xFFFFu)), v.w - gd_bf2f((unsigned short)(hi.y >> 16)));
            *reinterpret_cast<uint2*>(tile[0] + rl * TLD + cg) = hi;
            *reinterpret_cast<uint2*>(tile[1] + rl * TLD + cg) = lo;
        }
    }
    __syncthreads();
    if (plain) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int it = tid + 256 * kk;
            const int rl = it >> 3, g = it & 7;
            const int r = r0 + rl, c = c0 + 8 * g;
            if (r < R && c < ldp) {
                for (int j = 0; j < p_n; ++j) {
                    const uint4 o = *reinterpret_cast<const uint4*>(tile[(p_pat >> j) & 1] + rl * TLD + 8 * g);
                    *reinterpret_cast<uint4*>(plain + (long)j * p_cs + (long)b * p_bs + (long)r * ldp + c) = o;
                }
            }
        }
    }
    if (tr) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int it = tid + 256 * kk;
            const int cl = it >> 3, rg = (it & 7) * 8;
            const int c = c0 + cl, r = r0 + rg;
            if (c < Cc && r < R) {                       // R % 8 == 0 (checked by the host): 8 rows are all in or all out
                for (int j = 0; j < t_n; ++j) {
                    const unsigned short* tl = tile[(t_pat >> j) & 1];
                    unsigned int w[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        w[q] = (unsigned int)tl[(rg + 2 * q) * TLD + cl] | ((unsigned int)tl[(rg + 2 * q + 1) * TLD + cl] << 16);
                    *reinterpret_cast<uint4*>(tr + (long)j * t_cs + (long)b * t_bs + (long)c * ldt + r) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    }
}
// w (A, Bn, Cn) fp32 -> out (A, 3 Bn, Cn) fp32: [hi ; hi ; lo] along the middle axis (hi = the bf16 rounding of w as a
// float, lo = w - hi: exact in fp32; the 16-bit weight pack then rounds lo to bf16)
__global__ __launch_bounds__(256) void split3_weights_kernel(const float* __restrict__ w, long A, long Bn, long Cn,
                                                            float* __restrict__ out) {
    const long total = A * Bn * Cn;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long a = i / (Bn * Cn), rem = i - a * (Bn * Cn);
        const float v = w[i];
        const float hi = gd_bf2f((unsigned short)(gd_pack_bf2(v, 0.f) & 0xFFFFu));
        float* o = out + a * 3 * Bn * Cn + rem;
        o[0] = hi;
        o[Bn * Cn] = hi;
        o[2 * Bn * Cn] = v - hi;
    }
}

// ---- CustomDataset.apply_augmentation (datasets.py:181-208) as one gather -------------------------------------------
// op word of a sample: bit 0 horizontal flip, bit 1 vertical flip, bits 2-3 number of 90-degree turns (torch.rot90,
// dims [1, 2]), bit 4 additive noise.  The three geometric steps compose to one index map; tiles are square when
// the number of turns is odd.  dst[b][c][y][x] = src[b][c][sy][sx] (+ scale * noise[b][c][y][x])
__global__ void augment_d4_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                  const int* __restrict__ ops, const float* __restrict__ noise, float scale, long total) {
#pragma clang fp contract(off)      // x + noise * 0.05 stays a rounded multiply and a rounded add, like the reference
    const long plane = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bc = i / plane;
        const int rem = (int)(i - bc * plane);
        const int op = ops[bc / C];
        const int k = (op >> 2) & 3;
        // output geometry: (H, W) turns into (W, H) for odd k; square tiles keep the strides simple
        const int y = rem / W, x = rem - y * W;
        int ry, rx;                                   // source of the rotation: R1[i][j] = M[j][n-1-i], ...
        if (k == 0) { ry = y; rx = x; }
        else if (k == 1) { ry = x; rx = W - 1 - y; }
        else if (k == 2) { ry = H - 1 - y; rx = W - 1 - x; }
        else { ry = H - 1 - x; rx = y; }
        if (op & 2) ry = H - 1 - ry;                  // undo the vertical flip (applied before the rotation)
        if (op & 1) rx = W - 1 - rx;                  // undo the horizontal flip (applied first)
        float v = src[bc * plane + (long)ry * W + rx];
        if (noise && (op & 16)) v = v + scale * noise[i];
        dst[i] = v;
    }
}

// ---- attention gates of SqueezeExcitation / CBAMBlock (generator.py:70-101) ------------------------------------
// y[b][c][p] = x[b][c][p] * att ; mode 0: att[b][c] (channel gate), mode 1: att[b][p] (spatial gate)
__global__ void bcast_mul_kernel(const float* __restrict__ x, const float* __restrict__ att, float* __restrict__ y, int C,
                                 long HW, int mode, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bc = i / HW;
        const float a = mode == 0 ? att[bc] : att[(bc / C) * HW + (i - bc * HW)];
        y[i] = x[i] * a;
    }
}
// out[row] = sum_j a[row][j] * b[row][j]
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, long rows, long n) {
    __shared__ float red[4];
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* ap = a + row * n;
        const float* bp = b + row * n;
        float s = 0.f;
        for (long j = threadIdx.x; j < n; j += 256) s = fmaf(ap[j], bp[j], s);
        s = gd_block_sum(s, red);
        if (threadIdx.x == 0) out[row] = s;
    }
}
// y[b][0][p] = max_c x[b][c][p] (first maximal channel -> idx), y[b][1][p] = mean_c x[b][c][p]
__global__ void chan_maxmean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx, int C,
                                        long HW, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i - b * HW;
        const float* xp = x + b * C * HW + p;
        float mx = xp[0], sum = xp[0];
        int am = 0;
        for (int c = 1; c < C; ++c) {
            const float v = xp[(long)c * HW];
            sum += v;
            if (v > mx) { mx = v; am = c; }
        }
        y[b * 2 * HW + p] = mx;
        y[b * 2 * HW + HW + p] = sum / (float)C;
        idx[i] = am;
    }
}
__global__ void chan_maxmean_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, float* __restrict__ dx,
                                        int C, long HW, long total) {
    const float inv_c = 1.f / (float)C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bc = i / HW, p = i - bc * HW;
        const long b = bc / C;
        const int c = (int)(bc - b * C);
        const float gmax = dy[b * 2 * HW + p], gmean = dy[b * 2 * HW + HW + p];
        dx[i] = gmean * inv_c + (idx[b * HW + p] == c ? gmax : 0.f);
    }
}

// D[b][i] = sum_c a[b][c][i] * o[b][c][i]  (per-pixel channel dot) ; delta = gamma * D
__global__ __launch_bounds__(256) void chan_dot_kernel(const float* __restrict__ a, long a_bs, const float* __restrict__ o,
                                                      long o_bs, int C, int N, const float* __restrict__ gamma,
                                                      float* __restrict__ draw, float* __restrict__ delta) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* ap = a + (long)b * a_bs + i;
    const float* op = o + (long)b * o_bs + i;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(ap[(long)c * N], op[(long)c * N], s);
    draw[(long)b * N + i] = s;
    delta[(long)b * N + i] = s * (*gamma);
}

// ---- power-of-two scale of the PAM backward's fp16 operands ----------------------------------------------------
// IEEE fp16 flushes |v| < 6e-8 and loses precision below 6e-5; gamma * dOut of a real training step sits far below that
// (mean losses over 1e6..1e7 pixels).  Every output of the backward is linear in dOut, so dOut goes in as
// gamma * 2^k * dOut with max |.| in [0.5, 1) and the consumers of dQ / dK / dV multiply by 2^-k (their alpha).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ a, long a_bs, long chw, float* __restrict__ ws) {
    __shared__ float red[8];
    const float* ap = a + (long)blockIdx.y * a_bs;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chw; i += (long)gridDim.x * 256) {
        const float v = fabsf(ap[i]);
        m = v > m ? v : m;                       // NaN never enters (comparison false): an all-NaN gradient scales by 1
    }
    m = gd_block_max(m, red);
    if (threadIdx.x == 0) ws[blockIdx.y * gridDim.x + blockIdx.x] = m;
}
// scales[0] = gamma * 2^k, scales[1] = 2^-k with 2^k * |gamma| * amax in [0.5, 1); delta *= 2^k
__global__ __launch_bounds__(256) void f16_scale_kernel(const float* __restrict__ ws, int nparts, const float* __restrict__ gamma,
                                                       float* __restrict__ delta, long n, float* __restrict__ scales) {
    __shared__ float red[8];
    __shared__ float s_up;
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, ws[i]);
    m = gd_block_max(m, red);
    if (threadIdx.x == 0) {
        const float g = *gamma;
        const float t = m * fabsf(g);
        int e = 0;
        if (t > 0.f && t < 3.0e38f) frexpf(t, &e);          // t = f * 2^e, f in [0.5, 1)
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        s_up = ldexpf(1.f, -e);
        if (blockIdx.x == 0) {
            scales[0] = g * s_up;
            scales[1] = ldexpf(1.f, e);
        }
    }
    __syncthreads();
    const float up = s_up;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) delta[i] *= up;
}

// ---- losses --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, long n, float label, float inv_n,
                                                 float* __restrict__ dz, float* __restrict__ ws) {
    __shared__ float red[8];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = z[i];
        s += fmaxf(v, 0.f) - v * label + log1pf(expf(-fabsf(v)));
        if (dz) dz[i] = (1.f / (1.f + expf(-v)) - label) * inv_n;
    }
    s = gd_block_sum(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
// BCE with logits against a per-element target tensor (torch.nn.BCEWithLogitsLoss, mean); dt: gradient w.r.t. target
__global__ __launch_bounds__(256) void bce_target_kernel(const float* __restrict__ z, const float* __restrict__ t, long n,
                                                        float inv_n, float* __restrict__ dz, float* __restrict__ dt,
                                                        float* __restrict__ ws) {
    __shared__ float red[8];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = z[i], y = t[i];
        s += fmaxf(v, 0.f) - v * y + log1pf(expf(-fabsf(v)));
        if (dz) dz[i] = (1.f / (1.f + expf(-v)) - y) * inv_n;
        if (dt) dt[i] = -v * inv_n;
    }
    s = gd_block_sum(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
// LeakyReLU with an arbitrary slope (the fused epilogues know 0.2 only: discriminator.py:62-77 uses nothing else)
__global__ void leaky_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v >= 0.f ? v : slope * v;
    }
}
__global__ void leaky_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n,
                                 float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dx[i] = x[i] >= 0.f ? dy[i] : slope * dy[i];     // on the INPUT's sign: valid for every slope, 0 and negative too
}
template <bool L1>
__global__ __launch_bounds__(256) void diff_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                       float inv_n, float* __restrict__ da, float* __restrict__ ws) {
    __shared__ float red[8];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float dv = a[i] - b[i];
        if (L1) {
            s += fabsf(dv);
            if (da) da[i] = dv > 0.f ? inv_n : (dv < 0.f ? -inv_n : 0.f);
        } else {
            s = fmaf(dv, dv, s);
            if (da) da[i] = 2.f * dv * inv_n;
        }
    }
    s = gd_block_sum(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
// TV: sum of squared vertical / horizontal neighbour differences; gradient of
//   w*2*(h_tv/count_h + w_tv/count_w)/B  written per pixel (gather of the <=4 differences it appears in)
__global__ __launch_bounds__(256) void tv_kernel(const float* __restrict__ x, long planes, int H, int W, float kh, float kw,
                                                float* __restrict__ dx, float* __restrict__ ws_h, float* __restrict__ ws_w) {
    __shared__ float red[8];
    const long total = planes * H * W;
    float sh = 0.f, sw = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % W);
        const int yy = (int)((i / W) % H);
        const float v = x[i];
        float g = 0.f;
        if (yy + 1 < H) { const float d = x[i + W] - v; sh = fmaf(d, d, sh); g -= 2.f * kh * d; }
        if (yy > 0) { const float d = v - x[i - W]; g += 2.f * kh * d; }
        if (xx + 1 < W) { const float d = x[i + 1] - v; sw = fmaf(d, d, sw); g -= 2.f * kw * d; }
        if (xx > 0) { const float d = v - x[i - 1]; g += 2.f * kw * d; }
        if (dx) dx[i] = g;
    }
    sh = gd_block_sum(sh, red);
    sw = gd_block_sum(sw, red);
    if (threadIdx.x == 0) { ws_h[blockIdx.x] = sh; ws_w[blockIdx.x] = sw; }
}
__global__ __launch_bounds__(256) void tv_final_kernel(const float* __restrict__ ws_h, const float* __restrict__ ws_w,
                                                      int nparts, float kh, float kw, float* __restrict__ out) {
    __shared__ double redd[8];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) { a += ws_h[i]; b += ws_w[i]; }
    a = gd_wave_sum_d(a);
    b = gd_wave_sum_d(b);
    if ((threadIdx.x & 63) == 0) { redd[threadIdx.x >> 6] = a; redd[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    if (threadIdx.x == 0)
        out[0] = (float)((redd[0] + redd[1] + redd[2] + redd[3]) * (double)kh + (redd[4] + redd[5] + redd[6] + redd[7]) * (double)kw);
}

struct SsimWin { float g[16]; int n; };
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                  SsimWin win, float* __restrict__ ws) {
    __shared__ float red[8];
    const long plane = blockIdx.y;
    const float* pa = a + plane * (long)H * W;
    const float* pb = b + plane * (long)H * W;
    const int half = win.n / 2;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)H * W; i += (long)gridDim.x * 256) {
        const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
        float m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int dy = 0; dy < win.n; ++dy) {
            const int y2 = yy + dy - half;
            if (y2 < 0 || y2 >= H) continue;
            for (int dx = 0; dx < win.n; ++dx) {
                const int x2 = xx + dx - half;
                if (x2 < 0 || x2 >= W) continue;
                const float w = win.g[dy] * win.g[dx];
                const float u = pa[(long)y2 * W + x2], v = pb[(long)y2 * W + x2];
                m1 = fmaf(w, u, m1); m2 = fmaf(w, v, m2);
                s11 = fmaf(w, u * u, s11); s22 = fmaf(w, v * v, s22); s12 = fmaf(w, u * v, s12);
            }
        }
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
        const float v1 = s11 - m1s, v2 = s22 - m2s, v12 = s12 - m12;
        s += ((2.f * m12 + c1) * (2.f * v12 + c2)) / ((m1s + m2s + c1) * (v1 + v2 + c2));
    }
    s = gd_block_sum(s, red);
    if (threadIdx.x == 0) ws[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// SSIM backward (losses.py:118-136 under autograd), two passes over the image.
// With m1 = G*a, m2 = G*b, e11 = G*a^2, e22 = G*b^2, e12 = G*ab (G = the zero-padded Gaussian window, symmetric, so the
// adjoint of each filtering is the same filtering) and A1 = 2 m1 m2 + c1, A2 = 2 (e12 - m1 m2) + c2,
// B1 = m1^2 + m2^2 + c1, B2 = e11 - m1^2 + e22 - m2^2 + c2, ssim = A1 A2 / (B1 B2):
//   d/dm1 = 2 m2 (A2 - A1) / (B1 B2) - 2 m1 ssim (1/B1 - 1/B2)      d/dm2: m1 <-> m2
//   d/de12 = 2 A1 / (B1 B2)                                          d/de11 = d/de22 = -ssim / B2
// pass 1 writes the four coefficient planes (times the upstream factor of the sample), pass 2 filters them:
//   da = G*P1 + 2 a (G*P3) + b (G*P4),   db = G*P2 + 2 b (G*P3) + a (G*P4)
__global__ __launch_bounds__(256) void ssim_bwd_coef_kernel(const float* __restrict__ a, const float* __restrict__ b, int C,
                                                           int H, int W, SsimWin win, const float* __restrict__ gscale,
                                                           float* __restrict__ coef) {
    const long plane = blockIdx.y, hw = (long)H * W;
    const float* pa = a + plane * hw;
    const float* pb = b + plane * hw;
    const float g = gscale[plane / C];
    float* pc = coef + plane * 4 * hw;
    const int half = win.n / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
        float m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int dy = 0; dy < win.n; ++dy) {
            const int y2 = yy + dy - half;
            if (y2 < 0 || y2 >= H) continue;
            for (int dx = 0; dx < win.n; ++dx) {
                const int x2 = xx + dx - half;
                if (x2 < 0 || x2 >= W) continue;
                const float w = win.g[dy] * win.g[dx];
                const float u = pa[(long)y2 * W + x2], v = pb[(long)y2 * W + x2];
                m1 = fmaf(w, u, m1); m2 = fmaf(w, v, m2);
                s11 = fmaf(w, u * u, s11); s22 = fmaf(w, v * v, s22); s12 = fmaf(w, u * v, s12);
            }
        }
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
        const float A1 = 2.f * m12 + c1, A2 = 2.f * (s12 - m12) + c2;
        const float B1 = m1s + m2s + c1, B2 = (s11 - m1s) + (s22 - m2s) + c2;
        const float inv = 1.f / (B1 * B2), ss = A1 * A2 * inv;
        const float t = 2.f * (A2 - A1) * inv, u2 = 2.f * ss * (1.f / B1 - 1.f / B2);
        pc[i] = g * (m2 * t - m1 * u2);
        pc[hw + i] = g * (m1 * t - m2 * u2);
        pc[2 * hw + i] = g * (-ss / B2);
        pc[3 * hw + i] = g * (2.f * A1 * inv);
    }
}
__global__ __launch_bounds__(256) void ssim_bwd_filter_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             int H, int W, SsimWin win, const float* __restrict__ coef,
                                                             float* __restrict__ da, float* __restrict__ db) {
    const long plane = blockIdx.y, hw = (long)H * W;
    const float* pc = coef + plane * 4 * hw;
    const int half = win.n / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
        float f1 = 0, f2 = 0, f3 = 0, f4 = 0;
        for (int dy = 0; dy < win.n; ++dy) {
            const int y2 = yy + dy - half;
            if (y2 < 0 || y2 >= H) continue;
            for (int dx = 0; dx < win.n; ++dx) {
                const int x2 = xx + dx - half;
                if (x2 < 0 || x2 >= W) continue;
                const float w = win.g[dy] * win.g[dx];
                const long j = (long)y2 * W + x2;
                f1 = fmaf(w, pc[j], f1); f2 = fmaf(w, pc[hw + j], f2);
                f3 = fmaf(w, pc[2 * hw + j], f3); f4 = fmaf(w, pc[3 * hw + j], f4);
            }
        }
        const float u = a[plane * hw + i], v = b[plane * hw + i];
        if (da) da[plane * hw + i] = f1 + 2.f * u * f3 + v * f4;
        if (db) db[plane * hw + i] = f2 + 2.f * v * f3 + u * f4;
    }
}
// out[s] = scale * sum of row s of ws (rows x n partial sums)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ ws, int n, float scale,
                                                         float* __restrict__ out) {
    __shared__ double redd[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += ws[(long)blockIdx.x * n + i];
    acc = gd_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)((redd[0] + redd[1] + redd[2] + redd[3]) * (double)scale);
}

// ---- inference post-processing (test.ipynb c1:87-101 smooth_blend) ------------------------------------------------
// gen[b][c][sr + y][sc + x] = gen * (1 - mask[y][x]) + grace * mask[y][x] over the region, in place
__global__ void blend_region_kernel(float* __restrict__ gen, const float* __restrict__ grace, const float* __restrict__ mask,
                                    int H, int W, int sr, int sc, int rh, int rw, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long plane = i / ((long)rh * rw);
        const int rem = (int)(i - plane * (long)rh * rw);
        const int y = rem / rw, x = rem - y * rw;
        const long o = plane * (long)H * W + (long)(sr + y) * W + (sc + x);
        const float m = mask[rem];
        gen[o] = gen[o] * (1.f - m) + grace[o] * m;
    }
}

// ---- AdamW -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float beta1, float beta2, float eps, float wd, float gscale,
                                                   float inv_bc1, float inv_sqrt_bc2) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        pi -= (lr * inv_bc1) * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

}  // namespace

#define GD_S ((hipStream_t)stream)

extern "C" int gd_act_fwd(const float* x, float* y, long n, int act, void* stream) {
    GD_CHECK_ARG(x && y && n > 0, "gd_act_fwd: bad arguments");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, GD_S, x, y, n, act);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_act_bwd(const float* y, const float* dy, float* dx, long n, int act, void* stream) {
    GD_CHECK_ARG(y && dy && dx && n > 0, "gd_act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, GD_S, y, dy, dx, n, act);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_axpby(const float* x, float a, float* y, float b, long n, void* stream) {
    GD_CHECK_ARG(x && y && n > 0, "gd_axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, GD_S, x, a, y, b, n);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_scale_dev(const float* x, const float* s_dev, float* y, long n, int accumulate, void* stream) {
    GD_CHECK_ARG(x && y && s_dev && n > 0, "gd_scale_dev: bad arguments");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, GD_S, x, s_dev, y, n, accumulate);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_copy_rows(const float* src, long s_bs, long s_ld, float* dst, long d_bs, long d_ld, int B, int R, int Cc,
                            void* stream) {
    GD_CHECK_ARG(src && dst && B > 0 && B <= 65535 && R > 0 && Cc > 0 && s_ld >= Cc && d_ld >= Cc, "gd_copy_rows: bad arguments");
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((long)R * Cc), B), dim3(256), 0, GD_S, src, s_bs, s_ld, dst, d_bs,
                       d_ld, R, Cc);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_copy_slab(const float* src, long s_bs, float* dst, long d_bs, int B, long chw, int accumulate,
                            void* stream) {
    GD_CHECK_ARG(src && dst && B > 0 && B <= 65535 && chw > 0, "gd_copy_slab: bad arguments");
    if (chw % 4 == 0 && s_bs % 4 == 0 && d_bs % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0)
        hipLaunchKernelGGL(copy_slab_vec_kernel, dim3(grid_for(chw / 4), B), dim3(256), 0, GD_S, (const float4*)src, s_bs / 4,
                           (float4*)dst, d_bs / 4, chw / 4, accumulate);
    else
        hipLaunchKernelGGL(copy_slab_kernel, dim3(grid_for(chw), B), dim3(256), 0, GD_S, src, s_bs, dst, d_bs, chw, accumulate);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_softmax_rows(const float* x, float* y, long rows, int cols, float sign, void* stream) {
    GD_CHECK_ARG(x && y && rows > 0 && rows < (1L << 31) && cols > 0, "gd_softmax_rows: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, GD_S, x, y, cols, sign);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_softmax_rows_bwd(const float* p, const float* dp, float* dx, long rows, int cols, float sign,
                                   void* stream) {
    GD_CHECK_ARG(p && dp && dx && rows > 0 && rows < (1L << 31) && cols > 0, "gd_softmax_rows_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, GD_S, p, dp, dx, cols, sign);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_dot(const float* a, const float* b, long n, float* out, int accumulate, float* ws, void* stream) {
    GD_CHECK_ARG(a && out && ws && n > 0, "gd_dot: bad arguments");
    const int g = grid_for(n) > RED_BLOCKS ? RED_BLOCKS : grid_for(n);
    hipLaunchKernelGGL(dot_kernel, dim3(g), dim3(256), 0, GD_S, a, b, n, ws);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, GD_S, ws, g, 1.f, out, accumulate);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_transpose(const float* s, float* t, int B, int R, int Cc, void* stream) {
    GD_CHECK_ARG(s && t && B > 0 && B <= 65535 && R > 0 && Cc > 0, "gd_transpose: bad arguments");
    GD_CHECK_ARG(gd_cdiv(R, 32) <= 65535, "gd_transpose: too many rows");
    hipLaunchKernelGGL(transpose_kernel, dim3(gd_cdiv(Cc, 32), gd_cdiv(R, 32), B), dim3(256), 0, GD_S, s, t, R, Cc);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_add_transpose(const float* a, float* out, int B, int n, void* stream) {
    GD_CHECK_ARG(a && out && a != out && B > 0 && B <= 65535 && n > 0, "gd_add_transpose: bad arguments");
    hipLaunchKernelGGL(add_transpose_kernel, dim3(grid_for((long)n * n), B), dim3(256), 0, GD_S, a, out, n);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_pack_16(const float* s, long s_bs, int B, int R, int Cc, const float* scale_dev, float scale_imm,
                          void* plain, int Rp_plain, int ld_plain, void* transposed, int Ccp_t, int ld_t, int perm16,
                          int ones_row, int f16, void* stream) {
    GD_CHECK_ARG(s && (plain || transposed) && B > 0 && B <= 65535 && R > 0 && Cc > 0, "gd_pack_16: bad arguments");
    GD_CHECK_ARG(!plain || (Rp_plain >= R && ld_plain >= Cc), "gd_pack_16: plain padding smaller than the data");
    GD_CHECK_ARG(!perm16 || (plain && ld_plain % 16 == 0), "gd_pack_16: perm16 needs a plain output with ld % 16 == 0");
    GD_CHECK_ARG(!transposed || (Ccp_t >= Cc && ld_t >= R), "gd_pack_16: transposed padding smaller than the data");
    GD_CHECK_ARG(ones_row < 0 || (ones_row >= R && (!plain || ones_row < Rp_plain) && (!transposed || ones_row < ld_t)),
                 "gd_pack_16: ones_row must be a padding row");
    int rows = R, cols = Cc;
    if (plain) { rows = rows > Rp_plain ? rows : Rp_plain; cols = cols > ld_plain ? cols : ld_plain; }
    if (transposed) { rows = rows > ld_t ? rows : ld_t; cols = cols > Ccp_t ? cols : Ccp_t; }
    GD_CHECK_ARG(gd_cdiv(rows, 32) <= 65535, "gd_pack_16: too many rows");
    const bool aligned = Cc % 4 == 0 && s_bs % 4 == 0 && ((uintptr_t)s % 16) == 0 && (!plain || (ld_plain % 8 == 0 && (uintptr_t)plain % 16 == 0)) &&
                         (!transposed || (ld_t % 8 == 0 && (uintptr_t)transposed % 16 == 0));
    if (aligned)
        hipLaunchKernelGGL(pack16_tile64_kernel, dim3(gd_cdiv(cols, 64), gd_cdiv(rows, 64), B), dim3(256), 0, GD_S, s, s_bs, R, Cc,
                           scale_dev, scale_imm, (const float*)nullptr, (const float*)nullptr, 0, (unsigned short*)plain, Rp_plain,
                           ld_plain, (unsigned short*)transposed, Ccp_t, ld_t, perm16, ones_row, f16);
    else
        hipLaunchKernelGGL(pack_bf16_kernel, dim3(gd_cdiv(cols, 32), gd_cdiv(rows, 32), B), dim3(256), 0, GD_S, s, s_bs, R, Cc,
                           scale_dev, scale_imm, (unsigned short*)plain, Rp_plain, ld_plain, (unsigned short*)transposed, Ccp_t,
                           ld_t, perm16, ones_row, f16);
    GD_LAUNCH_CHECK();
    return 0;
}
// pack with a per-row affine + ReLU applied first (row r: max(0, row_scale[r] * x + row_shift[r]) when relu, R entries each):
// the BatchNorm + ReLU prologue of a dense layer, for the pixel-major bf16 copy its weight gradient reads.  Aligned case only
// (Cc % 4 == 0, leading dimensions % 8 == 0).
extern "C" int gd_pack_16_affine(const float* s, long s_bs, int B, int R, int Cc, const float* row_scale, const float* row_shift,
                                 int relu, void* plain, int Rp_plain, int ld_plain, void* transposed, int Ccp_t, int ld_t,
                                 int f16, void* stream) {
    GD_CHECK_ARG(s && (plain || transposed) && B > 0 && B <= 65535 && R > 0 && Cc > 0, "gd_pack_16_affine: bad arguments");
    GD_CHECK_ARG((row_scale == nullptr) == (row_shift == nullptr), "gd_pack_16_affine: row_scale / row_shift come together");
    GD_CHECK_ARG(!plain || (Rp_plain >= R && ld_plain >= Cc), "gd_pack_16_affine: plain padding smaller than the data");
    GD_CHECK_ARG(!transposed || (Ccp_t >= Cc && ld_t >= R), "gd_pack_16_affine: transposed padding smaller than the data");
    GD_CHECK_ARG(Cc % 4 == 0 && s_bs % 4 == 0 && ((uintptr_t)s % 16) == 0 && (!plain || (ld_plain % 8 == 0 && (uintptr_t)plain % 16 == 0)) &&
                     (!transposed || (ld_t % 8 == 0 && (uintptr_t)transposed % 16 == 0)),
                 "gd_pack_16_affine: needs Cc % 4 == 0, leading dimensions % 8 == 0 and 16-byte aligned pointers");
    int rows = R, cols = Cc;
    if (plain) { rows = rows > Rp_plain ? rows : Rp_plain; cols = cols > ld_plain ? cols : ld_plain; }
    if (transposed) { rows = rows > ld_t ? rows : ld_t; cols = cols > Ccp_t ? cols : Ccp_t; }
    GD_CHECK_ARG(gd_cdiv(rows, 64) <= 65535, "gd_pack_16_affine: too many rows");
    hipLaunchKernelGGL(pack16_tile64_kernel, dim3(gd_cdiv(cols, 64), gd_cdiv(rows, 64), B), dim3(256), 0, GD_S, s, s_bs, R, Cc,
                       (const float*)nullptr, 1.f, row_scale, row_shift, relu, (unsigned short*)plain, Rp_plain, ld_plain,
                       (unsigned short*)transposed, Ccp_t, ld_t, 0, -1, f16);
    GD_LAUNCH_CHECK();
    return 0;
}
// split-bf16 pack (see pack16_split_kernel).  plain: copies of the (R, Cc) tile as rows of length ldp (channel-major),
// tr: copies of its transpose as rows of length ldt (pixel-major); image b of copy j at ptr + j * cs + b * bs; copy j
// holds the lo part when bit j of the pattern is set, else the hi part.  Pad columns beyond the data are NOT written.
// mask (optional, (B, R, Cc) fp32 with batch stride m_bs): the ReLU backward fused into the pack -- elements whose mask value is
// not positive are packed as zero (dY of a conv whose output went through ReLU, masked by that output)
extern "C" int gd_pack_16_split_masked(const float* s, long s_bs, int B, int R, int Cc, const float* row_scale, const float* row_shift,
                                       int relu, void* plain, long p_bs, int ldp, long p_cs, int p_ncopy, int p_pattern, void* tr,
                                       long t_bs, int ldt, long t_cs, int t_ncopy, int t_pattern, const float* mask, long m_bs,
                                       void* stream) {
    GD_CHECK_ARG(s && (plain || tr) && B > 0 && B <= 65535 && R > 0 && Cc > 0, "gd_pack_16_split: bad arguments");
    GD_CHECK_ARG((row_scale == nullptr) == (row_shift == nullptr), "gd_pack_16_split: row_scale / row_shift come together");
    GD_CHECK_ARG(Cc % 8 == 0 && R % 8 == 0 && s_bs % 4 == 0 && ((uintptr_t)s % 16) == 0, "gd_pack_16_split: R, Cc must be multiples of 8");
    GD_CHECK_ARG(!plain || (ldp == Cc && p_ncopy >= 1 && p_ncopy <= 3 && p_cs % 8 == 0 && p_bs % 8 == 0 && (uintptr_t)plain % 16 == 0),
                 "gd_pack_16_split: plain output needs ldp == Cc, 1..3 copies, 16-byte aligned strides");
    GD_CHECK_ARG(!tr || (ldt >= R && ldt % 8 == 0 && t_ncopy >= 1 && t_ncopy <= 3 && t_cs % 8 == 0 && t_bs % 8 == 0 && (uintptr_t)tr % 16 == 0),
                 "gd_pack_16_split: transposed output needs ldt >= R, ldt % 8 == 0, 1..3 copies, 16-byte aligned strides");
    GD_CHECK_ARG(gd_cdiv(R, 64) <= 65535, "gd_pack_16_split: too many rows");
    GD_CHECK_ARG(!mask || (m_bs % 4 == 0 && ((uintptr_t)mask % 16) == 0), "gd_pack_16_split: mask must be 16-byte aligned");
    hipLaunchKernelGGL(pack16_split_kernel, dim3(gd_cdiv(Cc, 64), gd_cdiv(R, 64), B), dim3(256), 0, GD_S, s, s_bs, R, Cc, row_scale,
                       row_shift, relu, (unsigned short*)plain, p_bs, ldp, p_cs, p_ncopy, p_pattern, (unsigned short*)tr, t_bs, ldt,
                       t_cs, t_ncopy, t_pattern, mask, m_bs);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_pack_16_split(const float* s, long s_bs, int B, int R, int Cc, const float* row_scale, const float* row_shift,
                                int relu, void* plain, long p_bs, int ldp, long p_cs, int p_ncopy, int p_pattern, void* tr,
                                long t_bs, int ldt, long t_cs, int t_ncopy, int t_pattern, void* stream) {
    return gd_pack_16_split_masked(s, s_bs, B, R, Cc, row_scale, row_shift, relu, plain, p_bs, ldp, p_cs, p_ncopy, p_pattern, tr,
                                   t_bs, ldt, t_cs, t_ncopy, t_pattern, nullptr, 0, stream);
}
extern "C" int gd_split3_weights(const float* w, long A, long Bn, long Cn, float* out, void* stream) {
    GD_CHECK_ARG(w && out && A > 0 && Bn > 0 && Cn > 0, "gd_split3_weights: bad arguments");
    const long total = A * Bn * Cn;
    hipLaunchKernelGGL(split3_weights_kernel, dim3(grid_for(total) > 4096 ? 4096 : grid_for(total)), dim3(256), 0, GD_S, w, A, Bn, Cn, out);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_pack_bf16(const float* s, long s_bs, int B, int R, int Cc, const float* scale_dev, float scale_imm,
                            void* plain, int Rp_plain, int ld_plain, void* transposed, int Ccp_t, int ld_t, int perm16,
                            int ones_row, void* stream) {
    return gd_pack_16(s, s_bs, B, R, Cc, scale_dev, scale_imm, plain, Rp_plain, ld_plain, transposed, Ccp_t, ld_t, perm16,
                      ones_row, 0, stream);
}
extern "C" int gd_augment_d4(const float* src, float* dst, int B, int C, int H, int W, const int* ops, const float* noise,
                             float noise_scale, void* stream) {
    GD_CHECK_ARG(src && dst && ops && src != dst && B > 0 && C > 0 && H > 0 && W > 0, "gd_augment_d4: bad arguments");
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(augment_d4_kernel, dim3(grid_for(total)), dim3(256), 0, GD_S, src, dst, C, H, W, ops, noise,
                       noise_scale, total);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_bcast_mul(const float* x, const float* att, float* y, int B, int C, long HW, int mode, void* stream) {
    GD_CHECK_ARG(x && att && y && B > 0 && C > 0 && HW > 0 && (mode == 0 || mode == 1), "gd_bcast_mul: bad arguments");
    const long total = (long)B * C * HW;
    hipLaunchKernelGGL(bcast_mul_kernel, dim3(grid_for(total)), dim3(256), 0, GD_S, x, att, y, C, HW, mode, total);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_row_dot(const float* a, const float* b, float* out, long rows, long n, void* stream) {
    GD_CHECK_ARG(a && b && out && rows > 0 && n > 0, "gd_row_dot: bad arguments");
    hipLaunchKernelGGL(row_dot_kernel, dim3((unsigned)(rows < 8192 ? rows : 8192)), dim3(256), 0, GD_S, a, b, out, rows, n);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_chan_maxmean_fwd(const float* x, float* y, int* idx, int B, int C, long HW, void* stream) {
    GD_CHECK_ARG(x && y && idx && B > 0 && C > 0 && HW > 0, "gd_chan_maxmean_fwd: bad arguments");
    const long total = (long)B * HW;
    hipLaunchKernelGGL(chan_maxmean_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, GD_S, x, y, idx, C, HW, total);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_chan_maxmean_bwd(const float* dy, const int* idx, float* dx, int B, int C, long HW, void* stream) {
    GD_CHECK_ARG(dy && idx && dx && B > 0 && C > 0 && HW > 0, "gd_chan_maxmean_bwd: bad arguments");
    const long total = (long)B * C * HW;
    hipLaunchKernelGGL(chan_maxmean_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, GD_S, dy, idx, dx, C, HW, total);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_chan_dot(const float* a, long a_bs, const float* o, long o_bs, int B, int C, int N, const float* gamma,
                           float* d_raw, float* delta, void* stream) {
    GD_CHECK_ARG(a && o && gamma && d_raw && delta && B > 0 && B <= 65535 && C > 0 && N > 0, "gd_chan_dot: bad arguments");
    hipLaunchKernelGGL(chan_dot_kernel, dim3(gd_cdiv(N, 256), B), dim3(256), 0, GD_S, a, a_bs, o, o_bs, C, N, gamma, d_raw,
                       delta);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_pam_f16_scale(const float* dout, long dout_bs, int B, int C, int N, const float* gamma, float* delta,
                                float* scales, float* ws, void* stream) {
    GD_CHECK_ARG(dout && gamma && delta && scales && ws && B > 0 && B <= 65535 && C > 0 && N > 0, "gd_pam_f16_scale: bad arguments");
    const long chw = (long)C * N;
    int g = grid_for(chw);
    g = g > RED_BLOCKS / B ? RED_BLOCKS / B : g;
    g = g < 1 ? 1 : g;
    GD_CHECK_ARG((long)g * B <= RED_BLOCKS, "gd_pam_f16_scale: batch larger than the reduction workspace (1024 parts)");
    hipLaunchKernelGGL(absmax_kernel, dim3(g, B), dim3(256), 0, GD_S, dout, dout_bs, chw, ws);
    const long n = (long)B * N;
    hipLaunchKernelGGL(f16_scale_kernel, dim3(grid_for(n) > 256 ? 256 : grid_for(n)), dim3(256), 0, GD_S, ws, g * B, gamma, delta, n,
                       scales);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_bce_logits(const float* z, long n, float label, float* out, float* dz, float* ws, void* stream) {
    GD_CHECK_ARG(z && out && ws && n > 0, "gd_bce_logits: bad arguments");
    const int g = grid_for(n) > RED_BLOCKS ? RED_BLOCKS : grid_for(n);
    hipLaunchKernelGGL(bce_kernel, dim3(g), dim3(256), 0, GD_S, z, n, label, 1.f / (float)n, dz, ws);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, GD_S, ws, g, 1.f / (float)n, out, 0);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_bce_logits_target(const float* z, const float* t, long n, float* out, float* dz, float* dt, float* ws,
                                    void* stream) {
    GD_CHECK_ARG(z && t && out && ws && n > 0, "gd_bce_logits_target: bad arguments");
    const int g = grid_for(n) > RED_BLOCKS ? RED_BLOCKS : grid_for(n);
    hipLaunchKernelGGL(bce_target_kernel, dim3(g), dim3(256), 0, GD_S, z, t, n, 1.f / (float)n, dz, dt, ws);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, GD_S, ws, g, 1.f / (float)n, out, 0);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_leaky_fwd(const float* x, float* y, long n, float slope, void* stream) {
    GD_CHECK_ARG(x && y && n > 0, "gd_leaky_fwd: bad arguments");
    hipLaunchKernelGGL(leaky_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, GD_S, x, y, n, slope);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_leaky_bwd(const float* x, const float* dy, float* dx, long n, float slope, void* stream) {
    GD_CHECK_ARG(x && dy && dx && n > 0, "gd_leaky_bwd: bad arguments");
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, GD_S, x, dy, dx, n, slope);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_mse(const float* a, const float* b, long n, float* out, float* da, float* ws, void* stream) {
    GD_CHECK_ARG(a && b && out && ws && n > 0, "gd_mse: bad arguments");
    const int g = grid_for(n) > RED_BLOCKS ? RED_BLOCKS : grid_for(n);
    hipLaunchKernelGGL((diff_loss_kernel<false>), dim3(g), dim3(256), 0, GD_S, a, b, n, 1.f / (float)n, da, ws);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, GD_S, ws, g, 1.f / (float)n, out, 0);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_l1(const float* a, const float* b, long n, float* out, float* da, float* ws, void* stream) {
    GD_CHECK_ARG(a && b && out && ws && n > 0, "gd_l1: bad arguments");
    const int g = grid_for(n) > RED_BLOCKS ? RED_BLOCKS : grid_for(n);
    hipLaunchKernelGGL((diff_loss_kernel<true>), dim3(g), dim3(256), 0, GD_S, a, b, n, 1.f / (float)n, da, ws);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, GD_S, ws, g, 1.f / (float)n, out, 0);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_tv(const float* x, int B, int C, int H, int W, float weight, float* out, float* dx, float* ws,
                     void* stream) {
    GD_CHECK_ARG(x && out && ws && B > 0 && C > 0 && H > 1 && W > 1, "gd_tv: bad arguments");
    const long planes = (long)B * C;
    const double count_h = (double)planes * (H - 1) * W, count_w = (double)planes * H * (W - 1);
    const float kh = (float)(weight * 2.0 / count_h / B), kw = (float)(weight * 2.0 / count_w / B);
    const long n = planes * H * W;
    const int g = grid_for(n) > RED_BLOCKS ? RED_BLOCKS : grid_for(n);
    hipLaunchKernelGGL(tv_kernel, dim3(g), dim3(256), 0, GD_S, x, planes, H, W, kh, kw, dx, ws, ws + RED_BLOCKS);
    hipLaunchKernelGGL(tv_final_kernel, dim3(1), dim3(256), 0, GD_S, ws, ws + RED_BLOCKS, g, kh, kw, out);
    GD_LAUNCH_CHECK();
    return 0;
}
static int ssim_window(int window, SsimWin* win) {
    if (!(window > 0 && window <= 15 && (window & 1))) return -1;
    win->n = window;
    // SSIM._gaussian (losses.py:98-101): fp32 exp, normalised by the fp32 sum
    float tmp[16], sum = 0.f;
    for (int i = 0; i < window; ++i) {
        const float d = (float)(i - window / 2);
        tmp[i] = expf(-(d * d) / (2.f * 1.5f * 1.5f));
        sum += tmp[i];
    }
    for (int i = 0; i < 16; ++i) win->g[i] = i < window ? tmp[i] / sum : 0.f;
    return 0;
}
static int ssim_gx(int H, int W, int BC) {
    int gx = (int)(((long)H * W + 255) / 256);
    if (gx > 64) gx = 64;
    while ((long)gx * BC > 2048 && gx > 1) gx >>= 1;
    return gx;
}
extern "C" int gd_ssim(const float* a, const float* b, int BC, int H, int W, int window, float* out, float* ws,
                       void* stream) {
    GD_CHECK_ARG(a && b && out && ws && BC > 0 && H > 0 && W > 0, "gd_ssim: bad arguments");
    SsimWin win;
    GD_CHECK_ARG(ssim_window(window, &win) == 0, "gd_ssim: window must be odd and <= 15");
    const int gx = ssim_gx(H, W, BC);
    GD_CHECK_ARG((long)gx * BC <= 2048 && BC <= 65535, "gd_ssim: too many planes for the workspace");
    hipLaunchKernelGGL(ssim_kernel, dim3(gx, BC), dim3(256), 0, GD_S, a, b, H, W, win, ws);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, GD_S, ws, gx * BC, 1.f / ((float)BC * (float)H * (float)W),
                       out, 0);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_ssim_samples(const float* a, const float* b, int B, int C, int H, int W, int window, float* out,
                               float* ws, void* stream) {
    GD_CHECK_ARG(a && b && out && ws && B > 0 && C > 0 && H > 0 && W > 0, "gd_ssim_samples: bad arguments");
    SsimWin win;
    GD_CHECK_ARG(ssim_window(window, &win) == 0, "gd_ssim_samples: window must be odd and <= 15");
    const int BC = B * C, gx = ssim_gx(H, W, BC);
    GD_CHECK_ARG((long)gx * BC <= 2048 && BC <= 65535, "gd_ssim_samples: too many planes for the workspace");
    hipLaunchKernelGGL(ssim_kernel, dim3(gx, BC), dim3(256), 0, GD_S, a, b, H, W, win, ws);
    // ws is [plane][gx]: a sample's C planes are consecutive
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(B), dim3(256), 0, GD_S, ws, gx * C, 1.f / ((float)C * (float)H * (float)W), out);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_ssim_bwd(const float* a, const float* b, const float* gscale, int B, int C, int H, int W, int window,
                           float* coef_ws, float* da, float* db, void* stream) {
    GD_CHECK_ARG(a && b && gscale && coef_ws && (da || db) && B > 0 && C > 0 && H > 0 && W > 0, "gd_ssim_bwd: bad arguments");
    SsimWin win;
    GD_CHECK_ARG(ssim_window(window, &win) == 0, "gd_ssim_bwd: window must be odd and <= 15");
    GD_CHECK_ARG((long)B * C <= 65535, "gd_ssim_bwd: too many planes");
    int gx = (int)(((long)H * W + 255) / 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(ssim_bwd_coef_kernel, dim3(gx, B * C), dim3(256), 0, GD_S, a, b, C, H, W, win, gscale, coef_ws);
    hipLaunchKernelGGL(ssim_bwd_filter_kernel, dim3(gx, B * C), dim3(256), 0, GD_S, a, b, H, W, win, coef_ws, da, db);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_blend_region(float* gen, const float* grace, const float* mask, int BC, int H, int W, int sr, int er,
                               int sc, int ec, void* stream) {
    GD_CHECK_ARG(gen && grace && mask && BC > 0 && H > 0 && W > 0, "gd_blend_region: bad arguments");
    GD_CHECK_ARG(0 <= sr && sr < er && er <= H && 0 <= sc && sc < ec && ec <= W, "gd_blend_region: region outside the image");
    const long total = (long)BC * (er - sr) * (ec - sc);
    hipLaunchKernelGGL(blend_region_kernel, dim3(grid_for(total)), dim3(256), 0, GD_S, gen, grace, mask, H, W, sr, sc, er - sr,
                       ec - sc, total);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_adamw(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    GD_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "gd_adamw: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, GD_S, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, grad_scale, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
    GD_LAUNCH_CHECK();
    return 0;
}
