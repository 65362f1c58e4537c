// BatchNorm2d for gfx950: batch statistics (two-level, cancellation-safe), folded affine, fused
// affine+activation apply, and the fused backward of act(bn(x)).  All HBM-bound streaming kernels:
// one workgroup per (channel, image, chunk) reads a contiguous run of one channel plane.
#include "common.h"
#include "../../include/gandanet.h"

namespace {

constexpr int CHUNK = 16384;  // elements of one channel plane per workgroup

static inline int chunks_per_image(long HW) { return (int)((HW + CHUNK - 1) / CHUNK); }

// ---- statistics ---------------------------------------------------------------------------------
// partial (n, mean, M2) per (c, part); sums are taken relative to the first element of the run so
// that E[x^2]-E[x]^2 style cancellation cannot happen even for |mean| >> std.
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, long x_bs, int C, long HW,
                                                        int cpi, float* __restrict__ ws) {
    __shared__ float red[8];
    const int c = blockIdx.x;
    const int part = blockIdx.y;
    const int b = part / cpi, j = part - b * cpi;
    const long i0 = (long)j * CHUNK;
    const long i1 = min(HW, i0 + (long)CHUNK);
    const float* p = x + (long)b * x_bs + (long)c * HW;
    const float shift = p[i0];
    float s1 = 0.f, s2 = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float v = p[i] - shift;
        s1 += v;
        s2 = fmaf(v, v, s2);
    }
    s1 = gd_block_sum(s1, red);
    s2 = gd_block_sum(s2, red);
    if (threadIdx.x == 0) {
        const float n = (float)(i1 - i0);
        float* o = ws + ((long)c * gridDim.y + part) * 3;
        o[0] = n;
        o[1] = shift + s1 / n;
        o[2] = fmaxf(s2 - s1 * s1 / n, 0.f);
    }
}

// Chan merge (fp64) of `parts` (count, mean, M2) records of a channel; record i of channel c sits at
// ws[(c * c_stride + i * p_stride) * 3] (the kernel's own partials: c_stride = parts, p_stride = 1; the all-gathered
// per-rank records of SyncBN, (world, C, 3): c_stride = 1, p_stride = C).  stats_out != NULL: the merged record is
// written there instead of mean / invstd (the local half of SyncBN).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ ws, int parts, long c_stride,
                                                         long p_stride, float eps, float momentum,
                                                         float* __restrict__ mean, float* __restrict__ invstd,
                                                         float* running_mean, float* running_var,
                                                         float* __restrict__ stats_out) {
    __shared__ double redd[4];
    const int c = blockIdx.x;
    const float* w0 = ws + (long)c * c_stride * 3;
    const long ps = p_stride * 3;
    double n = 0, sm = 0;
    for (int i = threadIdx.x; i < parts; i += 256) {
        const float* w = w0 + i * ps;
        n += w[0];
        sm += (double)w[0] * w[1];
    }
    auto bsum = [&](double v) {
        v = gd_wave_sum_d(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = v;
        __syncthreads();
        return redd[0] + redd[1] + redd[2] + redd[3];
    };
    n = bsum(n);
    sm = bsum(sm);
    const double mu = sm / n;
    double m2 = 0;
    for (int i = threadIdx.x; i < parts; i += 256) {
        const float* w = w0 + i * ps;
        const double dm = (double)w[1] - mu;
        m2 += (double)w[2] + (double)w[0] * dm * dm;
    }
    m2 = bsum(m2);
    if (threadIdx.x == 0 && stats_out) {
        stats_out[c * 3 + 0] = (float)n;
        stats_out[c * 3 + 1] = (float)mu;
        stats_out[c * 3 + 2] = (float)m2;
    } else if (threadIdx.x == 0) {
        const double var = m2 / n;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = n > 1 ? m2 / (n - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
}

// per-channel sum over (B, HW): conv / linear bias gradients
__global__ __launch_bounds__(256) void chan_sum_partial_kernel(const float* __restrict__ x, long x_bs, long HW, int cpi,
                                                              float* __restrict__ ws) {
    __shared__ float red[8];
    const int c = blockIdx.x;
    const int part = blockIdx.y;
    const int b = part / cpi, j = part - b * cpi;
    const long i0 = (long)j * CHUNK, i1 = min(HW, i0 + (long)CHUNK);
    const float* p = x + (long)b * x_bs + (long)c * HW;
    float s = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) s += p[i];
    s = gd_block_sum(s, red);
    if (threadIdx.x == 0) ws[(long)c * gridDim.y + part] = s;
}
__global__ __launch_bounds__(256) void chan_sum_final_kernel(const float* __restrict__ ws, int parts,
                                                            float* __restrict__ out, int accumulate) {
    __shared__ double redd[4];
    const int c = blockIdx.x;
    double a = 0;
    for (int i = threadIdx.x; i < parts; i += 256) a += ws[(long)c * parts + i];
    a = gd_wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(redd[0] + redd[1] + redd[2] + redd[3]);
        out[c] = accumulate ? out[c] + v : v;
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* invstd,
                               const float* running_var, float eps, int C, float* scale, float* shift,
                               float* invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = invstd ? invstd[c] : 1.0f / sqrtf(running_var[c] + eps);
    if (invstd_out) invstd_out[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - mean[c] * sc;
}

// ---- y = act(x*scale + shift) -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, long x_bs,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, long HW, int cpi, int act,
                                                        float* __restrict__ y, long y_bs) {
    const int c = blockIdx.x;
    const int b = blockIdx.y / cpi, j = blockIdx.y - b * cpi;
    const long i0 = (long)j * CHUNK, i1 = min(HW, i0 + (long)CHUNK);
    const float* p = x + (long)b * x_bs + (long)c * HW;
    float* q = y + (long)b * y_bs + (long)c * HW;
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    if ((HW & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)q)) & 15) == 0) {
        for (long i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
            float4 v = *reinterpret_cast<const float4*>(p + i);
            v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
            if (act == GD_ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (act == GD_ACT_LEAKY02) {
                v.x = v.x >= 0 ? v.x : 0.2f * v.x; v.y = v.y >= 0 ? v.y : 0.2f * v.y;
                v.z = v.z >= 0 ? v.z : 0.2f * v.z; v.w = v.w >= 0 ? v.w : 0.2f * v.w;
            }
            *reinterpret_cast<float4*>(q + i) = v;
        }
    } else {
        for (long i = i0 + threadIdx.x; i < i1; i += 256) {
            float v = fmaf(p[i], sc, sh);
            if (act == GD_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == GD_ACT_LEAKY02) v = v >= 0 ? v : 0.2f * v;
            q[i] = v;
        }
    }
}

// ---- backward of y = act(x*scale + shift) with batch statistics ----------------------------------------
__device__ __forceinline__ float act_grad(float ypre, float dy, int act) {
    if (act == GD_ACT_RELU) return ypre > 0.f ? dy : 0.f;
    if (act == GD_ACT_LEAKY02) return ypre >= 0.f ? dy : 0.2f * dy;
    return dy;
}

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, long dy_bs,
                                                            const float* __restrict__ x, long x_bs,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, long HW, int cpi, int act,
                                                            float* __restrict__ ws) {
    __shared__ float red[8];
    const int c = blockIdx.x;
    const int part = blockIdx.y;
    const int b = part / cpi, j = part - b * cpi;
    const long i0 = (long)j * CHUNK, i1 = min(HW, i0 + (long)CHUNK);
    const float* px = x + (long)b * x_bs + (long)c * HW;
    const float* pd = dy + (long)b * dy_bs + (long)c * HW;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    float s_db = 0.f, s_dg = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float xv = px[i];
        const float g = act_grad(fmaf(xv, sc, sh), pd[i], act);
        s_db += g;
        s_dg = fmaf(g, (xv - mu) * is, s_dg);
    }
    s_db = gd_block_sum(s_db, red);
    s_dg = gd_block_sum(s_dg, red);
    if (threadIdx.x == 0) {
        float* o = ws + ((long)c * gridDim.y + part) * 2;
        o[0] = s_db;
        o[1] = s_dg;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ ws, int parts,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double redd[4];
    const int c = blockIdx.x;
    const float* w = ws + (long)c * parts * 2;
    double a = 0, g = 0;
    for (int i = threadIdx.x; i < parts; i += 256) {
        a += w[i * 2];
        g += w[i * 2 + 1];
    }
    auto bsum = [&](double v) {
        v = gd_wave_sum_d(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = v;
        __syncthreads();
        return redd[0] + redd[1] + redd[2] + redd[3];
    };
    a = bsum(a);
    g = bsum(g);
    if (threadIdx.x == 0) {
        dbeta[c] = (float)a;
        dgamma[c] = (float)g;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ dy, long dy_bs,
                                                       const float* __restrict__ x, long x_bs,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                       float inv_n, long HW, int cpi, int act, int train,
                                                       float* __restrict__ dx, long dx_bs, int accumulate) {
    const int c = blockIdx.x;
    const int b = blockIdx.y / cpi, j = blockIdx.y - b * cpi;
    const long i0 = (long)j * CHUNK, i1 = min(HW, i0 + (long)CHUNK);
    const float* px = x + (long)b * x_bs + (long)c * HW;
    const float* pd = dy + (long)b * dy_bs + (long)c * HW;
    float* po = dx + (long)b * dx_bs + (long)c * HW;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const float k_db = train ? dbeta[c] * inv_n : 0.f;
    const float k_dg = train ? dgamma[c] * inv_n : 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float xv = px[i];
        const float g = act_grad(fmaf(xv, sc, sh), pd[i], act);
        const float xh = (xv - mu) * is;
        float v = sc * (g - k_db - xh * k_dg);  // sc = gamma*invstd
        if (accumulate) v += po[i];
        po[i] = v;
    }
}

}  // namespace

extern "C" size_t gd_bn_stats_ws_floats(int B, int C, long HW) {
    return (size_t)C * (size_t)B * (size_t)chunks_per_image(HW) * 3;
}

extern "C" int gd_bn_stats(const float* x, long x_bs, int B, int C, long HW, float eps, float momentum, float* mean,
                           float* invstd, float* running_mean, float* running_var, float* ws, void* stream) {
    GD_CHECK_ARG(x && mean && invstd && ws, "gd_bn_stats: null pointer");
    GD_CHECK_ARG(B > 0 && C > 0 && HW > 0, "gd_bn_stats: bad sizes");
    GD_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "gd_bn_stats: running stats must come together");
    const int cpi = chunks_per_image(HW);
    GD_CHECK_ARG((long)B * cpi <= 65535, "gd_bn_stats: too many parts");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(C, B * cpi), dim3(256), 0, s, x, x_bs, C, HW, cpi, ws);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, ws, B * cpi, (long)B * cpi, 1L, eps, momentum, mean, invstd,
                       running_mean, running_var, (float*)nullptr);
    GD_LAUNCH_CHECK();
    return 0;
}

// ---- SyncBN (optional flag; default is per-replica statistics) -------------------------------------------
// forward : every rank reduces its shard to one (count, mean, M2) record per channel (gd_bn_stats_local), the records are
//           all-gathered (2 x 3 C floats per rank: torch.distributed) and merged by the same Chan formula
//           (gd_bn_stats_merge) -> the statistics of the GLOBAL batch on every rank, running statistics included.
// backward: the per-channel sums of dy and dy * xhat (gd_bn_act_bwd with dx = NULL) are all-reduced, then
//           gd_bn_act_bwd_dx forms dx from the global sums and the global element count.
extern "C" int gd_bn_stats_local(const float* x, long x_bs, int B, int C, long HW, float* stats, float* ws, void* stream) {
    GD_CHECK_ARG(x && stats && ws && B > 0 && C > 0 && HW > 0, "gd_bn_stats_local: bad arguments");
    const int cpi = chunks_per_image(HW);
    GD_CHECK_ARG((long)B * cpi <= 65535, "gd_bn_stats_local: too many parts");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(C, B * cpi), dim3(256), 0, s, x, x_bs, C, HW, cpi, ws);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, ws, B * cpi, (long)B * cpi, 1L, 0.f, 0.f, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, stats);
    GD_LAUNCH_CHECK();
    return 0;
}
extern "C" int gd_bn_stats_merge(const float* stats_all, int world, int C, float eps, float momentum, float* mean,
                                 float* invstd, float* running_mean, float* running_var, void* stream) {
    GD_CHECK_ARG(stats_all && mean && invstd && world > 0 && C > 0, "gd_bn_stats_merge: bad arguments");
    GD_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "gd_bn_stats_merge: running stats must come together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats_all, world, 1L, (long)C, eps,
                       momentum, mean, invstd, running_mean, running_var, (float*)nullptr);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_channel_sum(const float* x, long x_bs, int B, int C, long HW, float* out, int accumulate, float* ws,
                              void* stream) {
    GD_CHECK_ARG(x && out && ws && B > 0 && C > 0 && HW > 0, "gd_channel_sum: bad arguments");
    const int cpi = chunks_per_image(HW);
    GD_CHECK_ARG((long)B * cpi <= 65535, "gd_channel_sum: too many parts");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(chan_sum_partial_kernel, dim3(C, B * cpi), dim3(256), 0, s, x, x_bs, HW, cpi, ws);
    hipLaunchKernelGGL(chan_sum_final_kernel, dim3(C), dim3(256), 0, s, ws, B * cpi, out, accumulate);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_bn_fold(const float* gamma, const float* beta, const float* mean, const float* invstd, int C,
                          float* scale, float* shift, void* stream) {
    GD_CHECK_ARG(gamma && beta && mean && invstd && scale && shift && C > 0, "gd_bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(gd_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, invstd,
                       (const float*)nullptr, 0.f, C, scale, shift, (float*)nullptr);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, float eps, int C, float* scale, float* shift,
                               float* invstd_out, void* stream) {
    GD_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0, "gd_bn_fold_eval: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(gd_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       (const float*)nullptr, running_var, eps, C, scale, shift, invstd_out);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_affine_act(const float* x, long x_bs, const float* scale, const float* shift, int B, int C, long HW,
                             int act, float* y, long y_bs, void* stream) {
    GD_CHECK_ARG(x && y && B > 0 && C > 0 && HW > 0, "gd_affine_act: bad arguments");
    const int cpi = chunks_per_image(HW);
    GD_CHECK_ARG((long)B * cpi <= 65535, "gd_affine_act: too many parts");
    hipLaunchKernelGGL(affine_act_kernel, dim3(C, B * cpi), dim3(256), 0, (hipStream_t)stream, x, x_bs, scale, shift, HW,
                       cpi, act, y, y_bs);
    GD_LAUNCH_CHECK();
    return 0;
}

// dx (+)= scale * (g - dbeta_sum * inv_n - xhat * dgamma_sum * inv_n) from EXTERNALLY reduced sums (SyncBN: the all-reduced
// dbeta / dgamma of every rank, inv_n = 1 / the global element count per channel)
extern "C" int gd_bn_act_bwd_dx(const float* dy, long dy_bs, const float* x, long x_bs, const float* scale, const float* shift,
                                const float* mean, const float* invstd, const float* dgamma_sum, const float* dbeta_sum,
                                float inv_n, int B, int C, long HW, int act, float* dx, long dx_bs, int accumulate_dx,
                                void* stream) {
    GD_CHECK_ARG(dy && x && scale && shift && mean && invstd && dgamma_sum && dbeta_sum && dx, "gd_bn_act_bwd_dx: null pointer");
    GD_CHECK_ARG(B > 0 && C > 0 && HW > 0 && inv_n > 0.f, "gd_bn_act_bwd_dx: bad sizes");
    const int cpi = chunks_per_image(HW);
    GD_CHECK_ARG((long)B * cpi <= 65535, "gd_bn_act_bwd_dx: too many parts");
    hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(C, B * cpi), dim3(256), 0, (hipStream_t)stream, dy, dy_bs, x, x_bs, scale, shift, mean,
                       invstd, dgamma_sum, dbeta_sum, inv_n, HW, cpi, act, 1, dx, dx_bs, accumulate_dx);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_bn_act_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* scale,
                             const float* shift, const float* mean, const float* invstd, const float* gamma, int B, int C,
                             long HW, int act, int train, float* dgamma, float* dbeta, float* dx, long dx_bs,
                             int accumulate_dx, float* ws, void* stream) {
    (void)gamma;
    GD_CHECK_ARG(dy && x && scale && shift && mean && invstd && dgamma && dbeta && ws, "gd_bn_act_bwd: null pointer");
    GD_CHECK_ARG(B > 0 && C > 0 && HW > 0, "gd_bn_act_bwd: bad sizes");
    const int cpi = chunks_per_image(HW);
    GD_CHECK_ARG((long)B * cpi <= 65535, "gd_bn_act_bwd: too many parts");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(C, B * cpi), dim3(256), 0, s, dy, dy_bs, x, x_bs, scale, shift, mean,
                       invstd, HW, cpi, act, ws);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, ws, B * cpi, dgamma, dbeta);
    if (dx) {
        const float inv_n = 1.0f / ((float)B * (float)HW);
        hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(C, B * cpi), dim3(256), 0, s, dy, dy_bs, x, x_bs, scale, shift, mean,
                           invstd, dgamma, dbeta, inv_n, HW, cpi, act, train, dx, dx_bs, accumulate_dx);
    }
    GD_LAUNCH_CHECK();
    return 0;
}
