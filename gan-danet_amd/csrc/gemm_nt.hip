// "NT" GEMM with a long, split reduction:  C[b][m][n] (+)= alpha * sum_k A[b][m][k] * B~[b][n][k]
//
// Both operands are contiguous along k in memory, so a wave reads 32 consecutive k of 2 rows per load
// instruction (128-byte segments) and writes them straight into the [row][k] LDS image the MFMA
// fragments want.  k = (segment s, offset kk): a segment is one image of the batch, which is how the
// convolution weight gradient (k = output pixels of all images), CAM's Gram matrix, nn.Linear forward
// and the unfused fp32 PAM products all map onto one kernel.  The reduction is split over
// blockIdx.z; partial tiles are combined with fp32 atomics (hardware global_atomic_add_f32).
#include "common.h"
#include "tile_mma.h"
#include "../../include/gandanet.h"

namespace {

using gd::TILE_BK;
using gd::TILE_BN;
constexpr int BN = TILE_BN;
constexpr int BK = TILE_BK;

__global__ void fill_zero_kernel(float* c, long c_bs, long ldc, int B, int M, int N) {
    const long total = (long)B * M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const long t = i / N;
        const int m = (int)(t % M);
        const int b = (int)(t / M);
        c[(long)b * c_bs + (long)m * ldc + n] = 0.f;
    }
}

// VEC: plain (non-im2col) operands whose rows, leading dimensions and segments are 16-byte friendly are staged
// with float4 loads (4 consecutive k per thread) and 8-byte LDS writes instead of scalar loads / 2-byte writes.
template <int BM, int MODE, bool VEC = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const gd_gemm_nt_desc d, int ktiles, int tiles_per_split,
                                                      int splits) {
    constexpr bool BF16 = MODE != gd::MODE_F32;     // 16-bit LDS images (one, or hi + lo for the split mode)
    constexpr bool X3 = MODE == gd::MODE_X3;
    constexpr int A_PL = BM * gd::TilePol<MODE>::LD, B_PL = gd::TILE_BN * gd::TilePol<MODE>::LD;
    using P = gd::TilePol<MODE>;
    using elem = typename P::elem;
    constexpr int LD = P::LD;
    constexpr int WAVES_N = gd::TileGeom<BM>::WAVES_N;
    constexpr int TM = gd::TileGeom<BM>::TM;
    constexpr int TN = gd::TileGeom<BM>::TN;
    constexpr int RA = BM / 8;   // A rows per thread per tile
    constexpr int RB = BN / 8;   // B rows per thread per tile (16)

    __shared__ __attribute__((aligned(16))) elem As[P::PLANES * BM * LD];
    __shared__ __attribute__((aligned(16))) elem Bs[P::PLANES * BN * LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int b = blockIdx.z / splits;
    const int split = blockIdx.z - b * splits;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const long K = (long)d.kseg * d.klen;

    const int kk = tid & 31;   // this thread's k inside a tile
    const int r0 = tid >> 5;   // first row; rows r0 + 8*i

    const float* a_base = d.a + (long)b * d.a_bs;
    const float* b_base = d.bm + (long)b * d.b_bs;

    // im2col row decode (rows do not change across the k loop)
    int rc_off[RB];    // c * Hi*Wi  (fits int: checked on the host)
    int rc_dydx[RB];   // (kh << 16) | kw, or -1 for an invalid row
    const int kss = d.ks * d.ks;
    if (d.im2col) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + r0 + 8 * i;
            if (n < d.N) {
                const int c = n / kss, tap = n - c * kss;
                const int kh = tap / d.ks, kw = tap - kh * d.ks;
                rc_off[i] = c;
                rc_dydx[i] = (kh << 16) | kw;
            } else {
                rc_off[i] = 0;
                rc_dydx[i] = -1;
            }
        }
    }
    const long HiWi = (long)d.Hi * d.Wi;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[RA], rb[RB];
    // vector path: thread -> (row group r0v = tid>>3, k quad kq = tid&7); rows r0v + 32*i
    constexpr int RAV = BM / 32, RBV = BN / 32;
    float4 va[VEC ? RAV : 1], vb[VEC ? RBV : 1];
    const int kq = tid & 7, r0v = tid >> 3;

    auto load_tile_vec = [&](int t) {
        const long k = (long)t * BK + kq * 4;          // klen % 32 == 0: a tile never straddles a segment
        const bool k_ok = k < K;
        const int s = k_ok ? (int)(k / d.klen) : 0;
        const long kr = k_ok ? k - (long)s * d.klen : 0;
        const float* ap = a_base + (long)s * d.a_ss + kr;
        const float* bp = b_base + (long)s * d.b_ss + kr;
#pragma unroll
        for (int i = 0; i < RAV; ++i) {
            const int m = m0 + r0v + 32 * i;
            va[i] = (k_ok && m < d.M) ? *reinterpret_cast<const float4*>(ap + (long)m * d.lda) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RBV; ++i) {
            const int n = n0 + r0v + 32 * i;
            vb[i] = (k_ok && n < d.N) ? *reinterpret_cast<const float4*>(bp + (long)n * d.ldb) : make_float4(0, 0, 0, 0);
        }
    };
    auto store_tile_vec = [&]() {
#pragma unroll
        for (int i = 0; i < RAV; ++i) {
            if constexpr (X3) {
                uint2 whi, wlo;
                gd_split_bf2(va[i].x, va[i].y, whi.x, wlo.x);
                gd_split_bf2(va[i].z, va[i].w, whi.y, wlo.y);
                *reinterpret_cast<uint2*>(As + (r0v + 32 * i) * LD + kq * 4) = whi;
                *reinterpret_cast<uint2*>(As + A_PL + (r0v + 32 * i) * LD + kq * 4) = wlo;
            } else if constexpr (BF16) {
                uint2 w;
                w.x = gd_pack_bf2(va[i].x, va[i].y);
                w.y = gd_pack_bf2(va[i].z, va[i].w);
                *reinterpret_cast<uint2*>(As + (r0v + 32 * i) * LD + kq * 4) = w;
            } else {
                float* p = reinterpret_cast<float*>(As) + (r0v + 32 * i) * LD + kq * 4;
                p[0] = va[i].x; p[1] = va[i].y; p[2] = va[i].z; p[3] = va[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < RBV; ++i) {
            if (d.in_scale) {       // plain (non-im2col) operand: B row n is input channel n -> fused BN affine (+ ReLU)
                const int n = n0 + r0v + 32 * i;
                if (n < d.N) {
                    const float sc = d.in_scale[n], sh = d.in_shift[n];
                    vb[i].x = fmaf(vb[i].x, sc, sh); vb[i].y = fmaf(vb[i].y, sc, sh);
                    vb[i].z = fmaf(vb[i].z, sc, sh); vb[i].w = fmaf(vb[i].w, sc, sh);
                    if (d.in_relu) {
                        vb[i].x = fmaxf(vb[i].x, 0.f); vb[i].y = fmaxf(vb[i].y, 0.f);
                        vb[i].z = fmaxf(vb[i].z, 0.f); vb[i].w = fmaxf(vb[i].w, 0.f);
                    }
                }
            }
            if constexpr (X3) {
                uint2 whi, wlo;
                gd_split_bf2(vb[i].x, vb[i].y, whi.x, wlo.x);
                gd_split_bf2(vb[i].z, vb[i].w, whi.y, wlo.y);
                *reinterpret_cast<uint2*>(Bs + (r0v + 32 * i) * LD + kq * 4) = whi;
                *reinterpret_cast<uint2*>(Bs + B_PL + (r0v + 32 * i) * LD + kq * 4) = wlo;
            } else if constexpr (BF16) {
                uint2 w;
                w.x = gd_pack_bf2(vb[i].x, vb[i].y);
                w.y = gd_pack_bf2(vb[i].z, vb[i].w);
                *reinterpret_cast<uint2*>(Bs + (r0v + 32 * i) * LD + kq * 4) = w;
            } else {
                float* p = reinterpret_cast<float*>(Bs) + (r0v + 32 * i) * LD + kq * 4;
                p[0] = vb[i].x; p[1] = vb[i].y; p[2] = vb[i].z; p[3] = vb[i].w;
            }
        }
    };

    auto load_tile = [&](int t) {
        const long k = (long)t * BK + kk;
        const bool k_ok = k < K;
        const int s = k_ok ? (int)(k / d.klen) : 0;
        const long kr = k_ok ? k - (long)s * d.klen : 0;
        const float* ap = a_base + (long)s * d.a_ss + kr;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + r0 + 8 * i;
            ra[i] = (k_ok && m < d.M) ? ap[(long)m * d.lda] : 0.f;
        }
        if (!d.im2col) {
            const float* bp = b_base + (long)s * d.b_ss + kr;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int n = n0 + r0 + 8 * i;
                float v = (k_ok && n < d.N) ? bp[(long)n * d.ldb] : 0.f;
                if (d.in_scale && k_ok && n < d.N) {
                    v = fmaf(v, d.in_scale[n], d.in_shift[n]);
                    if (d.in_relu) v = fmaxf(v, 0.f);
                }
                rb[i] = v;
            }
        } else {
            const int oy = (int)(kr / d.Wo), ox = (int)(kr - (long)oy * d.Wo);
            const float* xs = b_base + (long)s * d.b_ss;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                float v = 0.f;
                if (k_ok && rc_dydx[i] >= 0) {
                    const int iy = oy * d.stride - d.pad + (rc_dydx[i] >> 16);
                    const int ix = ox * d.stride - d.pad + (rc_dydx[i] & 0xffff);
                    if (iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi) {
                        const int c = rc_off[i];
                        v = xs[(long)c * HiWi + (long)iy * d.Wi + ix];
                        if (d.in_scale) {
                            v = fmaf(v, d.in_scale[c], d.in_shift[c]);
                            if (d.in_relu) v = fmaxf(v, 0.f);
                        }
                    }
                }
                rb[i] = v;
            }
        }
    };

    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            if constexpr (X3) gd_split_bf(ra[i], As[(r0 + 8 * i) * LD + kk], As[A_PL + (r0 + 8 * i) * LD + kk]);
            else if constexpr (BF16) As[(r0 + 8 * i) * LD + kk] = gd_f2bf(ra[i]);
            else As[(r0 + 8 * i) * LD + kk] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if constexpr (X3) gd_split_bf(rb[i], Bs[(r0 + 8 * i) * LD + kk], Bs[B_PL + (r0 + 8 * i) * LD + kk]);
            else if constexpr (BF16) Bs[(r0 + 8 * i) * LD + kk] = gd_f2bf(rb[i]);
            else Bs[(r0 + 8 * i) * LD + kk] = rb[i];
        }
    };

    const int t_begin = split * tiles_per_split;
    const int t_end = min(ktiles, t_begin + tiles_per_split);
    if (t_begin >= t_end) return;  // whole block exits together (uniform)

    if constexpr (VEC) {
        load_tile_vec(t_begin);
        store_tile_vec();
    } else {
        load_tile(t_begin);
        store_tile();
    }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        if (t + 1 < t_end) {
            if constexpr (VEC) load_tile_vec(t + 1);
            else load_tile(t + 1);
        }
        gd::tile_mma<BM, MODE>(As, Bs, wm, wn, r, h, acc);
        __syncthreads();
        if (t + 1 < t_end) {
            if constexpr (VEC) store_tile_vec();
            else store_tile();
            __syncthreads();
        }
    }

    const float alpha = d.alpha ? *d.alpha : 1.f;
    float* cb = d.c + (long)b * d.c_bs;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + r;
            if (n >= d.N) continue;
            const float bias = (d.bias && split == 0) ? d.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                if (m >= d.M) continue;
                const float v = acc[i][j][e] * alpha + bias;
                float* cp = cb + (long)m * d.ldc + n;
                if (splits > 1) atomicAdd(cp, v);
                else if (d.accumulate) *cp += v;
                else *cp = v;
            }
        }
    }
}

template <int BM>
int launch(const gd_gemm_nt_desc& d, int ktiles, int splits, hipStream_t s) {
    const int tps = (ktiles + splits - 1) / splits;
    splits = (ktiles + tps - 1) / tps;  // no empty splits
    if (splits > 1 && !d.accumulate) {
        const long total = (long)d.B * d.M * d.N;
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(fill_zero_kernel, dim3(blocks), dim3(256), 0, s, d.c, d.c_bs, d.ldc, d.B, d.M, d.N);
    }
    dim3 grid(gd_cdiv(d.N, BN), gd_cdiv(d.M, BM), d.B * splits);
    // float4 staging needs: plain operands, every row start / segment / batch offset a multiple of 4 floats from a
    // 16-byte aligned base, and k tiles that never straddle a segment
    const bool vec = !d.im2col && d.klen % BK == 0 && d.lda % 4 == 0 && d.ldb % 4 == 0 &&
                     d.a_ss % 4 == 0 && d.b_ss % 4 == 0 && d.a_bs % 4 == 0 && d.b_bs % 4 == 0 &&
                     ((uintptr_t)d.a % 16) == 0 && ((uintptr_t)d.bm % 16) == 0;
    if (d.precision == GD_PREC_BF16) {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<BM, gd::MODE_BF16, true>), grid, dim3(256), 0, s, d, ktiles, tps, splits);
        else hipLaunchKernelGGL((gemm_nt_kernel<BM, gd::MODE_BF16, false>), grid, dim3(256), 0, s, d, ktiles, tps, splits);
    } else if (d.precision == GD_PREC_X3) {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<BM, gd::MODE_X3, true>), grid, dim3(256), 0, s, d, ktiles, tps, splits);
        else hipLaunchKernelGGL((gemm_nt_kernel<BM, gd::MODE_X3, false>), grid, dim3(256), 0, s, d, ktiles, tps, splits);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_nt_kernel<BM, gd::MODE_F32, true>), grid, dim3(256), 0, s, d, ktiles, tps, splits);
        else hipLaunchKernelGGL((gemm_nt_kernel<BM, gd::MODE_F32, false>), grid, dim3(256), 0, s, d, ktiles, tps, splits);
    }
    GD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int gd_gemm_nt(const gd_gemm_nt_desc* dp, void* stream) {
    GD_CHECK_ARG(dp != nullptr, "gd_gemm_nt: null descriptor");
    const gd_gemm_nt_desc& d = *dp;
    GD_CHECK_ARG(d.B > 0 && d.M > 0 && d.N > 0 && d.kseg > 0 && d.klen > 0, "gd_gemm_nt: bad sizes");
    GD_CHECK_ARG(d.a && d.bm && d.c, "gd_gemm_nt: null tensor");
    GD_CHECK_ARG((long)d.kseg * d.klen < (1L << 31), "gd_gemm_nt: reduction too long");
    GD_CHECK_ARG((d.in_scale == nullptr) == (d.in_shift == nullptr), "gd_gemm_nt: in_scale/in_shift must come together");
    GD_CHECK_ARG(d.precision == GD_PREC_FP32 || d.precision == GD_PREC_BF16 || d.precision == GD_PREC_X3, "gd_gemm_nt: bad precision");
    if (d.im2col) {
        GD_CHECK_ARG(d.ks > 0 && d.stride > 0 && d.pad >= 0 && d.Hi > 0 && d.Wi > 0 && d.Ho > 0 && d.Wo > 0,
                     "gd_gemm_nt: bad im2col geometry");
        GD_CHECK_ARG(d.klen == (long)d.Ho * d.Wo, "gd_gemm_nt: im2col needs klen == Ho*Wo");
        GD_CHECK_ARG(d.N % (d.ks * d.ks) == 0, "gd_gemm_nt: im2col needs N == C*ks*ks");
        GD_CHECK_ARG((d.Hi + 2 * d.pad - d.ks) / d.stride + 1 == d.Ho && (d.Wi + 2 * d.pad - d.ks) / d.stride + 1 == d.Wo,
                     "gd_gemm_nt: im2col Ho/Wo mismatch");
    }
    GD_CHECK_ARG((long)d.B * 4096 < 65535L * 64, "gd_gemm_nt: batch too large");
    const int ktiles = gd_cdiv((long)d.kseg * d.klen, BK);
    const int bm = d.M <= 32 ? 32 : (d.M <= 64 || (d.M % 128 != 0 && d.M % 128 <= 64)) ? 64 : 128;
    int splits = d.splits;
    if (splits <= 0) {
        const long tiles = (long)gd_cdiv(d.N, BN) * gd_cdiv(d.M, bm) * d.B;
        splits = (int)(2048 / tiles);
        if (splits < 1) splits = 1;
        // keep at least 8 k-tiles per split so the atomics stay a small fraction of the traffic
        if (splits > ktiles / 8) splits = ktiles / 8 > 0 ? ktiles / 8 : 1;
    }
    if (splits > ktiles) splits = ktiles;
    if (gd_get_deterministic()) splits = 1;                  // deterministic mode: no atomic combine of k-splits
    if ((long)d.B * splits > 65535) splits = (int)(65535 / d.B);
    hipStream_t s = (hipStream_t)stream;
    if (bm == 32) return launch<32>(d, ktiles, splits, s);
    if (bm == 64) return launch<64>(d, ktiles, splits, s);
    return launch<128>(d, ktiles, splits, s);
}
