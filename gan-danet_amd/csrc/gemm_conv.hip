// Implicit-GEMM convolution for gfx950 ("NN" form): out[b][m][p] = sum_{tap,c} A[m][c][tap] * X~[c][p(+)tap]
//
// GEMM view: M = output channels, N = output pixels of ONE image (blockIdx.z = image), K = taps x channels.
// NCHW keeps pixels contiguous, so the im2col operand is read coalesced along N and the result is stored
// coalesced along N.  Block tile BM x 128 x 32, 256 threads = 4 waves, each wave a grid of 32x32 MFMA tiles.
// Operands are staged global -> registers -> LDS (the fp32 -> bf16 conversion and the fused BatchNorm+ReLU
// input transform happen in that register stage), LDS rows padded (80 B for bf16) so that both the
// 16-byte staging writes and the 16-byte fragment reads are bank-conflict free.
// Precision policies: BF16 (v_mfma_f32_32x32x16_bf16, f32 accumulate) and FP32 (v_mfma_f32_32x32x2_f32,
// bit-exact f32 fma chain) share every line except the fragment code.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "tile_mma.h"
#include "../../include/gandanet.h"

namespace {

using gd::TILE_BK;
using gd::TILE_BN;
constexpr int BN = TILE_BN;
constexpr int BK = TILE_BK;

template <int BM, int MODE>
__global__ __launch_bounds__(256) void conv_nn_kernel(const gd_conv_desc d) {
    constexpr bool BF16 = MODE != gd::MODE_F32;     // 16-bit LDS images (one, or hi + lo for the split mode)
    constexpr bool X3 = MODE == gd::MODE_X3;
    using P = gd::TilePol<MODE>;
    using elem = typename P::elem;
    constexpr int LD = P::LD;
    constexpr int WAVES_N = gd::TileGeom<BM>::WAVES_N;
    constexpr int TM = gd::TileGeom<BM>::TM;
    constexpr int TN = gd::TileGeom<BM>::TN;
    constexpr int KPT_A = BM / 8;   // A elements per thread per tile (BM*32/256)
    constexpr int KPT_B = 16;       // B elements per thread per tile (128*32/256)

    __shared__ __attribute__((aligned(16))) elem As[P::PLANES * BM * LD];
    __shared__ __attribute__((aligned(16))) elem Bs[P::PLANES * BN * LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const long HiWi = (long)d.Hi * d.Wi;
    const int HoWo = d.Ho * d.Wo;

    // ---- A loader coordinates: thread -> (row am, k group) ------------------------------------
    const int am = tid % BM;
    const int akg = tid / BM;  // 0 .. 256/BM-1, each KPT_A wide
    const bool a_row_ok = (m0 + am) < d.M;
    const float* a_row = d.a + (long)b * d.a_bs + (long)(m0 + am) * d.a_sm;

    // ---- B loader coordinates: thread -> (pixel bn, k half) -----------------------------------
    const int bn = tid & (BN - 1);
    const int bkh = tid >> 7;  // 0/1, each 16 wide
    // pixels enumerate the output sub-lattice (step 1 = the whole image)
    const int sstep = d.sub_step > 1 ? d.sub_step : 1;
    const int Ws = (d.Wo - d.sub_ox + sstep - 1) / sstep, Hs = (d.Ho - d.sub_oy + sstep - 1) / sstep;
    const int HsWs = Hs * Ws;
    const int p = n0 + bn;
    const bool p_ok = p < HsWs;
    const int pa = p_ok ? p / Ws : 0;
    const int oy = d.sub_oy + pa * sstep;
    const int ox = d.sub_ox + (p_ok ? p - pa * Ws : 0) * sstep;
    const float* x_img = d.x + (long)b * d.x_bs;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int ktiles_per_tap = (d.Ck + BK - 1) / BK;
    const int ntaps = d.ks * d.ks;
    const int T = ntaps * ktiles_per_tap;

    float ra[KPT_A];
    float rb[KPT_B];

    auto load_tile = [&](int t) {
        const int tap = t / ktiles_per_tap;
        const int c0 = (t - tap * ktiles_per_tap) * BK;
        const int kh = tap / d.ks, kw = tap - kh * d.ks;
        // A: weights gather (small, cache resident)
        {
            const float* ap = a_row + (long)tap * d.a_st;
#pragma unroll
            for (int i = 0; i < KPT_A; ++i) {
                const int c = c0 + akg * KPT_A + i;
                ra[i] = (a_row_ok && c < d.Ck) ? ap[(long)c * d.a_sc] : 0.f;
            }
        }
        // B: im2col gather, coalesced along pixels
        {
            int iy, ix;
            bool ok = p_ok;
            if (!d.transposed) {
                iy = oy * d.stride - d.pad + kh;
                ix = ox * d.stride - d.pad + kw;
            } else {
                const int ty = oy + d.pad - kh, tx = ox + d.pad - kw;
                ok = ok && ty >= 0 && tx >= 0 && (ty % d.stride == 0) && (tx % d.stride == 0);
                iy = ty / d.stride;
                ix = tx / d.stride;
            }
            ok = ok && iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi;
            const float* xp = x_img + (long)iy * d.Wi + ix;
#pragma unroll
            for (int i = 0; i < KPT_B; ++i) {
                const int c = c0 + bkh * KPT_B + i;
                float v = 0.f;
                if (ok && c < d.Ck) {
                    v = xp[(long)c * HiWi];
                    if (d.in_scale) {
                        v = fmaf(v, d.in_scale[c], d.in_shift[c]);
                        if (d.in_relu) v = fmaxf(v, 0.f);
                    }
                }
                rb[i] = v;
            }
        }
    };

    // `n` (a multiple of 4) consecutive k of one LDS row: 16-bit image (+ the lo image of the split mode) or plain f32
    auto put_row = [&](elem* row, int plane_stride, const float* v, auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (BF16 && n % 8 == 0) {                // 16-byte LDS writes
#pragma unroll
            for (int i = 0; i < n; i += 8) {
                uint4 whi, wlo;
                if constexpr (X3) {
                    gd_split_bf2(v[i + 0], v[i + 1], whi.x, wlo.x);
                    gd_split_bf2(v[i + 2], v[i + 3], whi.y, wlo.y);
                    gd_split_bf2(v[i + 4], v[i + 5], whi.z, wlo.z);
                    gd_split_bf2(v[i + 6], v[i + 7], whi.w, wlo.w);
                    *reinterpret_cast<uint4*>(row + plane_stride + i) = wlo;
                } else {
                    whi.x = gd_pack_bf2(v[i + 0], v[i + 1]);
                    whi.y = gd_pack_bf2(v[i + 2], v[i + 3]);
                    whi.z = gd_pack_bf2(v[i + 4], v[i + 5]);
                    whi.w = gd_pack_bf2(v[i + 6], v[i + 7]);
                }
                *reinterpret_cast<uint4*>(row + i) = whi;
            }
        } else if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < n; i += 4) {
                uint2 whi, wlo;
                if constexpr (X3) {
                    gd_split_bf2(v[i + 0], v[i + 1], whi.x, wlo.x);
                    gd_split_bf2(v[i + 2], v[i + 3], whi.y, wlo.y);
                    *reinterpret_cast<uint2*>(row + plane_stride + i) = wlo;
                } else {
                    whi.x = gd_pack_bf2(v[i + 0], v[i + 1]);
                    whi.y = gd_pack_bf2(v[i + 2], v[i + 3]);
                }
                *reinterpret_cast<uint2*>(row + i) = whi;
            }
        } else {
#pragma unroll
            for (int i = 0; i < n; ++i) row[i] = v[i];
        }
    };
    auto store_tile = [&]() {
        put_row(As + am * LD + akg * KPT_A, BM * LD, ra, std::integral_constant<int, KPT_A>{});      // A: KPT_A consecutive k at row am
        put_row(Bs + bn * LD + bkh * KPT_B, BN * LD, rb, std::integral_constant<int, KPT_B>{});
    };

    auto compute_tile = [&]() { gd::tile_mma<BM, MODE>(As, Bs, wm, wn, r, h, acc); };

    // tiles whose tap cannot reach this output parity class are skipped (workgroup-uniform; strided data gradient)
    auto tile_live = [&](int t) -> bool {
        if (!d.transposed || sstep == 1) return true;
        const int tap = t / ktiles_per_tap;
        const int kh = tap / d.ks, kw = tap - kh * d.ks;
        return ((d.sub_oy + d.pad - kh) % d.stride == 0) && ((d.sub_ox + d.pad - kw) % d.stride == 0);
    };
    auto next_live = [&](int t) -> int {
        while (t < T && !tile_live(t)) ++t;
        return t;
    };
    int t = next_live(0);
    if (t < T) {
        load_tile(t);
        store_tile();
    }
    __syncthreads();
    while (t < T) {
        const int tn = next_live(t + 1);
        if (tn < T) load_tile(tn);  // global loads in flight under the MFMAs
        compute_tile();
        __syncthreads();
        if (tn < T) {
            store_tile();
            __syncthreads();
        }
        t = tn;
    }

    // ---- epilogue ----------------------------------------------------------------------------
    const float alpha = d.alpha ? *d.alpha : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ps = n0 + wn * TN * 32 + j * 32 + r;
            if (ps >= HsWs) continue;
            const int qa = ps / Ws;
            const int pn = (d.sub_oy + qa * sstep) * d.Wo + d.sub_ox + (ps - qa * Ws) * sstep;
            // residual / accumulate operands: all loads before the first store (see conv3x3_halo_kernel's epilogue)
            float rsv[16], oldv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                rsv[e] = (d.res && m < d.M) ? d.res[(long)b * d.res_bs + (long)m * HoWo + pn] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                const long idx = d.out_layout == 0 ? (long)b * d.y_bs + (long)m * HoWo + pn
                                                   : (long)b * d.y_bs + (long)pn * d.ldo + m;
                oldv[e] = (d.accumulate && !d.out_bf16 && m < d.Mstore) ? reinterpret_cast<const float*>(d.y)[idx] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                if (m >= d.Mstore) continue;
                float v = acc[i][j][e] * alpha;
                if (m < d.M) {
                    if (d.bias) v += d.bias[m];
                    v += rsv[e];
                }
                if (d.act == GD_ACT_RELU) v = fmaxf(v, 0.f);
                else if (d.act == GD_ACT_LEAKY02) v = v >= 0.f ? v : 0.2f * v;
                const long idx = d.out_layout == 0 ? (long)b * d.y_bs + (long)m * HoWo + pn
                                                   : (long)b * d.y_bs + (long)pn * d.ldo + m;
                if (d.out_bf16) reinterpret_cast<unsigned short*>(d.y)[idx] = gd_f2bf(v);
                else reinterpret_cast<float*>(d.y)[idx] = v + oldv[e];
            }
        }
    }
}

// =====================================================================================================
// 1x1 / stride 1 convolution (bf16): out[b][m][p] = sum_c A[b][m][c] * X~[b][c][p]
// The activation operand is staged the way it lies in memory -- [channel][pixel], pixel-contiguous, float4
// loads, 8-byte LDS writes -- and the MFMA B fragments (k = channel, n = pixel) are taken with
// ds_read_b64_tr_b16, the LDS transpose read.  Rows are 128 pixels + 32 pad (320 B): the four channel rows of a
// transpose-read block then start 16 banks apart, conflict free.
// =====================================================================================================
typedef short s16x4_t __attribute__((ext_vector_type(4)));
constexpr int XLD = BN + 32;   // elements per staged channel row

// NS: 128-pixel sub-tiles per workgroup, the k loop OUTSIDE them: a workgroup then reads 512 NS contiguous bytes of every
// channel row per k-tile instead of 512 (round 3 experiment on round 2's address-translation suspicion: the input is
// fp32 NCHW, a k-tile touches 32 channel rows 4 HW bytes apart).  MEASURED at the bench shapes: NS = 2 / 4 are 20 - 50 %
// SLOWER than NS = 1 (184 -> 230 projection: 3.1 / 3.6 / 3.6 ms; 184 -> 256 data gradient 1.7 / 2.2 / 3.0 ms): page
// locality is not what holds this kernel back, occupancy (312 registers at NS = 4) costs more.  NS = 1 is the default;
// the other instantiations stay behind GD_CONV1X1_NS for A/B runs.
// X3: split-bf16 operands (hi and lo LDS images of both tiles, three MFMAs per product; gd::MODE_X3)
template <int BM, int NS, bool X3 = false>
__global__ __launch_bounds__(256) void conv1x1_tr_kernel(const gd_conv_desc d) {
    constexpr int LDA = BK + 8;
    constexpr int PL = X3 ? 2 : 1, A_PL = BM * LDA, X_PL = NS * BK * XLD;
    constexpr int WAVES_N = gd::TileGeom<BM>::WAVES_N;
    constexpr int TM = gd::TileGeom<BM>::TM;
    constexpr int TN = gd::TileGeom<BM>::TN;
    constexpr int KPT_A = BM / 8;
    __shared__ __attribute__((aligned(16))) unsigned short As[PL * A_PL];
    __shared__ __attribute__((aligned(16))) unsigned short Xs[PL * X_PL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int b = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * (BN * NS);
    const int HW = d.Ho * d.Wo;

    const int am = tid % BM, akg = tid / BM;
    const bool a_row_ok = (m0 + am) < d.M;
    const float* a_row = d.a + (long)b * d.a_bs + (long)(m0 + am) * d.a_sm;

    // X loader: thread -> (channel rows xr0 + 8 i, pixel quad xq of every sub-tile): 32 rows x 32 quads x NS
    const int xq = tid & 31, xr0 = tid >> 5;
    const float* x_img = d.x + (long)b * d.x_bs;

    f32x16_t acc[NS][TM][TN];
#pragma unroll
    for (int sb = 0; sb < NS; ++sb)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[sb][i][j][e] = 0.f;

    float ra[KPT_A];
    float4 rx[NS][4];
    const int T = (d.Ck + BK - 1) / BK;
    auto load_tile = [&](int t) {
        const int c0 = t * BK;
#pragma unroll
        for (int i = 0; i < KPT_A; ++i) {
            const int c = c0 + akg * KPT_A + i;
            ra[i] = (a_row_ok && c < d.Ck) ? a_row[(long)c * d.a_sc] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + xr0 + 8 * i;
            float sc = 1.f, sh = 0.f;
            if (d.in_scale && c < d.Ck) { sc = d.in_scale[c]; sh = d.in_shift[c]; }
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) {
                const int pq = n0 + sb * BN + xq * 4;              // first pixel of this thread's quad
                const bool q_full = pq + 3 < HW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < d.Ck && pq < HW) {
                    const float* p = x_img + (long)c * HW + pq;
                    if (q_full) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (pq + 0 < HW) v.x = p[0];
                        if (pq + 1 < HW) v.y = p[1];
                        if (pq + 2 < HW) v.z = p[2];
                    }
                    if (d.in_scale) {
                        v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
                        if (d.in_relu) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        if (!q_full) {   // padding pixels stay zero
                            if (pq + 0 >= HW) v.x = 0.f;
                            if (pq + 1 >= HW) v.y = 0.f;
                            if (pq + 2 >= HW) v.z = 0.f;
                            v.w = 0.f;
                        }
                    }
                }
                rx[sb][i] = v;
            }
        }
    };
    auto store_tile = [&]() {
        if constexpr (X3) {
        unsigned short* ap = As + am * LDA + akg * KPT_A;
        if constexpr (KPT_A >= 8) {
#pragma unroll
            for (int i = 0; i < KPT_A; i += 8) {
                uint4 whi, wlo;
                if constexpr (X3) {
                    gd_split_bf2(ra[i + 0], ra[i + 1], whi.x, wlo.x);
                    gd_split_bf2(ra[i + 2], ra[i + 3], whi.y, wlo.y);
                    gd_split_bf2(ra[i + 4], ra[i + 5], whi.z, wlo.z);
                    gd_split_bf2(ra[i + 6], ra[i + 7], whi.w, wlo.w);
                    *reinterpret_cast<uint4*>(ap + A_PL + i) = wlo;
                } else {
                    whi.x = gd_pack_bf2(ra[i + 0], ra[i + 1]);
                    whi.y = gd_pack_bf2(ra[i + 2], ra[i + 3]);
                    whi.z = gd_pack_bf2(ra[i + 4], ra[i + 5]);
                    whi.w = gd_pack_bf2(ra[i + 6], ra[i + 7]);
                }
                *reinterpret_cast<uint4*>(ap + i) = whi;
            }
        } else {
            uint2 whi, wlo;
            if constexpr (X3) {
                gd_split_bf2(ra[0], ra[1], whi.x, wlo.x);
                gd_split_bf2(ra[2], ra[3], whi.y, wlo.y);
                *reinterpret_cast<uint2*>(ap + A_PL) = wlo;
            } else {
                whi.x = gd_pack_bf2(ra[0], ra[1]);
                whi.y = gd_pack_bf2(ra[2], ra[3]);
            }
            *reinterpret_cast<uint2*>(ap) = whi;
        }
#pragma unroll
        for (int sb = 0; sb < NS; ++sb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 whi, wlo;
                unsigned short* xp = Xs + sb * (BK * XLD) + (xr0 + 8 * i) * XLD + xq * 4;
                if constexpr (X3) {
                    gd_split_bf2(rx[sb][i].x, rx[sb][i].y, whi.x, wlo.x);
                    gd_split_bf2(rx[sb][i].z, rx[sb][i].w, whi.y, wlo.y);
                    *reinterpret_cast<uint2*>(xp + X_PL) = wlo;
                } else {
                    whi.x = gd_pack_bf2(rx[sb][i].x, rx[sb][i].y);
                    whi.y = gd_pack_bf2(rx[sb][i].z, rx[sb][i].w);
                }
                *reinterpret_cast<uint2*>(xp) = whi;
            }
        } else {     // the plain instantiation keeps round 2's statement order (its unrolled codegen is 18 % faster)
        unsigned short* ap = As + am * LDA + akg * KPT_A;
        if constexpr (KPT_A >= 8) {
#pragma unroll
            for (int i = 0; i < KPT_A; i += 8) {
                uint4 w;
                w.x = gd_pack_bf2(ra[i + 0], ra[i + 1]);
                w.y = gd_pack_bf2(ra[i + 2], ra[i + 3]);
                w.z = gd_pack_bf2(ra[i + 4], ra[i + 5]);
                w.w = gd_pack_bf2(ra[i + 6], ra[i + 7]);
                *reinterpret_cast<uint4*>(ap + i) = w;
            }
        } else {
            uint2 w;
            w.x = gd_pack_bf2(ra[0], ra[1]);
            w.y = gd_pack_bf2(ra[2], ra[3]);
            *reinterpret_cast<uint2*>(ap) = w;
        }
#pragma unroll
        for (int sb = 0; sb < NS; ++sb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 w;
                w.x = gd_pack_bf2(rx[sb][i].x, rx[sb][i].y);
                w.y = gd_pack_bf2(rx[sb][i].z, rx[sb][i].w);
                *reinterpret_cast<uint2*>(Xs + sb * (BK * XLD) + (xr0 + 8 * i) * XLD + xq * 4) = w;
            }
        }
    };
    // transpose-read roles: 16-lane group = 4 (k) x 16 (n) block; lane 4q+p supplies row q, columns 4p..4p+3
    const int li = lane & 15, tq = li >> 2, tp = li & 3, tg = (lane >> 4) & 1;
    auto mma = [&](f32x16_t& c, const bf16x8_t& a, const bf16x8_t& bb) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gd::bf16x8_native_t, a),
                                                    __builtin_bit_cast(gd::bf16x8_native_t, bb), c, 0, 0, 0);
    };
    auto compute_tile = [&]() {
        if constexpr (X3) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8_t fa[PL][TM];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[pl][i] = *reinterpret_cast<const bf16x8_t*>(As + pl * A_PL + (wm * TM * 32 + i * 32 + r) * LDA + ks * 16 + 8 * h);
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) {
                bf16x8_t fb[PL][TN];
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned short* p = Xs + pl * X_PL + sb * (BK * XLD) + (ks * 16 + 8 * h + tq) * XLD + (wn * TN + j) * 32 + 16 * tg + 4 * tp;
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4_t*)(p + 4 * XLD));
                        fb[pl][j] = bf16x8_t{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (X3) {
                            mma(acc[sb][i][j], fa[1][i], fb[0][j]);
                            mma(acc[sb][i][j], fa[0][i], fb[1][j]);
                        }
                        mma(acc[sb][i][j], fa[0][i], fb[0][j]);
                    }
            }
        }
        } else {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8_t fa[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * TM * 32 + i * 32 + r) * LDA + ks * 16 + 8 * h);
#pragma unroll
            for (int sb = 0; sb < NS; ++sb) {
                bf16x8_t fb[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned short* p = Xs + sb * (BK * XLD) + (ks * 16 + 8 * h + tq) * XLD + (wn * TN + j) * 32 + 16 * tg + 4 * tp;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(p + 4 * XLD));
                    fb[j] = bf16x8_t{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[sb][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(gd::bf16x8_native_t, fa[i]), __builtin_bit_cast(gd::bf16x8_native_t, fb[j]),
                            acc[sb][i][j], 0, 0, 0);
            }
        }
        }
    };

    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) load_tile(t + 1);
        compute_tile();
        __syncthreads();
        if (t + 1 < T) {
            store_tile();
            __syncthreads();
        }
    }

    const float alpha = d.alpha ? *d.alpha : 1.f;
#pragma unroll
    for (int sb = 0; sb < NS; ++sb)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pn = n0 + sb * BN + wn * TN * 32 + j * 32 + r;
            if (pn >= HW) continue;
            // residual / accumulate operands: all loads before the first store (see conv3x3_halo_kernel's epilogue)
            float rsv[16], oldv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                rsv[e] = (d.res && m < d.M) ? d.res[(long)b * d.res_bs + (long)m * HW + pn] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                oldv[e] = (d.accumulate && m < d.M) ? reinterpret_cast<const float*>(d.y)[(long)b * d.y_bs + (long)m * HW + pn] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                if (m >= d.M) continue;
                float v = acc[sb][i][j][e] * alpha + rsv[e];
                if (d.bias) v += d.bias[m];
                if (d.act == GD_ACT_RELU) v = fmaxf(v, 0.f);
                else if (d.act == GD_ACT_LEAKY02) v = v >= 0.f ? v : 0.2f * v;
                reinterpret_cast<float*>(d.y)[(long)b * d.y_bs + (long)m * HW + pn] = v + oldv[e];
            }
        }
}

// The plain (bf16, one 128-pixel sub-tile) form as round 2 wrote it: the generalised kernel above (NS sub-tiles, split-bf16
// images) compiles its NS = 1 / X3 = false instantiation to a loop that measures 18 % slower on every 1x1 shape of the bench
// (1.37 against 1.15 ms at 184 -> 184, 256 x 256, B = 32: profiles/r03_bench_kernel_stats.txt), so the default path keeps
// this one and the template serves GD_CONV1X1_NS > 1 and GD_PREC_X3.
template <int BM, bool X3 = false>
__global__ __launch_bounds__(256) void conv1x1_tr_plain_kernel(const gd_conv_desc d) {
    constexpr int LDA = BK + 8;
    constexpr int A_PL = BM * LDA, X_PL = BK * XLD;      // X3: a second (lo) LDS image behind each tile
    constexpr int WAVES_N = gd::TileGeom<BM>::WAVES_N;
    constexpr int TM = gd::TileGeom<BM>::TM;
    constexpr int TN = gd::TileGeom<BM>::TN;
    constexpr int KPT_A = BM / 8;
    __shared__ __attribute__((aligned(16))) unsigned short As[(X3 ? 2 : 1) * BM * LDA];
    __shared__ __attribute__((aligned(16))) unsigned short Xs[(X3 ? 2 : 1) * BK * XLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int b = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int HW = d.Ho * d.Wo;

    const int am = tid % BM, akg = tid / BM;
    const bool a_row_ok = (m0 + am) < d.M;
    const float* a_row = d.a + (long)b * d.a_bs + (long)(m0 + am) * d.a_sm;

    // X loader: thread -> (channel row xr = tid>>3 .. +0, pixel quad xq = tid&7 .. +8*i): 32 rows x 32 quads
    const int xq = tid & 31, xr0 = tid >> 5;          // 8 threads rows apart: rows xr0 + 8*i, quad xq
    const float* x_img = d.x + (long)b * d.x_bs;
    const int pq = n0 + xq * 4;                        // first pixel of this thread's quad
    const bool q_full = pq + 3 < HW;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[KPT_A];
    float4 rx[4];
    const int T = (d.Ck + BK - 1) / BK;
    auto load_tile = [&](int t) {
        const int c0 = t * BK;
#pragma unroll
        for (int i = 0; i < KPT_A; ++i) {
            const int c = c0 + akg * KPT_A + i;
            ra[i] = (a_row_ok && c < d.Ck) ? a_row[(long)c * d.a_sc] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + xr0 + 8 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < d.Ck) {
                const float* p = x_img + (long)c * HW + pq;
                if (q_full) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (pq + 0 < HW) v.x = p[0];
                    if (pq + 1 < HW) v.y = p[1];
                    if (pq + 2 < HW) v.z = p[2];
                }
                if (d.in_scale) {
                    const float sc = d.in_scale[c], sh = d.in_shift[c];
                    v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
                    if (d.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if (!q_full) {   // padding pixels stay zero
                        if (pq + 0 >= HW) v.x = 0.f;
                        if (pq + 1 >= HW) v.y = 0.f;
                        if (pq + 2 >= HW) v.z = 0.f;
                        v.w = 0.f;
                    }
                }
            }
            rx[i] = v;
        }
    };
    auto store_tile = [&]() {
        unsigned short* ap = As + am * LDA + akg * KPT_A;
        if constexpr (X3) {
#pragma unroll
            for (int i = 0; i < KPT_A; i += 4) {
                uint2 whi, wlo;
                gd_split_bf2(ra[i + 0], ra[i + 1], whi.x, wlo.x);
                gd_split_bf2(ra[i + 2], ra[i + 3], whi.y, wlo.y);
                *reinterpret_cast<uint2*>(ap + i) = whi;
                *reinterpret_cast<uint2*>(ap + A_PL + i) = wlo;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 whi, wlo;
                gd_split_bf2(rx[i].x, rx[i].y, whi.x, wlo.x);
                gd_split_bf2(rx[i].z, rx[i].w, whi.y, wlo.y);
                *reinterpret_cast<uint2*>(Xs + (xr0 + 8 * i) * XLD + xq * 4) = whi;
                *reinterpret_cast<uint2*>(Xs + X_PL + (xr0 + 8 * i) * XLD + xq * 4) = wlo;
            }
            return;
        }
        if constexpr (KPT_A >= 8) {
#pragma unroll
            for (int i = 0; i < KPT_A; i += 8) {
                uint4 w;
                w.x = gd_pack_bf2(ra[i + 0], ra[i + 1]);
                w.y = gd_pack_bf2(ra[i + 2], ra[i + 3]);
                w.z = gd_pack_bf2(ra[i + 4], ra[i + 5]);
                w.w = gd_pack_bf2(ra[i + 6], ra[i + 7]);
                *reinterpret_cast<uint4*>(ap + i) = w;
            }
        } else {
            uint2 w;
            w.x = gd_pack_bf2(ra[0], ra[1]);
            w.y = gd_pack_bf2(ra[2], ra[3]);
            *reinterpret_cast<uint2*>(ap) = w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 w;
            w.x = gd_pack_bf2(rx[i].x, rx[i].y);
            w.y = gd_pack_bf2(rx[i].z, rx[i].w);
            *reinterpret_cast<uint2*>(Xs + (xr0 + 8 * i) * XLD + xq * 4) = w;
        }
    };
    // transpose-read roles: 16-lane group = 4 (k) x 16 (n) block; lane 4q+p supplies row q, columns 4p..4p+3
    const int li = lane & 15, tq = li >> 2, tp = li & 3, tg = (lane >> 4) & 1;
    auto compute_tile = [&]() {
        if constexpr (X3) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8_t fa[2][TM], fb[2][TN];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[pl][i] = *reinterpret_cast<const bf16x8_t*>(As + pl * A_PL + (wm * TM * 32 + i * 32 + r) * LDA + ks * 16 + 8 * h);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned short* p = Xs + pl * X_PL + (ks * 16 + 8 * h + tq) * XLD + (wn * TN + j) * 32 + 16 * tg + 4 * tp;
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4_t*)(p + 4 * XLD));
                        fb[pl][j] = bf16x8_t{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gd::bf16x8_native_t, fa[1][i]),
                                                                           __builtin_bit_cast(gd::bf16x8_native_t, fb[0][j]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gd::bf16x8_native_t, fa[0][i]),
                                                                           __builtin_bit_cast(gd::bf16x8_native_t, fb[1][j]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gd::bf16x8_native_t, fa[0][i]),
                                                                           __builtin_bit_cast(gd::bf16x8_native_t, fb[0][j]), acc[i][j], 0, 0, 0);
                    }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * TM * 32 + i * 32 + r) * LDA + ks * 16 + 8 * h);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned short* p = Xs + (ks * 16 + 8 * h + tq) * XLD + (wn * TN + j) * 32 + 16 * tg + 4 * tp;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(p + 4 * XLD));
                fb[j] = bf16x8_t{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(gd::bf16x8_native_t, fa[i]), __builtin_bit_cast(gd::bf16x8_native_t, fb[j]),
                        acc[i][j], 0, 0, 0);
        }
    };

    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) load_tile(t + 1);
        compute_tile();
        __syncthreads();
        if (t + 1 < T) {
            store_tile();
            __syncthreads();
        }
    }

    const float alpha = d.alpha ? *d.alpha : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pn = n0 + wn * TN * 32 + j * 32 + r;
            if (pn >= HW) continue;
            // residual / accumulate operands: all loads before the first store (see conv3x3_halo_kernel's epilogue)
            float rsv[16], oldv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                rsv[e] = (d.res && m < d.M) ? d.res[(long)b * d.res_bs + (long)m * HW + pn] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                oldv[e] = (d.accumulate && m < d.M) ? reinterpret_cast<const float*>(d.y)[(long)b * d.y_bs + (long)m * HW + pn] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM * 32 + i * 32 + gd::acc_row(e, h);
                if (m >= d.M) continue;
                float v = acc[i][j][e] * alpha + rsv[e];
                if (d.bias) v += d.bias[m];
                if (d.act == GD_ACT_RELU) v = fmaxf(v, 0.f);
                else if (d.act == GD_ACT_LEAKY02) v = v >= 0.f ? v : 0.2f * v;
                reinterpret_cast<float*>(d.y)[(long)b * d.y_bs + (long)m * HW + pn] = v + oldv[e];
            }
        }
}

static bool conv1x1_tr_eligible(const gd_conv_desc& d) {
    const long HW = (long)d.Ho * d.Wo;
    return d.ks == 1 && d.stride == 1 && d.pad == 0 && (d.precision == GD_PREC_BF16 || d.precision == GD_PREC_X3) && d.out_layout == 0 && !d.out_bf16 &&
           (d.Mstore == 0 || d.Mstore == d.M) && d.Hi == d.Ho && d.Wi == d.Wo && HW % 4 == 0 && d.x_bs % 4 == 0 &&
           ((uintptr_t)d.x % 16) == 0;
}

template <int BM>
int launch(const gd_conv_desc& d, hipStream_t s) {
    const int sstep = d.sub_step > 1 ? d.sub_step : 1;
    const long hs = (d.Ho - d.sub_oy + sstep - 1) / sstep, ws = (d.Wo - d.sub_ox + sstep - 1) / sstep;
    if (hs <= 0 || ws <= 0) return 0;
    dim3 grid(gd_cdiv(hs * ws, BN), gd_cdiv(d.Mstore, BM), d.B);
    if (d.precision == GD_PREC_BF16)
        hipLaunchKernelGGL((conv_nn_kernel<BM, gd::MODE_BF16>), grid, dim3(256), 0, s, d);
    else if (d.precision == GD_PREC_X3)
        hipLaunchKernelGGL((conv_nn_kernel<BM, gd::MODE_X3>), grid, dim3(256), 0, s, d);
    else
        hipLaunchKernelGGL((conv_nn_kernel<BM, gd::MODE_F32>), grid, dim3(256), 0, s, d);
    GD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int gd_conv2d(const gd_conv_desc* dp, void* stream) {
    GD_CHECK_ARG(dp != nullptr, "gd_conv2d: null descriptor");
    gd_conv_desc d = *dp;
    if (d.Mstore < d.M) d.Mstore = d.M;
    GD_CHECK_ARG(d.B > 0 && d.M > 0 && d.Ck > 0 && d.ks > 0 && d.stride > 0 && d.pad >= 0, "gd_conv2d: bad sizes");
    GD_CHECK_ARG(d.Hi > 0 && d.Wi > 0 && d.Ho > 0 && d.Wo > 0, "gd_conv2d: bad spatial sizes");
    GD_CHECK_ARG(d.a && d.x && d.y, "gd_conv2d: null tensor");
    GD_CHECK_ARG((d.in_scale == nullptr) == (d.in_shift == nullptr), "gd_conv2d: in_scale/in_shift must come together");
    GD_CHECK_ARG(d.precision == GD_PREC_FP32 || d.precision == GD_PREC_BF16 || d.precision == GD_PREC_X3, "gd_conv2d: bad precision");
    GD_CHECK_ARG(d.out_layout == 0 || (d.out_layout == 1 && d.ldo >= d.Mstore), "gd_conv2d: bad output layout");
    GD_CHECK_ARG(!(d.out_bf16 && d.accumulate), "gd_conv2d: accumulate needs an fp32 output");
    GD_CHECK_ARG((long)d.Ho * d.Wo < (1L << 31) && d.B <= 65535, "gd_conv2d: image too large for one launch");
    GD_CHECK_ARG(d.sub_step >= 0 && d.sub_oy >= 0 && d.sub_ox >= 0 && (d.sub_step > 1 || (d.sub_oy == 0 && d.sub_ox == 0)),
                 "gd_conv2d: bad output sub-lattice");
    if (!d.transposed) {
        // forward gather: the output size must be what this geometry produces
        GD_CHECK_ARG((d.Hi + 2 * d.pad - d.ks) / d.stride + 1 == d.Ho && (d.Wi + 2 * d.pad - d.ks) / d.stride + 1 == d.Wo,
                     "gd_conv2d: Ho/Wo do not match Hi/Wi, ks, stride, pad");
    } else {
        // transposed gather: X is the forward OUTPUT (Hi x Wi), y the forward input (Ho x Wo)
        GD_CHECK_ARG((d.Ho + 2 * d.pad - d.ks) / d.stride + 1 == d.Hi && (d.Wo + 2 * d.pad - d.ks) / d.stride + 1 == d.Wi,
                     "gd_conv2d: transposed geometry mismatch");
    }
    hipStream_t s = (hipStream_t)stream;
    static const int tr_env = getenv("GD_CONV1X1_TR") ? atoi(getenv("GD_CONV1X1_TR")) : 1;
    if (tr_env && conv1x1_tr_eligible(d)) {
        const int bm = d.M <= 32 ? 32 : (d.M <= 64 || (d.M % 128 != 0 && d.M % 128 <= 64)) ? 64 : 128;
        // long pixel runs per channel row (NS sub-tiles per workgroup) once the image is large enough to keep the chip full
        static const int ns_env = getenv("GD_CONV1X1_NS") ? atoi(getenv("GD_CONV1X1_NS")) : 1;   // measured: NS 2 / 4 are 20-50 % SLOWER (round 3)
        const long HWl = (long)d.Ho * d.Wo;
        int ns = (HWl * d.B >= (1L << 18) && d.precision == GD_PREC_BF16) ? ns_env : 1;
        if (bm == 128 && ns > 2) ns = 2;                  // 64 accumulator registers per sub-tile
        if (ns != 1 && ns != 2 && ns != 4) ns = 1;
        dim3 grid(gd_cdiv(HWl, BN * ns), gd_cdiv(d.M, bm), d.B);
#define GD_C1X1(BM_, NS_) hipLaunchKernelGGL((conv1x1_tr_kernel<BM_, NS_>), grid, dim3(256), 0, s, d)
        if (d.precision == GD_PREC_X3) {
            if (bm == 32) hipLaunchKernelGGL((conv1x1_tr_plain_kernel<32, true>), grid, dim3(256), 0, s, d);
            else if (bm == 64) hipLaunchKernelGGL((conv1x1_tr_plain_kernel<64, true>), grid, dim3(256), 0, s, d);
            else hipLaunchKernelGGL((conv1x1_tr_plain_kernel<128, true>), grid, dim3(256), 0, s, d);
        } else if (ns == 1) {
            if (bm == 32) hipLaunchKernelGGL((conv1x1_tr_plain_kernel<32>), grid, dim3(256), 0, s, d);
            else if (bm == 64) hipLaunchKernelGGL((conv1x1_tr_plain_kernel<64>), grid, dim3(256), 0, s, d);
            else hipLaunchKernelGGL((conv1x1_tr_plain_kernel<128>), grid, dim3(256), 0, s, d);
        } else
        if (bm == 32) { if (ns == 4) GD_C1X1(32, 4); else if (ns == 2) GD_C1X1(32, 2); else GD_C1X1(32, 1); }
        else if (bm == 64) { if (ns == 4) GD_C1X1(64, 4); else if (ns == 2) GD_C1X1(64, 2); else GD_C1X1(64, 1); }
        else { if (ns == 2) GD_C1X1(128, 2); else GD_C1X1(128, 1); }
#undef GD_C1X1
        GD_LAUNCH_CHECK();
        return 0;
    }
    auto run = [&](const gd_conv_desc& dd) -> int {
        if (dd.Mstore <= 32) return launch<32>(dd, s);
        if (dd.Mstore <= 64 || (dd.Mstore % 128 != 0 && dd.Mstore % 128 <= 64)) return launch<64>(dd, s);
        return launch<128>(dd, s);
    };
    if (d.transposed && d.stride > 1 && d.sub_step <= 1) {
        // data gradient of a strided conv: one launch per output parity class, each with its reachable taps only
        for (int py = 0; py < d.stride; ++py)
            for (int px = 0; px < d.stride; ++px) {
                gd_conv_desc dd = d;
                dd.sub_oy = py; dd.sub_ox = px; dd.sub_step = d.stride;
                const int rc = run(dd);
                if (rc) return rc;
            }
        return 0;
    }
    return run(d);
}
