// One block-tile worth of MFMA work, shared by the NN (implicit-GEMM conv) and NT (split-K) kernels.
// LDS images: As[BM][LD], Bs[128][LD], k contiguous in both.  4 waves arranged WAVES_M x WAVES_N,
// each wave owns TM x TN MFMA tiles of 32x32.
//
// Fragment maps (guide section 3):
//   v_mfma_f32_32x32x16_bf16 : lane l (r = l&31, h = l>>5) supplies A[row r][k = 8h..8h+7], B[k = 8h..8h+7][col r]
//   v_mfma_f32_32x32x2_f32   : lane l supplies A[row r][k = h], B[k = h][col r]
//   C/D (both)               : col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma once
#include "common.h"

namespace gd {

constexpr int TILE_BN = 128;
constexpr int TILE_BK = 32;

// operand modes of the tile kernels (= GD_PREC_*): 0 exact f32, 1 bf16, 2 split-bf16 ("x3": every operand staged as a hi
// and a lo bf16 image, three MFMAs per product: hi*hi + lo*hi + hi*lo, ~2^-16 relative, at a third of the bf16 rate
// instead of the eighth of the f32 MFMA)
constexpr int MODE_F32 = 0, MODE_BF16 = 1, MODE_X3 = 2;
template <int MODE> struct TilePol;
template <> struct TilePol<MODE_BF16> {
    using elem = unsigned short;
    static constexpr int LD = TILE_BK + 8;  // 40 bf16 = 80-byte rows: 16-B writes and reads conflict free
    static constexpr int PLANES = 1;
};
template <> struct TilePol<MODE_F32> {
    using elem = float;
    static constexpr int LD = TILE_BK + 1;  // 33 floats
    static constexpr int PLANES = 1;
};
template <> struct TilePol<MODE_X3> {
    using elem = unsigned short;
    static constexpr int LD = TILE_BK + 8;
    static constexpr int PLANES = 2;        // plane 0 hi, plane 1 lo (As + BM * LD, Bs + TILE_BN * LD)
};

template <int BM> struct TileGeom {
    static constexpr int WAVES_M = (BM >= 64) ? 2 : 1;
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int TM = BM / (32 * WAVES_M);
    static constexpr int TN = TILE_BN / (32 * WAVES_N);
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_native_t;

template <int BM, int MODE>
__device__ __forceinline__ void tile_mma(const typename TilePol<MODE>::elem* As, const typename TilePol<MODE>::elem* Bs,
                                         int wm, int wn, int r, int h,
                                         f32x16_t (&acc)[TileGeom<BM>::TM][TileGeom<BM>::TN]) {
    constexpr int LD = TilePol<MODE>::LD;
    constexpr int TM = TileGeom<BM>::TM, TN = TileGeom<BM>::TN;
    if constexpr (MODE == MODE_X3) {
#pragma unroll
        for (int ks = 0; ks < TILE_BK / 16; ++ks) {
            bf16x8_t fa[2][TM], fb[2][TN];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[pl][i] = *reinterpret_cast<const bf16x8_t*>(As + pl * BM * LD + (wm * TM * 32 + i * 32 + r) * LD + ks * 16 + 8 * h);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[pl][j] = *reinterpret_cast<const bf16x8_t*>(Bs + pl * TILE_BN * LD + (wn * TN * 32 + j * 32 + r) * LD + ks * 16 + 8 * h);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // the two small products first, the hi*hi product last
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_native_t, fa[1][i]),
                                                                       __builtin_bit_cast(bf16x8_native_t, fb[0][j]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_native_t, fa[0][i]),
                                                                       __builtin_bit_cast(bf16x8_native_t, fb[1][j]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_native_t, fa[0][i]),
                                                                       __builtin_bit_cast(bf16x8_native_t, fb[0][j]), acc[i][j], 0, 0, 0);
                }
        }
    } else if constexpr (MODE == MODE_BF16) {
#pragma unroll
        for (int ks = 0; ks < TILE_BK / 16; ++ks) {
            bf16x8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * TM * 32 + i * 32 + r) * LD + ks * 16 + 8 * h);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * TN * 32 + j * 32 + r) * LD + ks * 16 + 8 * h);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_native_t, fa[i]),
                                                                       __builtin_bit_cast(bf16x8_native_t, fb[j]),
                                                                       acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll 4
        for (int ks = 0; ks < TILE_BK / 2; ++ks) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = As[(wm * TM * 32 + i * 32 + r) * LD + ks * 2 + h];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Bs[(wn * TN * 32 + j * 32 + r) * LD + ks * 2 + h];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
}

// row index inside a 32x32 accumulator tile held by (lane half h, register e)
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

}  // namespace gd
