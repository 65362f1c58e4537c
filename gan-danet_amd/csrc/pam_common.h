// Helpers shared by the PAM kernels (pam.hip, pam_bwd64.hip): 16-bit MFMA wrappers for both operand types
// (bf16 = the training default, f16 = BASELINE config 5), accumulator-as-operand packing, LDS transpose reads.
#pragma once
#include "common.h"
#include "tile_mma.h"

namespace pam {

using gd::acc_row;
using gd::bf16x8_native_t;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // staging registers: first-class vectors, never
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));   // demoted to scratch like arrays of uint4 structs
typedef _Float16 f16x8_native_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_native_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// D = A B + C on 32x32x16 tiles; operands are 8 x 16-bit per lane (bf16 or f16 bit patterns in a short8)
template <bool F16>
__device__ __forceinline__ f32x16_t mfma16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_native_t, a),
                                                      __builtin_bit_cast(f16x8_native_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_native_t, a),
                                                       __builtin_bit_cast(bf16x8_native_t, b), c, 0, 0, 0);
}

// D = A B + C on 16x16x32 tiles (A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], D[row 4(l>>4)+e][col l&15]): the
// shape that draws less power per FLOP under load (tools/micro/mfma_shape.hip: +8..13 % FLOP/s in bare loops)
typedef float f32x4_acc_t __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ f32x4_acc_t mfma16x16(bf16x8_t a, bf16x8_t b, f32x4_acc_t c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_native_t, a),
                                                      __builtin_bit_cast(f16x8_native_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_native_t, a),
                                                       __builtin_bit_cast(bf16x8_native_t, b), c, 0, 0, 0);
}
// v_permlane16_swap on every dword of two fragments: x <- [x.row0, y.row0, x.row2, y.row2], y <- [x.row1, y.row1, x.row3,
// y.row3] (rows = 16-lane groups).  For the two k-step fragments of a 32x32 accumulator tile (lane = column r, half h)
// this gathers the columns 0..15 into x and 16..31 into y, each with its 32 k values spread over the four lane groups:
// the B operand of a 16x16x32 MFMA over those 16 columns (slot t of group g <-> row 16 (g&1) + 8 (t>>2) + 4 (g>>1) + (t&3)).
__device__ __forceinline__ void swap16_frags(bf16x8_t& x, bf16x8_t& y) {
    // one volatile asm statement: the scheduler would otherwise move the swaps up into hand-placed MFMA phases (whose asm
    // MFMAs it cannot see the hazards of); s_nop 1 = the two wait states between a VALU write and a permlane swap of it
    u32x4_t a = __builtin_bit_cast(u32x4_t, x), b = __builtin_bit_cast(u32x4_t, y);
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                 "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    x = __builtin_bit_cast(bf16x8_t, a);
    y = __builtin_bit_cast(bf16x8_t, b);
}

// two floats -> one dword of two 16-bit values (lo in bits 0..15), round to nearest even
template <bool F16>
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
    if constexpr (F16) {
        const gd_f32x2_t v = {lo, hi};
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_native_t));
    } else {
        return gd_pack_bf2(lo, hi);
    }
}
template <bool F16>
__device__ __forceinline__ float unpack_lo(unsigned int w) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2_native_t, w)[0];
    else return gd_bf2f((unsigned short)(w & 0xFFFFu));
}
template <bool F16>
__device__ __forceinline__ float unpack_hi(unsigned int w) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2_native_t, w)[1];
    else return gd_bf2f((unsigned short)(w >> 16));
}

// registers 8s..8s+7 of a 32x32 accumulator -> the 16-bit fragment of k-step s (k = accumulator ROW index,
// element j of lane half h <-> row 16s + 8(j>>2) + 4h + (j&3))
template <bool F16>
__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& a, int s) {
    const u32x4_t w = {pack2<F16>(a[8 * s + 0], a[8 * s + 1]), pack2<F16>(a[8 * s + 2], a[8 * s + 3]),
                       pack2<F16>(a[8 * s + 4], a[8 * s + 5]), pack2<F16>(a[8 * s + 6], a[8 * s + 7])};
    return __builtin_bit_cast(bf16x8_t, w);
}

__device__ __forceinline__ s16x4_t lds_tr16(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
// A fragment (row = column c of X, k = accumulator-row-ordered query index) of k-step s from X[i][c] (LDS, ld):
// element j of lane half h <-> query 16s + 8(j>>2) + 4h + (j&3)
__device__ __forceinline__ bf16x8_t read_tr_frag(const unsigned short* X, int ld, int s, int ccol, int lane) {
    const int li = lane & 15, hh = lane >> 5;
    const unsigned short* p = X + (16 * s + 4 * hh + (li >> 2)) * ld + ccol + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
    const s16x4_t lo = lds_tr16(p);             // queries 16s + 4h + 0..3
    const s16x4_t hi = lds_tr16(p + 8 * ld);    // queries 16s + 8 + 4h + 0..3
    const bf16x8_t f = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return f;
}

// dO tile image: rows of DOLD = CP + 32 elements (448 B at CP = 192: 28 sixteen-byte units, 28 = 12 mod 16) with the
// 16-byte chunk index XOR-swizzled by (row >> 2) & 3.  Both ways the tile is read are then bank-conflict free:
//   plain 16-byte reads, 16 rows x one chunk per pass : units {0,12,8,4} (row & 3) + {c^0..c^3} (row >> 2) -> 16 distinct;
//   transpose reads, 4 rows x 4 chunks per 32-lane pass: units {0,12,8,4} + {g..g+3}              -> 16 distinct
// (with plain 400-byte rows the transpose reads ran 2-way conflicted on half the banks: PMC SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int do_off(int row, int chunk, int ld) { return row * ld + ((chunk ^ ((row >> 2) & 3)) << 3); }
__device__ __forceinline__ bf16x8_t read_tr_frag_sw(const unsigned short* X, int ld, int s, int ct, int lane) {
    const int li = lane & 15, hh = lane >> 5;
    const int row = 16 * s + 4 * hh + (li >> 2);
    const int chunk = 4 * ct + 2 * ((lane >> 4) & 1) + ((li & 3) >> 1), sub = 4 * (li & 1);
    const s16x4_t lo = lds_tr16(X + do_off(row, chunk, ld) + sub);          // queries 16s + 4h + 0..3
    const s16x4_t hi = lds_tr16(X + do_off(row + 8, chunk, ld) + sub);      // queries 16s + 8 + 4h + 0..3
    const bf16x8_t f = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return f;
}

constexpr int B_QLD = 40;   // Q tile rows [i][32 d] (80 B): 16-B reads

}  // namespace pam
