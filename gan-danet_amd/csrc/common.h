// Shared device helpers for the GAN-DANet gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GD_WAVE 64

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPR (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;   // 4 bf16 = 2 VGPR
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- bf16 <-> f32 ---------------------------------------------------------------------------
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned short gd_f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float gd_bf2f(unsigned short h) {
    return __builtin_bit_cast(float, ((unsigned int)h) << 16);
}
typedef float gd_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 gd_bf16x2_t __attribute__((ext_vector_type(2)));
// two floats -> one dword of two bf16 (lo in bits 0..15): a vector convert lowers to ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int gd_pack_bf2(float lo, float hi) {
    const gd_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, gd_bf16x2_t));
}
// split-bf16 ("x3" operands, set_precision("mixed")): v = hi + lo with hi = bf16(v), lo = bf16(v - hi): 16 mantissa bits.
// Two values per dword like gd_pack_bf2.  A product of two split operands is hi*hi + lo*hi + hi*lo (+ 2^-16 relative).
__device__ __forceinline__ void gd_split_bf2(float a, float b, unsigned int& hi, unsigned int& lo) {
    hi = gd_pack_bf2(a, b);
    lo = gd_pack_bf2(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}
__device__ __forceinline__ void gd_split_bf(float a, unsigned short& hi, unsigned short& lo) {
    hi = gd_f2bf(a);
    lo = gd_f2bf(a - gd_bf2f(hi));
}
// raw v_exp_f32 (2^x) without the denormal-range fix-up sequence exp2f() expands to; callers pass x <= ~0
__device__ __forceinline__ float gd_exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- wave / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float gd_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gd_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double gd_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of one float; `red` = LDS scratch of >= blockDim/64 floats; result valid in ALL threads
__device__ __forceinline__ float gd_block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = gd_wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float gd_block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = gd_wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

// ---- error plumbing for the C ABI -------------------------------------------------------------
extern "C" void gd_set_error(const char* msg);
extern "C" int gd_get_deterministic(void);   // api.hip: 1 = no order-dependent (atomic) reductions
#define GD_CHECK_ARG(cond, msg)          \
    do {                                 \
        if (!(cond)) {                   \
            gd_set_error(msg);           \
            return -1;                   \
        }                                \
    } while (0)
#define GD_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) {                             \
            gd_set_error(hipGetErrorString(e_));            \
            return -2;                                      \
        }                                                   \
    } while (0)

static inline int gd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
