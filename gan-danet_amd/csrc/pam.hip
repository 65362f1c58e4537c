// Fused (flash-style) PAM attention for gfx950, bf16 MFMA with fp32 softmax statistics.
//
//   energy[i][j] = sum_d q[i][d] k[j][d]   (no 1/sqrt(d): generator.py:117)
//   P = softmax_j(energy),  O[c][i] = sum_j V[c][j] P[i][j],  out = gamma * O + x   (generator.py:118-122)
//
// The N x N matrices are never materialised.  d is zero-padded to 32, C to a multiple of 32 (Cp).
//
// Forward: one workgroup = 4 waves = 128 queries (32 per wave), K/V streamed in 64-key tiles through LDS.
//   S^T = K Q^T is computed with the KEY index on the accumulator rows and the QUERY on the lane, so
//   * the row softmax is lane-local (+ one cross-half shuffle),
//   * the probability tile is already the B operand of O^T = V P^T  (accumulator-as-operand, no LDS trip),
//   * O^T comes out with the query on the lane: the online rescale is lane-local and the store to the NCHW
//     output is coalesced along pixels.
// Backward: one workgroup = 4 waves = 128 keys (32 per wave) holding dV^T and dK^T in accumulators while it
//   sweeps the queries in 32-row tiles; S and dP are computed with the key on the lane, so P and dS are
//   directly the B operands of dV^T += dO^T P and dK^T += Q^T dS; only dS crosses LDS (for dQ += dS K, which is
//   accumulated across workgroups with fp32 atomics in 128-byte row segments).
#include <stdlib.h>

#include "common.h"
#include "tile_mma.h"
#include "../../include/gandanet.h"

namespace {

using gd::acc_row;
using gd::bf16x8_native_t;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // staging registers: first-class vectors, never
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));   // demoted to scratch like arrays of uint4 structs

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ f32x16_t mfma_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_native_t, a),
                                                   __builtin_bit_cast(bf16x8_native_t, b), c, 0, 0, 0);
}

// registers 8s..8s+7 of a 32x32 accumulator -> the bf16 fragment of k-step s (k = accumulator ROW index,
// element j of lane half h <-> row 16s + 8(j>>2) + 4h + (j&3))
__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& a, int s) {
    const u32x4_t w = {gd_pack_bf2(a[8 * s + 0], a[8 * s + 1]), gd_pack_bf2(a[8 * s + 2], a[8 * s + 3]),
                       gd_pack_bf2(a[8 * s + 4], a[8 * s + 5]), gd_pack_bf2(a[8 * s + 6], a[8 * s + 7])};
    return __builtin_bit_cast(bf16x8_t, w);
}

// A/B fragment whose k runs over an accumulator-row-ordered index stored contiguously in an LDS row:
// elements [base + 4h .. +3] and [base + 8 + 4h .. +3]  (two 8-byte reads)
__device__ __forceinline__ bf16x8_t read_perm_frag(const unsigned short* row, int base, int h) {
    const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(row + base + 4 * h);
    const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(row + base + 8 + 4 * h);
    const u32x4_t w = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8_t, w);
}

// =====================================================================================================
// forward
// =====================================================================================================
constexpr int F_KT = 64;    // keys per tile
constexpr int F_KLD = 40;   // K row: 32 d + 8 pad (80 B) -> conflict-free 16-B fragment reads
constexpr int F_VLD = 72;   // V row: 64 keys + 8 pad (144 B) -> conflict-free 16-B fragment reads (keys perm16-ordered)

template <int CT>
__global__ __launch_bounds__(256, 2) void pam_fwd_kernel(const unsigned short* __restrict__ qt,
                                                        const unsigned short* __restrict__ kt,
                                                        const unsigned short* __restrict__ v, int N, int Npad, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ x,
                                                        long x_bs, float* __restrict__ out, long out_bs,
                                                        float* __restrict__ o_attn, float* __restrict__ lse) {
    constexpr int CP = CT * 32;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[F_KT * F_KLD];
    __shared__ __attribute__((aligned(16))) unsigned short Vs[CP * F_VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;

    const unsigned short* ktb = kt + (long)b * Npad * 32;
    const unsigned short* vb = v + (long)b * CP * Npad;

    // Q as the B operand of S^T = K Q^T: lane holds Q[query r][d = 16s + 8h + j]
    bf16x8_t qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        qf[s] = *reinterpret_cast<const bf16x8_t*>(qt + ((long)b * Npad + q0 + r) * 32 + s * 16 + 8 * h);

    f32x16_t o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[ct][e] = 0.f;
    float m = -1e30f, l = 0.f;  // running max (log2 domain) and sum of this lane's query

    const int nkt = (N + F_KT - 1) / F_KT;

    // staging registers: K tile = 256 x 16 B; V tile = CP*8 chunks of 16 B = CT per thread
    u32x4_t kreg;
    u32x4_t vreg[CT];
    const int k_key = tid >> 2, k_chunk = tid & 3;
    auto load_tile = [&](int t) {
        const int k0 = t * F_KT;
        kreg = *reinterpret_cast<const u32x4_t*>(ktb + (long)(k0 + k_key) * 32 + k_chunk * 8);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int idx = tid + i * 256;
            const int c = idx >> 3, qd = idx & 7;
            vreg[i] = *reinterpret_cast<const u32x4_t*>(vb + (long)c * Npad + k0 + qd * 8);
        }
    };
    auto store_tile = [&]() {
        *reinterpret_cast<u32x4_t*>(Ks + k_key * F_KLD + k_chunk * 8) = kreg;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int idx = tid + i * 256;
            const int c = idx >> 3, qd = idx & 7;
            *reinterpret_cast<u32x4_t*>(Vs + c * F_VLD + qd * 8) = vreg[i];   // 144-B rows: 16-byte aligned
        }
    };

    load_tile(0);
    store_tile();
    __syncthreads();

    for (int t = 0; t < nkt; ++t) {
        if (t + 1 < nkt) load_tile(t + 1);

        // ---- S^T tiles (keys on rows, query on the lane) ----
        f32x16_t sacc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t kf =
                    *reinterpret_cast<const bf16x8_t*>(Ks + (sub * 32 + r) * F_KLD + s * 16 + 8 * h);
                sacc[sub] = mfma_bf16(kf, qf[s], sacc[sub]);
            }
        }
        // ---- online softmax (log2 domain): m, l track max and sum of s*log2(e) ----
        if ((t + 1) * F_KT > N) {   // wave-uniform: only the last tile masks padded keys
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((t * F_KT + sub * 32 + acc_row(e, h)) >= N) sacc[sub][e] = -1e30f;
        }
        float mloc = sacc[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, sacc[sub][e]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * LOG2E;
        const float m_new = fmaxf(m, mloc);
        if (__any(m_new > m)) {  // wave-uniform: skip the O rescale when no query's max moved
            const float alpha = gd_exp2_fast(m - m_new);
            l *= alpha;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[ct][e] *= alpha;
            m = m_new;
        }
        float lsum = 0.f;
        const float neg_m = -m;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = gd_exp2_fast(fmaf(sacc[sub][e], LOG2E, neg_m));
                sacc[sub][e] = p;
                lsum += p;
            }
        lsum += __shfl_xor(lsum, 32, 64);
        l += lsum;

        // ---- O^T += V P^T : A = V rows (channel), k = keys in accumulator-row order; B = P fragments ----
        bf16x8_t pf[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s) pf[sub][s] = pack_frag(sacc[sub], s);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const unsigned short* vrow = Vs + (ct * 32 + r) * F_VLD;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vrow + sub * 32 + s * 16 + 8 * h);
                    o[ct] = mfma_bf16(vf, pf[sub][s], o[ct]);
                }
        }
        __syncthreads();
        if (t + 1 < nkt) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: O / l, residual, log-sum-exp ----
    const int qi = q0 + r;
    if (qi < N) {
        const float inv_l = 1.f / l;
        const float g = *gamma;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = ct * 32 + acc_row(e, h);
                if (c < C) {
                    const float val = o[ct][e] * inv_l;
                    o_attn[((long)b * C + c) * N + qi] = val;
                    out[(long)b * out_bs + (long)c * N + qi] = fmaf(g, val, x[(long)b * x_bs + (long)c * N + qi]);
                }
            }
        if (h == 0) lse[(long)b * N + qi] = (m + log2f(l)) * LN2;
    }
}

// =====================================================================================================
// forward, LDS-DMA variant: K/V tiles go global -> LDS directly (global_load_lds_dwordx4, no staging registers,
// no store phase) into a 2-deep ring; one barrier per key tile.  A DMA wave-instruction writes 64 lanes x 16 B
// = 1 KiB of CONTIGUOUS LDS, so the padded row layouts are kept by letting one lane in five (K rows: 4 data
// chunks + 1 pad) / nine (V rows: 8 + 1) fetch a duplicate chunk into the pad slot.
// =====================================================================================================
constexpr int D_KROWCH = 5;                 // 16-byte chunks per K row (80 B)
constexpr int D_VROWCH = 9;                 // 16-byte chunks per V row (144 B)
constexpr int D_VLD = D_VROWCH * 8;         // 72 elements

// NW waves = NW*32 queries per workgroup share each staged K/V tile (8 waves: half the L2 -> LDS streaming per query)
// KT keys per staged tile (64 or 128: fewer barriers and DMA issue rounds per key, more independent work per wave)
template <int CT, int NW, int KT = 64>
__global__ __launch_bounds__(NW * 64, 2) void pam_fwd_dma_kernel(const unsigned short* __restrict__ qt,
                                                            const unsigned short* __restrict__ kt,
                                                            const unsigned short* __restrict__ v, int N, int Npad, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ x,
                                                            long x_bs, float* __restrict__ out, long out_bs,
                                                            float* __restrict__ o_attn, float* __restrict__ lse) {
    constexpr int CP = CT * 32;
    constexpr int NSUB = KT / 32;                   // 32-key sub-tiles per staged tile
    constexpr int VROWCH = KT / 8 + 1;              // 16-byte chunks per V row (data + 1 pad)
    constexpr int VLD = KT + 8;
    constexpr int KCH = KT * D_KROWCH;              // chunks of K
    constexpr int NCH = KCH + CP * VROWCH;          // + V chunks
    constexpr int NPIECE = (NCH + 63) / 64;         // 1-KiB DMA pieces per tile
    constexpr int PPW = (NPIECE + NW - 1) / NW;     // pieces per wave
    constexpr int TILE = NPIECE * 64 * 8;           // elements per ring slot
    __shared__ __attribute__((aligned(16))) unsigned short ring[2 * TILE];   // the ONLY LDS object (DMA + ds_read)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;
    const unsigned short* ktb = kt + (long)b * Npad * 32;
    const unsigned short* vb = v + (long)b * CP * Npad;

    bf16x8_t qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        qf[s] = *reinterpret_cast<const bf16x8_t*>(qt + ((long)b * Npad + q0 + r) * 32 + s * 16 + 8 * h);

    // ---- DMA plan of this lane: source of its chunk in tile 0 and elements to advance per tile ----
    const unsigned short* src[PPW];
    int adv[PPW];
    bool live[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        const int c = piece * 64 + lane;
        live[i] = piece < NPIECE && c < NCH;
        if (c < KCH) {
            const int row = c / D_KROWCH, part = c - row * D_KROWCH;
            src[i] = ktb + (long)row * 32 + (part < 4 ? part : 3) * 8;
            adv[i] = KT * 32;
        } else {
            const int c2 = (c < NCH ? c : NCH - 1) - KCH;
            const int row = c2 / VROWCH, part = c2 - row * VROWCH;
            src[i] = vb + (long)row * Npad + (part < KT / 8 ? part : KT / 8 - 1) * 8;
            adv[i] = KT;
        }
    }
    auto dma_tile = [&](int t, int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (live[i]) {
                unsigned short* dst = ring + slot * TILE + (wave + NW * i) * 512;   // wave-uniform piece base
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)t * adv[i]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    f32x16_t o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[ct][e] = 0.f;
    float m = -1e30f, l = 0.f;
    const int nkt = (N + KT - 1) / KT;

    dma_tile(0, 0);
    for (int t = 0; t < nkt; ++t) {
        // an LDS-DMA is ordered for other waves' ds_reads only by the ISSUING wave's vmcnt wait followed by a
        // barrier; hipcc does not emit that wait for us inside the loop (checked in the .s), so it is explicit
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // tile t landed everywhere, and slot (t+1)&1 is no longer being read
        if (t + 1 < nkt) dma_tile(t + 1, (t + 1) & 1);
        const unsigned short* Ks = ring + (t & 1) * TILE;
        const unsigned short* Vs = Ks + KCH * 8;

        f32x16_t sacc[NSUB];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (sub * 32 + r) * F_KLD + s * 16 + 8 * h);
                sacc[sub] = mfma_bf16(kf, qf[s], sacc[sub]);
            }
        }
        if ((t + 1) * KT > N) {   // wave-uniform: only the last tile masks padded keys
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((t * KT + sub * 32 + acc_row(e, h)) >= N) sacc[sub][e] = -1e30f;
        }
        float mloc = sacc[0][0];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, sacc[sub][e]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * LOG2E;
        const float m_new = fmaxf(m, mloc);
        if (__any(m_new > m)) {
            const float alpha = gd_exp2_fast(m - m_new);
            l *= alpha;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[ct][e] *= alpha;
            m = m_new;
        }
        float lsum = 0.f;
        const float neg_m = -m;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = gd_exp2_fast(fmaf(sacc[sub][e], LOG2E, neg_m));
                sacc[sub][e] = p;
                lsum += p;
            }
        lsum += __shfl_xor(lsum, 32, 64);
        l += lsum;

        bf16x8_t pf[NSUB][2];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s) pf[sub][s] = pack_frag(sacc[sub], s);
        // V fragments are read in batches of FB ahead of the MFMAs that consume them (the scheduler is pinned with
        // sched_group_barrier: FB LDS reads, then FB MFMAs), so one LDS round trip is paid per batch, not per MFMA
        constexpr int FB = 8, NF = 2 * NSUB * CT;
#pragma unroll
        for (int f0 = 0; f0 < NF; f0 += FB) {
            bf16x8_t vf[FB];
#pragma unroll
            for (int i = 0; i < FB; ++i) {
                const int f = f0 + i;
                if (f < NF) {
                    const int ct = f / (2 * NSUB), sub = (f >> 1) % NSUB, s2 = f & 1;
                    vf[i] = *reinterpret_cast<const bf16x8_t*>(Vs + (ct * 32 + r) * VLD + sub * 32 + s2 * 16 + 8 * h);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, FB, 0);   // DS reads
#pragma unroll
            for (int i = 0; i < FB; ++i) {
                const int f = f0 + i;
                if (f < NF) {
                    const int ct = f / (2 * NSUB), sub = (f >> 1) % NSUB, s2 = f & 1;
                    o[ct] = mfma_bf16(vf[i], pf[sub][s2], o[ct]);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, FB, 0);   // MFMAs
        }
    }

    const int qi = q0 + r;
    if (qi < N) {
        const float inv_l = 1.f / l;
        const float g = *gamma;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = ct * 32 + acc_row(e, h);
                if (c < C) {
                    const float val = o[ct][e] * inv_l;
                    o_attn[((long)b * C + c) * N + qi] = val;
                    out[(long)b * out_bs + (long)c * N + qi] = fmaf(g, val, x[(long)b * x_bs + (long)c * N + qi]);
                }
            }
        if (h == 0) lse[(long)b * N + qi] = (m + log2f(l)) * LN2;
    }
}

// =====================================================================================================
// backward, part 1: dK^T and dV^T  (key-parallel; a workgroup owns NW*32 keys and sweeps the queries)
// =====================================================================================================
constexpr int B_QLD = 40;   // Q tile rows [i][32 d] (80 B): 16-B reads
constexpr int B_TLD = 36;   // transposed tiles rows [..][32 i] (72 B): 8-B reads, conflict free

// VLDS = true keeps this workgroup's V rows in LDS instead of 8*CT registers per lane (Cp = 192 would not fit 256)
template <int CT, int NW, bool VLDS = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void pam_bwd_dkv_kernel(
    const unsigned short* __restrict__ qt, const unsigned short* __restrict__ kt, const unsigned short* __restrict__ qn,
    const unsigned short* __restrict__ vt, const unsigned short* __restrict__ dot_, const unsigned short* __restrict__ don,
    const float* __restrict__ lse, const float* __restrict__ delta, int N, int Npad, float* __restrict__ dkn,
    float* __restrict__ dv) {
    constexpr int CP = CT * 32;
    constexpr int NT = NW * 64;
    constexpr int DLD = CP + 8;                    // dO tile rows [i][CP c] (+16 B): 16-B reads conflict free
    constexpr int NCHUNK = 256 + 256 * CT;         // 16-byte chunks staged per query tile
    constexpr int NPRE = (NCHUNK + NT - 1) / NT;   // chunks per thread
    __shared__ __attribute__((aligned(16))) unsigned short Qs[32 * B_QLD];
    __shared__ __attribute__((aligned(16))) unsigned short QTs[32 * B_TLD];
    __shared__ __attribute__((aligned(16))) unsigned short dOs[32 * DLD];
    __shared__ __attribute__((aligned(16))) unsigned short dOTs[CP * B_TLD];
    __shared__ __attribute__((aligned(16))) unsigned short Vls[VLDS ? NW * 32 * DLD : 8];   // V rows [key][CP c]
    __shared__ float Ls[32], Ds[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * (NW * 32) + wave * 32;  // this wave's 32 keys
    const long nb = (long)b * Npad;

    // ---- persistent per-wave operands: K (B operand of S = Q K^T), V (B operand of dP = dO V^T) ----
    bf16x8_t kfB[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) kfB[s] = *reinterpret_cast<const bf16x8_t*>(kt + (nb + j0 + r) * 32 + s * 16 + 8 * h);
    bf16x8_t vfB[VLDS ? 1 : 2 * CT];
    if constexpr (VLDS) {
        // the workgroup's NW*32 keys x CP channels, once: 16-byte chunks, rows padded to DLD (conflict-free reads)
        const unsigned short* vsrc = vt + (nb + (long)blockIdx.x * (NW * 32)) * CP;
        for (int c = tid; c < NW * 32 * (CP / 8); c += NT) {
            const int row = c / (CP / 8), ch = c - row * (CP / 8);
            *reinterpret_cast<u32x4_t*>(Vls + row * DLD + ch * 8) =
                *reinterpret_cast<const u32x4_t*>(vsrc + (long)row * CP + ch * 8);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 2 * CT; ++s)
            vfB[s] = *reinterpret_cast<const bf16x8_t*>(vt + (nb + j0 + r) * CP + s * 16 + 8 * h);
    }

    f32x16_t dvacc[CT], dkacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) dkacc[e] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) dvacc[ct][e] = 0.f;

    const bool key_ok = (j0 + r) < N;
    const bool need_mask = (int)(blockIdx.x + 1) * (NW * 32) > N;   // uniform over the workgroup
    const int nqt = (N + 31) / 32;

    // ---- staging: global -> registers (prefetch, issued before the MFMAs of the previous tile) -> LDS ----
    u32x4_t pre[NPRE];
    float pre_s = 0.f;
    auto load_tile = [&](int qtile) {
        const int i0 = qtile * 32;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int c = tid + k * NT;
            if (c < 128) {                                   // Q rows [i][32 d]
                pre[k] = *reinterpret_cast<const u32x4_t*>(qt + (nb + i0 + (c >> 2)) * 32 + (c & 3) * 8);
            } else if (c < 256) {                            // Q^T rows [d][32 i]
                const int c2 = c - 128;
                pre[k] = *reinterpret_cast<const u32x4_t*>(qn + ((long)b * 32 + (c2 >> 2)) * Npad + i0 + (c2 & 3) * 8);
            } else if (c < 256 + 128 * CT) {                 // dO rows [i][CP c]
                const int c2 = c - 256;
                const int i = c2 / (4 * CT), ch = c2 - i * (4 * CT);
                pre[k] = *reinterpret_cast<const u32x4_t*>(dot_ + (nb + i0 + i) * CP + ch * 8);
            } else if (c < NCHUNK) {                         // dO^T rows [c][32 i]
                const int c2 = c - 256 - 128 * CT;
                pre[k] = *reinterpret_cast<const u32x4_t*>(don + ((long)b * CP + (c2 >> 2)) * Npad + i0 + (c2 & 3) * 8);
            }
        }
        if (tid < 64) {
            const int i = i0 + (tid & 31);
            pre_s = i < N ? -(tid < 32 ? lse[(long)b * N + i] : delta[(long)b * N + i]) : 0.f;   // negated once here
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int c = tid + k * NT;
            if (c < 128) {
                *reinterpret_cast<u32x4_t*>(Qs + (c >> 2) * B_QLD + (c & 3) * 8) = pre[k];
            } else if (c < 256) {
                const int c2 = c - 128;
                u32x2_t* dst = reinterpret_cast<u32x2_t*>(QTs + (c2 >> 2) * B_TLD + (c2 & 3) * 8);
                dst[0] = u32x2_t{pre[k].x, pre[k].y};
                dst[1] = u32x2_t{pre[k].z, pre[k].w};
            } else if (c < 256 + 128 * CT) {
                const int c2 = c - 256;
                const int i = c2 / (4 * CT), ch = c2 - i * (4 * CT);
                *reinterpret_cast<u32x4_t*>(dOs + i * DLD + ch * 8) = pre[k];
            } else if (c < NCHUNK) {
                const int c2 = c - 256 - 128 * CT;
                u32x2_t* dst = reinterpret_cast<u32x2_t*>(dOTs + (c2 >> 2) * B_TLD + (c2 & 3) * 8);
                dst[0] = u32x2_t{pre[k].x, pre[k].y};
                dst[1] = u32x2_t{pre[k].z, pre[k].w};
            }
        }
        if (tid < 32) Ls[tid] = pre_s;
        else if (tid < 64) Ds[tid - 32] = pre_s;
    };

    load_tile(0);
    store_tile();
    __syncthreads();

    for (int qtile = 0; qtile < nqt; ++qtile) {
        const int i0 = qtile * 32;
        if (qtile + 1 < nqt) load_tile(qtile + 1);

        // ---- S' = Q K^T - lse (rows i, lane j) and dP - delta = dO V^T - delta -----------------------------
        f32x16_t sacc, dpacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = Ls[acc_row(e, h)];    // -lse   (row constants as initial accumulators)
            dpacc[e] = Ds[acc_row(e, h)];   // -delta
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(Qs + r * B_QLD + s * 16 + 8 * h);
            sacc = mfma_bf16(qa, kfB[s], sacc);
        }
#pragma unroll
        for (int s = 0; s < 2 * CT; ++s) {
            const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(dOs + r * DLD + s * 16 + 8 * h);
            if constexpr (VLDS) {
                const bf16x8_t vb = *reinterpret_cast<const bf16x8_t*>(Vls + (wave * 32 + r) * DLD + s * 16 + 8 * h);
                dpacc = mfma_bf16(da, vb, dpacc);
            } else {
                dpacc = mfma_bf16(da, vfB[s], dpacc);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = gd_exp2_fast(sacc[e] * LOG2E);   // P
        if (need_mask || i0 + 32 > N) {   // workgroup-uniform: padded keys in this block, or the ragged last query tile
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (!(key_ok && (i0 + acc_row(e, h)) < N)) sacc[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) dpacc[e] *= sacc[e];   // dS = P (dP - delta)
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            pf[s] = pack_frag(sacc, s);
            dsf[s] = pack_frag(dpacc, s);
        }
        // ---- dV^T[c][j] += dO^T[c][i] P[i][j] ;  dK^T[d][j] += Q^T[d][i] dS[i][j] ---------------------------
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const unsigned short* row = dOTs + (ct * 32 + r) * B_TLD;
#pragma unroll
            for (int s = 0; s < 2; ++s) dvacc[ct] = mfma_bf16(read_perm_frag(row, s * 16, h), pf[s], dvacc[ct]);
        }
        {
            const unsigned short* row = QTs + r * B_TLD;
#pragma unroll
            for (int s = 0; s < 2; ++s) dkacc = mfma_bf16(read_perm_frag(row, s * 16, h), dsf[s], dkacc);
        }
        __syncthreads();  // everyone is done reading this tile
        if (qtile + 1 < nqt) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- write dV^T (channel-major, coalesced along keys) and dK^T ----------------------------------------
    const int j = j0 + r;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) dv[((long)b * CP + ct * 32 + acc_row(e, h)) * Npad + j] = dvacc[ct][e];
#pragma unroll
    for (int e = 0; e < 16; ++e) dkn[((long)b * 32 + acc_row(e, h)) * Npad + j] = dkacc[e];
}

// =====================================================================================================
// backward, part 1c: dK^T / dV^T, transpose-read variant.  Only Q [i][d] and dO [i][c] are staged per query
// tile; the "transposed" A operands of dV^T += dO^T P and dK^T += Q^T dS are taken from the SAME images with
// ds_read_b64_tr_b16 (4 query rows x 16 channel columns per 16-lane group), so the q^T / dO^T copies, half of
// the staging traffic and half of the tile LDS disappear.  V rows of the workgroup's 128 keys live in LDS.
// 4 waves per workgroup and two independent workgroups per CU: their phases drift apart, so one's MFMA
// segments overlap the other's softmax / staging segments instead of all eight waves meeting at one barrier.
// =====================================================================================================
typedef short s16x4_t __attribute__((ext_vector_type(4)));
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

template <int CT>
__global__ __launch_bounds__(256, 2) void pam_bwd_dkv3_kernel(
    const unsigned short* __restrict__ qt, const unsigned short* __restrict__ kt, const unsigned short* __restrict__ vt,
    const unsigned short* __restrict__ dot_, const float* __restrict__ lse, const float* __restrict__ delta, int N,
    int Npad, float* __restrict__ dkn, float* __restrict__ dv) {
    constexpr int CP = CT * 32;
    constexpr int NT = 256;
    constexpr int DLD = CP + 8;
    constexpr int NCHUNK = 128 + 128 * CT;
    constexpr int NPRE = (NCHUNK + NT - 1) / NT;
    constexpr bool RAGGED = (NCHUNK % NT) != 0;
    __shared__ __attribute__((aligned(16))) unsigned short Vls[128 * DLD];
    __shared__ __attribute__((aligned(16))) unsigned short Qs[32 * B_QLD];
    __shared__ __attribute__((aligned(16))) unsigned short dOs[32 * DLD];
    __shared__ float Ls[32], Ds[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * 128 + wave * 32;
    const long nb = (long)b * Npad;

    bf16x8_t kfB[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) kfB[s] = *reinterpret_cast<const bf16x8_t*>(kt + (nb + j0 + r) * 32 + s * 16 + 8 * h);
    {
        const unsigned short* vsrc = vt + (nb + (long)blockIdx.x * 128) * CP;
        for (int c = tid; c < 128 * (CP / 8); c += NT) {
            const int row = c / (CP / 8), ch = c - row * (CP / 8);
            *reinterpret_cast<u32x4_t*>(Vls + row * DLD + ch * 8) =
                *reinterpret_cast<const u32x4_t*>(vsrc + (long)row * CP + ch * 8);
        }
    }

    f32x16_t dvacc[CT], dkacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) dkacc[e] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) dvacc[ct][e] = 0.f;

    const bool key_ok = (j0 + r) < N;
    const bool need_mask = (int)(blockIdx.x + 1) * 128 > N;
    const int nqt = (N + 31) / 32;

    // staging plan (fixed per thread): source at tile 0, elements per tile, LDS slot
    const unsigned short* src[NPRE];
    int step[NPRE];
    unsigned short* dst[NPRE];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c = tid + k * NT;
        if (RAGGED && k == NPRE - 1 && c >= NCHUNK) c = 0;
        if (c < 128) {
            src[k] = qt + (nb + (c >> 2)) * 32 + (c & 3) * 8;  step[k] = 32 * 32;
            dst[k] = Qs + (c >> 2) * B_QLD + (c & 3) * 8;
        } else {
            const int c2 = c - 128;
            const int i = c2 / (4 * CT), ch = c2 - i * (4 * CT);
            src[k] = dot_ + (nb + i) * CP + ch * 8;  step[k] = 32 * CP;
            dst[k] = dOs + i * DLD + ch * 8;
        }
    }
    const bool last_ok = !RAGGED || (tid + (NPRE - 1) * NT) < NCHUNK;

    u32x4_t pre[NPRE];
    float pre_s = 0.f;
    auto load_tile = [&](int qtile) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
            pre[k] = *reinterpret_cast<const u32x4_t*>(src[k] + (long)qtile * step[k]);
        if (tid < 64) {
            const int i = qtile * 32 + (tid & 31);
            pre_s = i < N ? -(tid < 32 ? lse[(long)b * N + i] : delta[(long)b * N + i]) : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
            if (k < NPRE - 1 || last_ok) *reinterpret_cast<u32x4_t*>(dst[k]) = pre[k];
        if (tid < 32) Ls[tid] = pre_s;
        else if (tid < 64) Ds[tid - 32] = pre_s;
    };

    load_tile(0);
    store_tile();
    __syncthreads();

    for (int qtile = 0; qtile < nqt; ++qtile) {
        const int i0 = qtile * 32;
        if (qtile + 1 < nqt) load_tile(qtile + 1);

        f32x16_t sacc, dpacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = Ls[acc_row(e, h)];
            dpacc[e] = Ds[acc_row(e, h)];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(Qs + r * B_QLD + s * 16 + 8 * h);
            sacc = mfma_bf16(qa, kfB[s], sacc);
        }
#pragma unroll
        for (int s = 0; s < 2 * CT; ++s) {
            const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(dOs + r * DLD + s * 16 + 8 * h);
            const bf16x8_t vb = *reinterpret_cast<const bf16x8_t*>(Vls + (wave * 32 + r) * DLD + s * 16 + 8 * h);
            dpacc = mfma_bf16(da, vb, dpacc);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = gd_exp2_fast(sacc[e] * LOG2E);   // P
        if (need_mask || i0 + 32 > N) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (!(key_ok && (i0 + acc_row(e, h)) < N)) sacc[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) dpacc[e] *= sacc[e];   // dS
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            pf[s] = pack_frag(sacc, s);
            dsf[s] = pack_frag(dpacc, s);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) dvacc[ct] = mfma_bf16(read_tr_frag(dOs, DLD, s, ct * 32, lane), pf[s], dvacc[ct]);
#pragma unroll
        for (int s = 0; s < 2; ++s) dkacc = mfma_bf16(read_tr_frag(Qs, B_QLD, s, 0, lane), dsf[s], dkacc);
        __syncthreads();
        if (qtile + 1 < nqt) {
            store_tile();
            __syncthreads();
        }
    }

    const int j = j0 + r;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) dv[((long)b * CP + ct * 32 + acc_row(e, h)) * Npad + j] = dvacc[ct][e];
#pragma unroll
    for (int e = 0; e < 16; ++e) dkn[((long)b * 32 + acc_row(e, h)) * Npad + j] = dkacc[e];
}

// =====================================================================================================
// backward, part 2: dQ^T  (query-parallel like the forward: a workgroup owns NW*32 queries and streams keys)
//   S^T[j][i], dP^T[j][i] with the key on the accumulator rows and the query on the lane (lse/delta are lane
//   constants), dS^T is the B operand of dQ^T[d][i] += K^T[d][j] dS^T[j][i]; dQ^T accumulates in registers --
//   no atomics, deterministic, and it is stored channel-major (the layout the projection gradients read).
//   K [key][d], K^T [d][key] (keys perm16-ordered) and V [key][c] tiles arrive by LDS-DMA into a 2-slot ring.
// =====================================================================================================
template <int CT, int NW, int KT>
__global__ __launch_bounds__(NW * 64, 2) void pam_bwd_dq_kernel(
    const unsigned short* __restrict__ qt, const unsigned short* __restrict__ kt, const unsigned short* __restrict__ kn,
    const unsigned short* __restrict__ vt, const unsigned short* __restrict__ dot_, const float* __restrict__ lse,
    const float* __restrict__ delta, int N, int Npad, float* __restrict__ dqn) {
    constexpr int CP = CT * 32;
    constexpr int NSUB = KT / 32;
    constexpr int KNROWCH = KT / 8 + 1, KNLD = KT + 8;     // K^T rows [d][KT keys] + 16 B pad
    constexpr int VROWCH = CP / 8 + 1, VLD = CP + 8;       // V rows [key][CP c] + 16 B pad
    constexpr int KCH = KT * D_KROWCH;
    constexpr int NCHN = 32 * KNROWCH;
    constexpr int VCH = KT * VROWCH;
    constexpr int NCH = KCH + NCHN + VCH;
    constexpr int NPIECE = (NCH + 63) / 64;
    constexpr int PPW = (NPIECE + NW - 1) / NW;
    constexpr int TILE = NPIECE * 64 * 8;
    __shared__ __attribute__((aligned(16))) unsigned short ring[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;
    const long nb = (long)b * Npad;
    const int qi = q0 + r;

    bf16x8_t qf[2], dof[2 * CT];
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qt + (nb + q0 + r) * 32 + s * 16 + 8 * h);
#pragma unroll
    for (int s = 0; s < 2 * CT; ++s)
        dof[s] = *reinterpret_cast<const bf16x8_t*>(dot_ + (nb + q0 + r) * CP + s * 16 + 8 * h);
    const float nlse = qi < N ? -lse[(long)b * N + qi] * LOG2E : 0.f;   // log2 domain
    const float ndelta = qi < N ? -delta[(long)b * N + qi] : 0.f;

    // ---- DMA plan of this lane ----
    const unsigned short* src[PPW];
    int adv[PPW];
    bool live[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        int c = piece * 64 + lane;
        live[i] = piece < NPIECE && c < NCH;
        if (c >= NCH) c = NCH - 1;
        if (c < KCH) {
            const int row = c / D_KROWCH, part = c - row * D_KROWCH;
            src[i] = kt + (nb + row) * 32 + (part < 4 ? part : 3) * 8;
            adv[i] = KT * 32;
        } else if (c < KCH + NCHN) {
            const int c2 = c - KCH;
            const int row = c2 / KNROWCH, part = c2 - row * KNROWCH;
            src[i] = kn + ((long)b * 32 + row) * Npad + (part < KT / 8 ? part : KT / 8 - 1) * 8;
            adv[i] = KT;
        } else {
            const int c3 = c - KCH - NCHN;
            const int row = c3 / VROWCH, part = c3 - row * VROWCH;
            src[i] = vt + (nb + row) * CP + (part < CP / 8 ? part : CP / 8 - 1) * 8;
            adv[i] = KT * CP;
        }
    }
    auto dma_tile = [&](int t, int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (live[i]) {
                unsigned short* dst = ring + slot * TILE + (wave + NW * i) * 512;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)t * adv[i]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    f32x16_t dq;
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[e] = 0.f;

    const int nkt = (N + KT - 1) / KT;
    dma_tile(0, 0);
    for (int t = 0; t < nkt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // see pam_fwd_dma_kernel: the DMA wait is ours to place
        __syncthreads();
        if (t + 1 < nkt) dma_tile(t + 1, (t + 1) & 1);
        const unsigned short* Ks = ring + (t & 1) * TILE;
        const unsigned short* KNs = Ks + KCH * 8;
        const unsigned short* VTs = KNs + NCHN * 8;
        const bool tail = (t + 1) * KT > N;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            f32x16_t sacc, dpacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                sacc[e] = 0.f;
                dpacc[e] = ndelta;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (sub * 32 + r) * F_KLD + s * 16 + 8 * h);
                sacc = mfma_bf16(kf, qf[s], sacc);
            }
            // V fragments in batches ahead of their MFMAs (one LDS round trip per batch)
            constexpr int FB = 6;
#pragma unroll
            for (int s0 = 0; s0 < 2 * CT; s0 += FB) {
                bf16x8_t va[FB];
#pragma unroll
                for (int i = 0; i < FB; ++i)
                    if (s0 + i < 2 * CT)
                        va[i] = *reinterpret_cast<const bf16x8_t*>(VTs + (sub * 32 + r) * VLD + (s0 + i) * 16 + 8 * h);
                __builtin_amdgcn_sched_group_barrier(0x100, FB, 0);
#pragma unroll
                for (int i = 0; i < FB; ++i)
                    if (s0 + i < 2 * CT) dpacc = mfma_bf16(va[i], dof[s0 + i], dpacc);
                __builtin_amdgcn_sched_group_barrier(0x008, FB, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = gd_exp2_fast(fmaf(sacc[e], LOG2E, nlse));
                dpacc[e] = p * dpacc[e];  // dS^T
            }
            if (tail) {   // wave-uniform: padded keys of the last tile contribute nothing
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((t * KT + sub * 32 + acc_row(e, h)) >= N) dpacc[e] = 0.f;
            }
            const unsigned short* krow = KNs + r * KNLD;   // K^T row d = r, keys perm16-ordered: one 16-byte read
#pragma unroll
            for (int s = 0; s < 2; ++s)
                dq = mfma_bf16(*reinterpret_cast<const bf16x8_t*>(krow + sub * 32 + s * 16 + 8 * h), pack_frag(dpacc, s), dq);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) dqn[((long)b * 32 + acc_row(e, h)) * Npad + qi] = dq[e];
}

}  // namespace

#define PAM_DISPATCH_CT(CT_, CALL)                         \
    switch (CT_) {                                         \
        case 1: { constexpr int CT = 1; CALL; } break;     \
        case 2: { constexpr int CT = 2; CALL; } break;     \
        case 3: { constexpr int CT = 3; CALL; } break;     \
        case 4: { constexpr int CT = 4; CALL; } break;     \
        case 5: { constexpr int CT = 5; CALL; } break;     \
        case 6: { constexpr int CT = 6; CALL; } break;     \
        default: gd_set_error("pam: Cp must be 32..192"); return -1; \
    }

extern "C" int gd_pam_flash_fwd(const void* qt, const void* kt, const void* v, int B, int N, int Npad, int C, int Cp,
                                const float* gamma, const float* x, long x_bs, float* out, long out_bs, float* o_attn,
                                float* lse, void* stream) {
    GD_CHECK_ARG(qt && kt && v && gamma && x && out && o_attn && lse, "gd_pam_flash_fwd: null pointer");
    GD_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && Npad >= N && Npad % 128 == 0, "gd_pam_flash_fwd: Npad must be a multiple of 128 >= N");
    GD_CHECK_ARG(C > 0 && Cp >= C && Cp % 32 == 0 && Cp <= 192, "gd_pam_flash_fwd: Cp must be a multiple of 32, C <= Cp <= 192");
    dim3 grid(Npad / 128, B);
    // default 16: LDS-DMA ring, 8 waves (256 queries) per workgroup, 128-key tiles; 8 / 1: 64-key tiles with 8 / 4
    // waves; 0: register-staged 4-wave kernel.  Measured at B=4, N=65536, C=184: 7.72 / 7.82 / 8.43 / 8.75 ms.
    static const int dma_env = getenv("GD_PAM_FWD_DMA") ? atoi(getenv("GD_PAM_FWD_DMA")) : 16;
    if (dma_env == 16 && Npad % 256 == 0) {  // 8 waves, 128-key tiles
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_dma_kernel<CT, 8, 128>), dim3(Npad / 256, B), dim3(512), 0,
                                                     (hipStream_t)stream, (const unsigned short*)qt,
                                                     (const unsigned short*)kt, (const unsigned short*)v, N, Npad, C,
                                                     gamma, x, x_bs, out, out_bs, o_attn, lse));
    } else if (dma_env == 8 && Npad % 256 == 0) {   // 8 waves = 256 queries per workgroup
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_dma_kernel<CT, 8>), dim3(Npad / 256, B), dim3(512), 0,
                                                     (hipStream_t)stream, (const unsigned short*)qt,
                                                     (const unsigned short*)kt, (const unsigned short*)v, N, Npad, C,
                                                     gamma, x, x_bs, out, out_bs, o_attn, lse));
    } else if (dma_env) {   // default: LDS-DMA ring (no staging registers, one barrier per key tile)
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_dma_kernel<CT, 4>), grid, dim3(256), 0, (hipStream_t)stream,
                                                     (const unsigned short*)qt, (const unsigned short*)kt,
                                                     (const unsigned short*)v, N, Npad, C, gamma, x, x_bs, out, out_bs,
                                                     o_attn, lse));
    } else {         // GD_PAM_FWD_DMA=0: register-staged variant (A/B reference)
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_kernel<CT>), grid, dim3(256), 0, (hipStream_t)stream,
                                                     (const unsigned short*)qt, (const unsigned short*)kt,
                                                     (const unsigned short*)v, N, Npad, C, gamma, x, x_bs, out, out_bs,
                                                     o_attn, lse));
    }
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_pam_flash_bwd(const void* qt, const void* kt, const void* qn, const void* kn, const void* vt,
                                const void* dot_, const void* don, const float* lse, const float* delta, int B, int N,
                                int Npad, int Cp, float* dqn, float* dkn, float* dv, void* stream) {
    GD_CHECK_ARG(qt && kt && kn && vt && dot_ && lse && delta && dqn && dkn && dv, "gd_pam_flash_bwd: null pointer");
    GD_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && Npad >= N && Npad % 256 == 0, "gd_pam_flash_bwd: Npad must be a multiple of 256 >= N");
    GD_CHECK_ARG(Cp > 0 && Cp % 32 == 0 && Cp <= 192, "gd_pam_flash_bwd: Cp must be a multiple of 32 <= 192");
    hipStream_t s = (hipStream_t)stream;
    // 8 waves (2 per SIMD, 256 keys per workgroup) while the accumulators fit 256 registers; Cp = 192 needs the
    // whole 512-register file: 4 waves, one per SIMD, 128 keys per workgroup
    static const int v3_env = getenv("GD_PAM_DKV_V3") ? atoi(getenv("GD_PAM_DKV_V3")) : 1;
    GD_CHECK_ARG(v3_env || (qn && don), "gd_pam_flash_bwd: qn/don are required by the GD_PAM_DKV_V3=0 dK/dV variant");
    if (v3_env) {                   // default: transpose-read variant, 4 waves, two independent workgroups per CU
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_bwd_dkv3_kernel<CT>), dim3(Npad / 128, B), dim3(256), 0, s,
                                                     (const unsigned short*)qt, (const unsigned short*)kt,
                                                     (const unsigned short*)vt, (const unsigned short*)dot_, lse, delta,
                                                     N, Npad, dkn, dv));
    } else if (Cp == 192) {         // A/B reference: 8-wave kernel staging q^T / dO^T copies; V rows in LDS at Cp = 192
        hipLaunchKernelGGL((pam_bwd_dkv_kernel<6, 8, true>), dim3(Npad / 256, B), dim3(512), 0, s,
                           (const unsigned short*)qt, (const unsigned short*)kt, (const unsigned short*)qn,
                           (const unsigned short*)vt, (const unsigned short*)dot_, (const unsigned short*)don, lse, delta,
                           N, Npad, dkn, dv);
    } else {
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_bwd_dkv_kernel<CT, 8>), dim3(Npad / 256, B), dim3(512), 0, s,
                                                     (const unsigned short*)qt, (const unsigned short*)kt,
                                                     (const unsigned short*)qn, (const unsigned short*)vt,
                                                     (const unsigned short*)dot_, (const unsigned short*)don, lse, delta,
                                                     N, Npad, dkn, dv));
    }
    // dQ: 8 waves (256 queries) per workgroup, 128-key LDS-DMA tiles; kn must be packed perm16 along the keys
    PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_bwd_dq_kernel<CT, 8, 128>), dim3(Npad / 256, B), dim3(512), 0, s,
                                                 (const unsigned short*)qt, (const unsigned short*)kt,
                                                 (const unsigned short*)kn, (const unsigned short*)vt,
                                                 (const unsigned short*)dot_, lse, delta, N, Npad, dqn));
    GD_LAUNCH_CHECK();
    return 0;
}
