// Fused (flash-style) PAM attention for gfx950, bf16 MFMA with fp32 softmax statistics.
//
//   energy[i][j] = sum_d q[i][d] k[j][d]   (no 1/sqrt(d): generator.py:117)
//   P = softmax_j(energy),  O[c][i] = sum_j V[c][j] P[i][j],  out = gamma * O + x   (generator.py:118-122)
//
// The N x N matrices are never materialised.  d is zero-padded to 32, C to a multiple of 32 (Cp).
//
// Operand contract: q is packed PRE-SCALED by log2(e) (gd_pack_bf16 scale_imm), so every S tile comes out of the
//   MFMA in the log2 domain.  The forward feeds its running row maximum through the spare k-slot d = 31 (k holds 1.0
//   there), the backward feeds -log-sum-exp and the row term -delta in as the MFMA accumulator input: exp2 is applied
//   to the accumulator as it stands, no per-element multiply / subtract is left on the VALU.
// Forward: one workgroup = 8 waves = 256 queries (32 per wave), K/V streamed in 128-key tiles through LDS.
//   S^T = K Q^T is computed with the KEY index on the accumulator rows and the QUERY on the lane, so
//   * the row softmax is lane-local (+ one cross-half shuffle),
//   * the probability tile is already the B operand of O^T = V P^T  (accumulator-as-operand, no LDS trip),
//   * O^T comes out with the query on the lane: the online rescale is lane-local and the store to the NCHW
//     output is coalesced along pixels.
// Backward: one workgroup = 8 (or 4) waves = 256 (128) keys, 32 per wave, holding dV^T and dK^T in accumulators while
//   it sweeps the queries in 32-row tiles; S and dP are computed with the key on the lane, so P and dS are directly
//   the B operands of dV^T += dO^T P and dK^T += Q^T dS.  For dQ the dS tile is turned around through wave-private
//   LDS (transpose read), multiplied by the wave's K rows and the workgroup's sum is stored as a bf16 part per key
//   block; pam_dq_reduce_kernel adds the key blocks.  No atomics in this file's kernels.  (pam_bwd_dq_kernel: the
//   scratch-free alternative, a query-parallel kernel that recomputes S and dP.)  The DEFAULT backward is the K64 kernel
//   of pam_bwd64.hip (64 keys per wave, one wave per SIMD, fp32 atomics or deterministic parts for dQ); the kernels
//   here are its reproducible / scratch-free fallbacks (gd_pam_flash_bwd forms 2 and 3).
#include <stdlib.h>
#include <type_traits>

#include "pam_common.h"
#include "../../include/gandanet.h"

namespace {

using namespace pam;

__device__ __forceinline__ f32x16_t mfma_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) { return mfma16<false>(a, b, c); }
__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& a, int s) { return pam::pack_frag<false>(a, s); }

// =====================================================================================================
// forward
// =====================================================================================================
constexpr int F_KLD = 40;   // K row: 32 d + 8 pad (80 B) -> conflict-free 16-B fragment reads

// =====================================================================================================
// forward, LDS-DMA variant: K/V tiles go global -> LDS directly (global_load_lds_dwordx4, no staging registers,
// no store phase) into a 2-deep ring; one barrier per key tile.  A DMA wave-instruction writes 64 lanes x 16 B
// = 1 KiB of CONTIGUOUS LDS, so the padded row layouts are kept by letting one lane in five (K rows: 4 data
// chunks + 1 pad) / nine (V rows: 8 + 1) fetch a duplicate chunk into the pad slot.
// =====================================================================================================
constexpr int D_KROWCH = 5;                 // 16-byte chunks per K row (80 B)

// NW waves = NW*32 queries per workgroup share each staged K/V tile (8 waves: half the L2 -> LDS streaming per query)
// KT keys per staged tile (64 or 128: fewer barriers and DMA issue rounds per key, more independent work per wave)
// ONES: the packed V carries a row of ones in its last padded channel (Cp - 1), so the softmax denominator is
// accumulated by the same MFMAs as O (it is row 31 of the last channel tile) instead of 1 VALU add per element
template <int CT, int NW, int KT, bool ONES, bool F16>
__global__ __launch_bounds__(NW * 64, 2) void pam_fwd_dma_kernel(const unsigned short* __restrict__ qt,
                                                            const unsigned short* __restrict__ kt,
                                                            const unsigned short* __restrict__ v, int N, int Npad, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ x,
                                                            long x_bs, float* __restrict__ out, long out_bs,
                                                            float* __restrict__ o_attn, float* __restrict__ lse,
                                                            const float* __restrict__ k_sqmax,
                                                            const float* __restrict__ mshift, int* __restrict__ redo) {
    // redo != NULL without mshift: the fallback pass of the sampled-shift sweep -- only workgroups flagged there run
    if (redo && !mshift && redo[blockIdx.y * gridDim.x + blockIdx.x] == 0) return;
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

    // ---- DMA plan of this lane: 32-bit byte offset of its chunk inside the tile's K / V source.  Pieces never
    // straddle K and V (KCH is a multiple of 64), so the 64-bit base of a piece is wave-uniform (SGPRs) and only
    // these offsets live in VGPRs across the loop ----
    static_assert(KCH % 64 == 0, "a DMA piece must be all-K or all-V");
    unsigned int voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        const int c = piece * 64 + lane;
        if (piece * 64 < KCH) {
            const int row = c / D_KROWCH, part = c - row * D_KROWCH;
            voff[i] = (unsigned int)(row * 32 + (part < 4 ? part : 3) * 8) * 2u;
        } else {
            const int c2 = (c < NCH ? c : NCH - 1) - KCH;
            const int row = c2 / VROWCH, part = c2 - row * VROWCH;
            voff[i] = ((unsigned int)row * (unsigned int)Npad + (unsigned int)(part < KT / 8 ? part : KT / 8 - 1) * 8u) * 2u;
        }
    }
    auto dma_tile = [&](int t, int slot) {
        const char* kbase = reinterpret_cast<const char*>(ktb + (long)t * (KT * 32));
        const char* vbase = reinterpret_cast<const char*>(vb + (long)t * KT);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + NW * i;
            if (piece < NPIECE && piece * 64 + lane < NCH) {
                unsigned short* dst = ring + slot * TILE + piece * 512;   // wave-uniform piece base
                const char* base = piece * 64 < KCH ? kbase : vbase;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff[i]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    f32x16_t o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[ct][e] = 0.f;
    // m: running row maximum (log2 domain; q arrives pre-scaled by log2 e), kept bf16-representable and fed to the
    // S MFMAs through the spare k-slot d = 31: k carries 1.0 there (gd_pack_bf16 ones_row) and this wave's Q
    // fragment carries -m, so a tile comes out of the matrix pipe as s - m with zero-initialised accumulators and
    // no per-score subtract.  Only a tile that RAISES the maximum (wave-uniform test) pays a shift and the O
    // rescale; tile 0 always does and thereby sets m.
    float m = 0.f, l = 0.f;
    const int nkt = (N + KT - 1) / KT;

    dma_tile(0, 0);
    auto sweep = [&](auto nm) {
    constexpr bool NOMAX = decltype(nm)::value;
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
                sacc[sub] = mfma16<F16>(kf, qf[s], sacc[sub]);
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
        if constexpr (!NOMAX) {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, sacc[sub][e]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        }
        float lsum = 0.f;
        bf16x8_t pf[NSUB][2];     // P^T tile as the B operand of O^T += V P^T (accumulator-as-operand)
        if (!NOMAX && (t == 0 || __any(mloc > 0.f))) {
            // new maximum, rounded UP to the operand type (bf16 / fp16) so that it survives the trip through the Q fragment exactly
            const float want = m + (t == 0 ? mloc : fmaxf(mloc, 0.f));
            float m_new;
            unsigned short m_neg16;          // -m_new in the operand type
            if constexpr (F16) {
                // fp16: round to nearest, then step one ulp towards +inf if that fell below `want`
                _Float16 hm = (_Float16)want;
                unsigned short hb = __builtin_bit_cast(unsigned short, hm);
                if ((float)hm < want) hb = (hb & 0x8000u) ? (unsigned short)(hb - 1) : (unsigned short)(hb + 1);
                m_new = (float)__builtin_bit_cast(_Float16, hb);
                m_neg16 = hb ^ 0x8000u;
            } else {
                const unsigned int wb = __builtin_bit_cast(unsigned int, want);
                m_new = __builtin_bit_cast(float, want > 0.f ? (wb + 0xFFFFu) & 0xFFFF0000u : wb & 0xFFFF0000u);
                m_neg16 = (unsigned short)(__builtin_bit_cast(unsigned int, -m_new) >> 16);
            }
            const float shift = m_new - m;
            if (t != 0) {
                const float alpha = gd_exp2_fast(-shift);
                l *= alpha;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[ct][e] *= alpha;
            }
            m = m_new;
            if (h) qf[1][7] = (short)m_neg16;   // d = 31 lives in lane half 1
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sacc[sub][e] = gd_exp2_fast(sacc[sub][e] - shift);
                    if (!ONES) lsum += sacc[sub][e];
                }
                pf[sub][0] = pam::pack_frag<F16>(sacc[sub], 0);
                pf[sub][1] = pam::pack_frag<F16>(sacc[sub], 1);
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sacc[sub][e] = gd_exp2_fast(sacc[sub][e]);
                    if (!ONES) lsum += sacc[sub][e];
                }
                pf[sub][0] = pam::pack_frag<F16>(sacc[sub], 0);
                pf[sub][1] = pam::pack_frag<F16>(sacc[sub], 1);
            }
        }
        if (!ONES) {
            lsum += __shfl_xor(lsum, 32, 64);
            l += lsum;
        }

        // V fragments are read in batches ahead of the MFMAs that consume them (the scheduler is pinned with
        // sched_group_barrier: CT LDS reads, then CT MFMAs), so one LDS round trip is paid per batch, not per MFMA.
        // A batch is one k-step of all CT channel tiles: its MFMAs write CT different accumulators (no
        // back-to-back dependent MFMAs on one accumulator)
#pragma unroll
        for (int ks = 0; ks < 2 * NSUB; ++ks) {
            const int sub = ks >> 1, s2 = ks & 1;
            bf16x8_t vf[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                vf[ct] = *reinterpret_cast<const bf16x8_t*>(Vs + (ct * 32 + r) * VLD + sub * 32 + s2 * 16 + 8 * h);
            __builtin_amdgcn_sched_group_barrier(0x100, CT, 0);   // DS reads
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) o[ct] = mfma16<F16>(vf[ct], pf[sub][s2], o[ct]);
            __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);   // MFMAs
        }
    }
    };
    // Softmax is shift-invariant: the running maximum only guards exp2 against overflow.  |s_ij| <= |q_i| |k_j|
    // (Cauchy-Schwarz on the operands exactly as the MFMA sees them, q pre-scaled by log2 e), so when that bound stays
    // inside the exponent range of the P operand type for every query of the wave, the whole sweep runs with m = 0:
    // no row maximum, no cross-half shuffle, no test, no rescale branch (8 % of the kernel at the bench shape).
    // k_sqmax[b] = max_j |k_j|^2 comes from gd_pam_key_sqnorm_max; NULL keeps the running maximum unconditionally.
    // Sampled shift (gd_pam_row_shift): ANY per-query shift m_i gives the same softmax as long as exp2(s - m_i) stays in
    // range; m_i = the maximum of the query's logits over a strided sample of the keys is at most the true row maximum
    // (so the row sum is >= ~1: no underflow) and, for every distribution short of a > 100-unit gap between the sample
    // and the true maximum, close enough that fp32 / bf16 exponents hold the rest.  The sweep then runs max-free
    // whatever the logits' magnitude; a row sum outside (1e-30, 1e30) flags the workgroup for the fallback pass.
    bool nomax = false;
    if (mshift) {
        float mv = mshift[(long)b * Npad + q0 + r];                    // bf16-representable by construction
        m = mv;
        if (h) qf[1][7] = (short)(__builtin_bit_cast(unsigned int, -mv) >> 16);   // d = 31 lives in lane half 1
        nomax = true;
    } else if (k_sqmax) {
        float q2 = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float qv = F16 ? (float)__builtin_bit_cast(_Float16, (unsigned short)qf[s][e]) : gd_bf2f((unsigned short)qf[s][e]);
                q2 = fmaf(qv, qv, q2);
            }
        q2 += __shfl_xor(q2, 32, 64);
        const float bound = sqrtf(q2 * k_sqmax[b]) * 1.0001f;
        // WORKGROUP-uniform choice: the two sweep instantiations each own a barrier site in the key-tile loop (the LDS-DMA
        // ring relies on them), so all eight waves must enter the same one
        nomax = __syncthreads_and(bound <= (F16 ? 13.f : 60.f)) != 0;
    }
    if (nomax) sweep(std::true_type{});
    else sweep(std::false_type{});

    if (ONES) l = __shfl(o[CT - 1][15], r + 32, 64);   // channel Cp-1 = accumulator row 31: register 15 of lane half 1
    if (mshift) {
        const bool bad = !(l > 1e-30f && l < 1e30f) && (q0 + r) < N;       // NaN / inf / 0: the shift missed this row
        if (__syncthreads_or(bad) && tid == 0) redo[blockIdx.y * gridDim.x + blockIdx.x] = 1;
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
        if (h == 0) lse[(long)b * N + qi] = (m + log2f(l)) * LN2;   // natural-log LSE of the UNSCALED energies
    }
}


// m[b][i] = max over NS strided sample keys j < N of q_i . k_j (log2 domain: q is pre-scaled), rounded UP to bf16.
// Thread = query; the sample keys sit in LDS (NS x 32 16-bit).  q slot 31 is zero in memory (the kernels only use it in
// registers), so k's ones column does not enter.  Padded queries (zero rows) get 0.
template <int NS, bool F16>
__global__ __launch_bounds__(256) void pam_row_shift_kernel(const unsigned short* __restrict__ qt,
                                                           const unsigned short* __restrict__ kt, int N, int Npad,
                                                           float* __restrict__ mshift) {
    __shared__ __attribute__((aligned(16))) unsigned short ks[NS * 32];
    const int b = blockIdx.y, tid = threadIdx.x;
    const long nb = (long)b * Npad;
    const int stride = N / NS > 0 ? N / NS : 1;
    for (int c = tid; c < NS * 4; c += 256) {
        const int j = c >> 2, part = c & 3;
        const long key = (long)j * stride < N ? (long)j * stride : (long)N - 1;
        *reinterpret_cast<u32x4_t*>(ks + j * 32 + part * 8) = *reinterpret_cast<const u32x4_t*>(kt + (nb + key) * 32 + part * 8);
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= Npad) return;
    float q[32];
    {
        const u32x4_t* qp = reinterpret_cast<const u32x4_t*>(qt + (nb + i) * 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4_t w = qp[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q[c * 8 + 2 * e] = pam::unpack_lo<F16>(w[e]);
                q[c * 8 + 2 * e + 1] = pam::unpack_hi<F16>(w[e]);
            }
        }
        q[31] = 0.f;
    }
    float best = -3.0e38f;
    for (int j = 0; j < NS; ++j) {
        const u32x4_t* kp = reinterpret_cast<const u32x4_t*>(ks + j * 32);      // same address in every lane: LDS broadcast
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4_t w = kp[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d = fmaf(q[c * 8 + 2 * e], pam::unpack_lo<F16>(w[e]), d);
                d = fmaf(q[c * 8 + 2 * e + 1], pam::unpack_hi<F16>(w[e]), d);
            }
        }
        best = fmaxf(best, d);
    }
    const unsigned int wb = __builtin_bit_cast(unsigned int, best);
    const float up = __builtin_bit_cast(float, best > 0.f ? (wb + 0xFFFFu) & 0xFFFF0000u : wb & 0xFFFF0000u);
    mshift[nb + i] = up;
}

// =====================================================================================================
// backward, part 1c: dK^T / dV^T, transpose-read variant.  Only Q [i][d] and dO [i][c] are staged per query
// tile; the "transposed" A operands of dV^T += dO^T P and dK^T += Q^T dS are taken from the SAME images with
// ds_read_b64_tr_b16 (4 query rows x 16 channel columns per 16-lane group), so the q^T / dO^T copies, half of
// the staging traffic and half of the tile LDS disappear.  V rows of the workgroup's keys live in LDS.
// NW = 4 (without dQ): 4 waves per workgroup and two independent workgroups per CU: their phases drift apart, so one's MFMA
// segments overlap the other's softmax / staging segments instead of all eight waves meeting at one barrier.
// =====================================================================================================
// DQ: also emit this key block's contribution to dQ.  dS is already in registers here; recomputing S and dP in a
// second query-parallel kernel (pam_bwd_dq_kernel) costs 14 MFMAs per 32x32 tile, turning the tile around through
// 2.5 KB of wave-private LDS and 2 more MFMAs does not: each wave transposes its dS tile (LDS transpose read),
// multiplies by its K rows, the NW waves' partial dQ^T tiles are summed after the barrier and stored as bf16
// [key block][query][32 d]; pam_dq_reduce_kernel sums the key blocks.  No atomics: deterministic.
template <int CT, bool DQ, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void pam_bwd_dkv3_kernel(
    const unsigned short* __restrict__ qt, const unsigned short* __restrict__ kt, const unsigned short* __restrict__ kn,
    const unsigned short* __restrict__ vt, const unsigned short* __restrict__ dot_, const float* __restrict__ lse,
    const float* __restrict__ delta, int N, int Npad, float* __restrict__ dkn, float* __restrict__ dv,
    unsigned short* __restrict__ dq_part) {
    constexpr int CP = CT * 32;
    constexpr int NT = NW * 64, KEYS = NW * 32;    // NW = 4: 128 keys, two workgroups per CU; NW = 8: 256 keys, one
    constexpr int DLD = CP + 8;
    constexpr int NCHUNK = 128 + 128 * CT;
    constexpr int NPRE = (NCHUNK + NT - 1) / NT;
    constexpr bool RAGGED = (NCHUNK % NT) != 0;
    __shared__ __attribute__((aligned(16))) unsigned short Vls[KEYS * DLD];
    __shared__ __attribute__((aligned(16))) unsigned short Qs[32 * B_QLD];
    constexpr int DOLD = CP + 32;                                // dO rows, chunk-swizzled (do_off)
    __shared__ __attribute__((aligned(16))) unsigned short dOs[32 * DOLD];
    __shared__ float Ls[32], Ds[32];
    constexpr int XLD = 36;                                      // 72-byte rows: 32 lanes x 8 bytes hit 32 distinct bank pairs (80-byte rows: 2-way, PMC)
    __shared__ __attribute__((aligned(16))) unsigned short Xs[DQ ? NW * 32 * XLD : 8];   // per wave: dS^T, then its dQ^T part

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * KEYS + wave * 32;
    const long nb = (long)b * Npad;

    bf16x8_t kfB[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) kfB[s] = *reinterpret_cast<const bf16x8_t*>(kt + (nb + j0 + r) * 32 + s * 16 + 8 * h);
    // K^T rows of this wave's keys as the A operand of dQ^T[d][i] += K^T[d][j] dS^T[j][i]: lane = d, 8 keys per k-step
    // (kn is stored perm16: one 16-byte read is an accumulator-row-ordered k-step)
    bf16x8_t knA[2];
    if (DQ) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            knA[s] = *reinterpret_cast<const bf16x8_t*>(kn + ((long)b * 32 + r) * Npad + j0 + s * 16 + 8 * h);
    }
    unsigned short* Xw = Xs + (DQ ? wave * 32 * XLD : 0);
    unsigned short* part = DQ ? dq_part + ((long)b * (Npad / KEYS) + blockIdx.x) * Npad * 32 : nullptr;
    {
        const unsigned short* vsrc = vt + (nb + (long)blockIdx.x * KEYS) * CP;
        for (int c = tid; c < KEYS * (CP / 8); c += NT) {
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
    const bool need_mask = (int)(blockIdx.x + 1) * KEYS > N;
    const int nqt = (N + 31) / 32;

    // staging plan: 32-bit element offset inside the tile's Q or dO source and LDS slot.  Chunks 0..127 are Q, i.e.
    // exactly waves 0-1 of round k = 0: the source kind of a (round, wave) is wave-uniform, so the 64-bit base stays
    // in SGPRs
    // (packed two to a register: at CT = 6 the kernel sits on the 256-VGPR line)
    unsigned int plan[NPRE];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        int c = tid + k * NT;
        if (RAGGED && k == NPRE - 1 && c >= NCHUNK) c = 128;
        unsigned int so, dof;
        if (c < 128) {
            so = (unsigned int)((c >> 2) * 32 + (c & 3) * 8);
            dof = (unsigned int)((c >> 2) * B_QLD + (c & 3) * 8);
        } else {
            const int c2 = c - 128;
            const int i = c2 / (4 * CT), ch = c2 - i * (4 * CT);
            so = (unsigned int)(i * CP + ch * 8);
            dof = (unsigned int)do_off(i, ch, DOLD);
        }
        plan[k] = so | (dof << 16);          // both < 2^16
    }
    const bool last_ok = !RAGGED || (tid + (NPRE - 1) * NT) < NCHUNK;
    const bool q_round0 = __builtin_amdgcn_readfirstlane(tid < 128 ? 1 : 0) != 0;

    u32x4_t pre[NPRE];
    float pre_s = 0.f;
    auto load_tile = [&](int qtile) {
        const unsigned short* qbase = qt + (nb + (long)qtile * 32) * 32;
        const unsigned short* dbase = dot_ + (nb + (long)qtile * 32) * CP;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            pre[k] = *reinterpret_cast<const u32x4_t*>((k == 0 && q_round0 ? qbase : dbase) + (plan[k] & 0xFFFFu));
        }
        if (tid < 64) {
            const int i = qtile * 32 + (tid & 31);
            pre_s = i < N ? -(tid < 32 ? lse[(long)b * N + i] * LOG2E : delta[(long)b * N + i]) : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
            if (k < NPRE - 1 || last_ok)
                *reinterpret_cast<u32x4_t*>((k == 0 && q_round0 ? Qs : dOs) + (plan[k] >> 16)) = pre[k];
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
            const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(dOs + do_off(r, 2 * s + h, DOLD));
            const bf16x8_t vb = *reinterpret_cast<const bf16x8_t*>(Vls + (wave * 32 + r) * DLD + s * 16 + 8 * h);
            dpacc = mfma_bf16(da, vb, dpacc);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = gd_exp2_fast(sacc[e]);   // P (q pre-scaled, -lse*log2e was the accumulator input)
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
        // the dS tile takes a round trip through wave-private LDS (transpose) for dQ; it is written before the dV^T
        // MFMAs and read back after them, so the trip is covered by matrix work
        if (DQ) {
            // dS tile [query rows][key lanes] -> X[key][query] (this lane's 16 queries are 4 runs of 4)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4_t w = __builtin_bit_cast(u32x4_t, dsf[s]);
                const u32x2_t lo = {w.x, w.y}, hi = {w.z, w.w};
                *reinterpret_cast<u32x2_t*>(Xw + r * XLD + 16 * s + 4 * h) = lo;
                *reinterpret_cast<u32x2_t*>(Xw + r * XLD + 16 * s + 8 + 4 * h) = hi;
            }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) dvacc[ct] = mfma_bf16(read_tr_frag_sw(dOs, DOLD, s, ct, lane), pf[s], dvacc[ct]);
        f32x16_t dqp;
        if (DQ) {
            // transpose read = the B operand (lane = query, k = key) of dQ^T[d][i] += K^T[d][j] dS^T[j][i]
#pragma unroll
            for (int e = 0; e < 16; ++e) dqp[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) dqp = mfma_bf16(knA[s], read_tr_frag(Xw, XLD, s, 0, lane), dqp);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) dkacc = mfma_bf16(read_tr_frag(Qs, B_QLD, s, 0, lane), dsf[s], dkacc);
        if (DQ) {
            // dQ^T part [d rows][query lanes] -> P[query][d] over the same wave-private region
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4_t w = __builtin_bit_cast(u32x4_t, pack_frag(dqp, s));
                const u32x2_t lo = {w.x, w.y}, hi = {w.z, w.w};
                *reinterpret_cast<u32x2_t*>(Xw + r * XLD + 16 * s + 4 * h) = lo;
                *reinterpret_cast<u32x2_t*>(Xw + r * XLD + 16 * s + 8 + 4 * h) = hi;
            }
        }
        __syncthreads();
        if (DQ) {   // sum the NW waves' parts: thread = (query, 4 d) ; 2 KiB contiguous per key block and query tile
            const int q = (tid & 255) >> 3, dg = (tid & 7) * 4;
            float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w4 = 0; w4 < NW; ++w4) {
                const u32x2_t v2 = *reinterpret_cast<const u32x2_t*>(Xs + w4 * 32 * XLD + q * XLD + dg);
                acc4[0] += gd_bf2f((unsigned short)(v2.x & 0xFFFFu));
                acc4[1] += gd_bf2f((unsigned short)(v2.x >> 16));
                acc4[2] += gd_bf2f((unsigned short)(v2.y & 0xFFFFu));
                acc4[3] += gd_bf2f((unsigned short)(v2.y >> 16));
            }
            const u32x2_t o = {gd_pack_bf2(acc4[0], acc4[1]), gd_pack_bf2(acc4[2], acc4[3])};
            if (tid < 256) *reinterpret_cast<u32x2_t*>(part + ((long)i0 + q) * 32 + dg) = o;
        }
        if (qtile + 1 < nqt) store_tile();
        if (DQ || qtile + 1 < nqt) __syncthreads();
    }

    const int j = j0 + r;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) dv[((long)b * CP + ct * 32 + acc_row(e, h)) * Npad + j] = dvacc[ct][e];
#pragma unroll
    for (int e = 0; e < 16; ++e) dkn[((long)b * 32 + acc_row(e, h)) * Npad + j] = dkacc[e] * LN2;   // Q^T was q * log2 e
}

// dQ^T[b][d][i] = sum over key blocks of the bf16 parts written by pam_bwd_dkv3_kernel<CT, true>
//   part : (B, KB, Npad, 32) bf16 ; dqn : (B, 32, Npad) fp32.  One workgroup = 64 queries of one image; a key block's
//   slab for them is 4 KiB contiguous (thread = query x 8 d, 16-byte loads).
__global__ __launch_bounds__(256) void pam_dq_reduce_kernel(const unsigned short* __restrict__ part, int KB, int Npad,
                                                           float* __restrict__ dqn) {
    __shared__ float tile[32][65];
    const int b = blockIdx.y, i0 = blockIdx.x * 64;
    const int q = threadIdx.x >> 2, d8 = (threadIdx.x & 3) * 8;
    const unsigned short* p = part + ((long)b * KB * Npad + i0 + q) * 32 + d8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll 4
    for (int kb = 0; kb < KB; ++kb) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(p + (long)kb * Npad * 32);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[2 * k] += gd_bf2f((unsigned short)(v[k] & 0xFFFFu));
            acc[2 * k + 1] += gd_bf2f((unsigned short)(v[k] >> 16));
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[d8 + k][q] = acc[k];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {
        const int d = idx >> 6, qq = idx & 63;
        dqn[((long)b * 32 + d) * Npad + i0 + qq] = tile[d][qq];
    }
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
                sacc[e] = nlse;
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
                dpacc[e] = gd_exp2_fast(sacc[e]) * dpacc[e];  // dS^T = P (dP - delta)
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

// max_j |k_j|^2 over the keys of an image, on the packed operand kt (B, Npad, 32) (slot d = 31 holds 1.0: excluded)
__global__ __launch_bounds__(256) void pam_key_sqnorm_max_kernel(const unsigned short* __restrict__ kt, int N, int Npad,
                                                                int f16, float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    float s2 = 0.f;
    if (j < N) {
        const u32x4_t* p = reinterpret_cast<const u32x4_t*>(kt + ((long)b * Npad + j) * 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4_t w = p[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = f16 ? unpack_lo<true>(w[e]) : unpack_lo<false>(w[e]);
                const float hi = f16 ? unpack_hi<true>(w[e]) : unpack_hi<false>(w[e]);
                s2 = fmaf(lo, lo, s2);
                if (c != 3 || e != 3) s2 = fmaf(hi, hi, s2);      // element 31 = the ones slot
            }
        }
    }
    s2 = gd_block_max(s2, red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(out + b), __builtin_bit_cast(int, s2));   // non-negative floats order as ints
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

static int pam_fwd_launch(const void* qt, const void* kt, const void* v, int B, int N, int Npad, int C, int Cp,
                          int v_ones, int f16, const float* gamma, const float* x, long x_bs, float* out,
                          long out_bs, float* o_attn, float* lse, const float* k_sqnorm_max, const float* mshift, int* redo,
                          void* stream);

extern "C" int gd_pam_flash_fwd(const void* qt, const void* kt, const void* v, int B, int N, int Npad, int C, int Cp,
                                int v_ones, int f16, const float* gamma, const float* x, long x_bs, float* out,
                                long out_bs, float* o_attn, float* lse, const float* k_sqnorm_max, void* stream) {
    return pam_fwd_launch(qt, kt, v, B, N, Npad, C, Cp, v_ones, f16, gamma, x, x_bs, out, out_bs, o_attn, lse, k_sqnorm_max,
                          nullptr, nullptr, stream);
}

// The same forward with the max-free sweep for logits of ANY magnitude (bf16 operands): a prepass takes every query's
// maximum over `nsample` (128 / 256 / 512) strided keys as its softmax shift (see pam_fwd_dma_kernel); workgroups whose row
// sums leave the fp32-safe range are redone by the running-maximum sweep (second launch, everyone else exits at once).
// ws: gd_pam_fwd_shift_ws_bytes(B, Npad) bytes.
extern "C" size_t gd_pam_fwd_shift_ws_bytes(int B, int Npad) {
    return (size_t)B * Npad * sizeof(float) + (size_t)B * (Npad / 256) * sizeof(int);
}
extern "C" int gd_pam_flash_fwd_shift(const void* qt, const void* kt, const void* v, int B, int N, int Npad, int C, int Cp,
                                      int v_ones, const float* gamma, const float* x, long x_bs, float* out, long out_bs,
                                      float* o_attn, float* lse, int nsample, void* ws, size_t ws_bytes, void* stream) {
    GD_CHECK_ARG(qt && kt && ws && B > 0 && B <= 65535 && N > 0 && Npad >= N && Npad % 256 == 0, "gd_pam_flash_fwd_shift: bad arguments");
    GD_CHECK_ARG(ws_bytes >= gd_pam_fwd_shift_ws_bytes(B, Npad), "gd_pam_flash_fwd_shift: workspace too small");
    GD_CHECK_ARG(nsample == 128 || nsample == 256 || nsample == 512, "gd_pam_flash_fwd_shift: nsample must be 128, 256 or 512");
    hipStream_t s = (hipStream_t)stream;
    float* mshift = reinterpret_cast<float*>(ws);
    int* redo = reinterpret_cast<int*>(mshift + (size_t)B * Npad);
    GD_CHECK_ARG(hipMemsetAsync(redo, 0, (size_t)B * (Npad / 256) * sizeof(int), s) == hipSuccess, "gd_pam_flash_fwd_shift: memset failed");
    const dim3 g(Npad / 256, B);
    const unsigned short *q16 = (const unsigned short*)qt, *k16 = (const unsigned short*)kt;
    if (nsample == 128) hipLaunchKernelGGL((pam_row_shift_kernel<128, false>), g, dim3(256), 0, s, q16, k16, N, Npad, mshift);
    else if (nsample == 256) hipLaunchKernelGGL((pam_row_shift_kernel<256, false>), g, dim3(256), 0, s, q16, k16, N, Npad, mshift);
    else hipLaunchKernelGGL((pam_row_shift_kernel<512, false>), g, dim3(256), 0, s, q16, k16, N, Npad, mshift);
    int rc = pam_fwd_launch(qt, kt, v, B, N, Npad, C, Cp, v_ones, 0, gamma, x, x_bs, out, out_bs, o_attn, lse, nullptr, mshift, redo, stream);
    if (rc) return rc;
    return pam_fwd_launch(qt, kt, v, B, N, Npad, C, Cp, v_ones, 0, gamma, x, x_bs, out, out_bs, o_attn, lse, nullptr, nullptr, redo, stream);
}

static int pam_fwd_launch(const void* qt, const void* kt, const void* v, int B, int N, int Npad, int C, int Cp,
                          int v_ones, int f16, const float* gamma, const float* x, long x_bs, float* out,
                          long out_bs, float* o_attn, float* lse, const float* k_sqnorm_max, const float* mshift, int* redo,
                          void* stream) {
    GD_CHECK_ARG(qt && kt && v && gamma && x && out && o_attn && lse, "gd_pam_flash_fwd: null pointer");
    GD_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && Npad >= N && Npad % 256 == 0, "gd_pam_flash_fwd: Npad must be a multiple of 256 >= N");
    GD_CHECK_ARG(C > 0 && Cp >= C && Cp % 32 == 0 && Cp <= 192, "gd_pam_flash_fwd: Cp must be a multiple of 32, C <= Cp <= 192");
    GD_CHECK_ARG(!v_ones || C < Cp, "gd_pam_flash_fwd: v_ones needs a spare padded channel (C < Cp)");
    // LDS-DMA ring, 8 waves (256 queries) per workgroup, 128-key tiles.  Rejected after measurement (B=4, N=65536,
    // C=184, this kernel 7.23 ms): 64-key tiles with 8 / 4 waves (+1 % / +9 %), register-staged 4-wave kernel
    // (+13 %), 4 waves x 64 queries with O in the accumulation registers (inline-asm MFMA; halves the LDS reads
    // per MFMA but leaves one wave per SIMD with nothing to overlap its softmax: +39 %).  Halving the V fragment
    // reads of THIS kernel (experiment, wrong numerics) changed nothing: it is not LDS-bandwidth bound.
    const dim3 grid(Npad / 256, B), block(512);
    hipStream_t s = (hipStream_t)stream;
#define PAM_FWD_ARGS (const unsigned short*)qt, (const unsigned short*)kt, (const unsigned short*)v, N, Npad, C, gamma, x, x_bs, out, out_bs, o_attn, lse, k_sqnorm_max, mshift, redo
    if (f16) {
        if (v_ones) {
            PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_dma_kernel<CT, 8, 128, true, true>), grid, block, 0, s, PAM_FWD_ARGS));
        } else {
            PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_dma_kernel<CT, 8, 128, false, true>), grid, block, 0, s, PAM_FWD_ARGS));
        }
    } else if (v_ones) {
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_dma_kernel<CT, 8, 128, true, false>), grid, block, 0, s, PAM_FWD_ARGS));
    } else {
        PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_fwd_dma_kernel<CT, 8, 128, false, false>), grid, block, 0, s, PAM_FWD_ARGS));
    }
#undef PAM_FWD_ARGS
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" int gd_pam_key_sqnorm_max(const void* kt, int B, int N, int Npad, int f16, float* out, void* stream) {
    GD_CHECK_ARG(kt && out && B > 0 && B <= 65535 && N > 0 && Npad >= N, "gd_pam_key_sqnorm_max: bad arguments");
    GD_CHECK_ARG(hipMemsetAsync(out, 0, (size_t)B * sizeof(float), (hipStream_t)stream) == hipSuccess, "gd_pam_key_sqnorm_max: memset failed");
    hipLaunchKernelGGL(pam_key_sqnorm_max_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)kt, N, Npad, f16, out);
    GD_LAUNCH_CHECK();
    return 0;
}

extern "C" void gd_pam_dq_reduce_launch(const void* part, int KB, int Npad, int nb, float* dqn, void* stream) {
    hipLaunchKernelGGL(pam_dq_reduce_kernel, dim3(Npad / 64, nb), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)part, KB, Npad, dqn);
}

// pam_bwd64.hip
extern "C" size_t gd_pam_bwd64_scratch_bytes(int Npad, int deterministic);
extern "C" int gd_pam_bwd64_slice(const void* qt, const void* kt, const void* kn, const void* vt, const void* dot_,
                                  const float* lse, const float* delta, int nb, int N, int Npad, int Cp, int f16,
                                  int vreg, int deterministic, float* dqn, float* dkn, float* dv, long out_bs,
                                  void* scratch, void* stream);

// Backward forms (gandanet.h GD_PAM_BWD_*):
//   0 K64_ATOMIC : 4 waves x 64 keys, one wave per SIMD, fp32 atomics for dQ            (default, fastest)
//   1 K64_PARTS  : the same kernel, dQ as bf16 parts per 256-key block + streaming sum   (bitwise reproducible)
//   2 K32_PARTS  : round-1 kernel, 8 waves x 32 keys, bf16 parts                         (bitwise reproducible)
//   3 TWO_KERNEL : dK/dV kernel + query-parallel dQ kernel recomputing S and dP          (no scratch)
extern "C" size_t gd_pam_bwd_scratch_bytes(int Npad, int form) {
    if (form == 0 || form == 1) return gd_pam_bwd64_scratch_bytes(Npad, form == 1);
    if (form == 2) return (size_t)(Npad / 256) * (size_t)Npad * 32 * sizeof(unsigned short);
    return 0;
}

// V fragments of both key tiles of a wave in registers (default; 15.9 ms at B=4, N=65536, C=184) or, GD_PAM_K64_VREG=1,
// one in registers and one in LDS (16.6 ms)
static int pam_k64_vreg() {
    static const int v = getenv("GD_PAM_K64_VREG") ? atoi(getenv("GD_PAM_K64_VREG")) : 2;
    return v == 1 ? 1 : 2;
}

extern "C" int gd_pam_flash_bwd(const void* qt, const void* kt, const void* kn, const void* vt, const void* dot_,
                                const float* lse, const float* delta, int B, int N, int Npad, int Cp, int f16, int form,
                                float* dqn, float* dkn, float* dv, long out_bs, void* scratch, size_t scratch_bytes,
                                void* stream) {
    GD_CHECK_ARG(qt && kt && kn && vt && dot_ && lse && delta && dqn && dkn && dv, "gd_pam_flash_bwd: null pointer");
    GD_CHECK_ARG(out_bs == 0 || (form == 0 && out_bs >= (long)Cp * Npad), "gd_pam_flash_bwd: a shared output batch stride needs form 0");
    GD_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && Npad >= N && Npad % 256 == 0, "gd_pam_flash_bwd: Npad must be a multiple of 256 >= N");
    GD_CHECK_ARG(Cp > 0 && Cp % 32 == 0 && Cp <= 192, "gd_pam_flash_bwd: Cp must be a multiple of 32 <= 192");
    GD_CHECK_ARG(form >= 0 && form <= 3, "gd_pam_flash_bwd: form must be 0..3");
    GD_CHECK_ARG(!f16 || form <= 1, "gd_pam_flash_bwd: fp16 operands need form 0 or 1");
    hipStream_t s = (hipStream_t)stream;
    const unsigned short *q = (const unsigned short*)qt, *k = (const unsigned short*)kt, *kT = (const unsigned short*)kn;
    const unsigned short *v = (const unsigned short*)vt, *dO = (const unsigned short*)dot_;
    const size_t per_image = gd_pam_bwd_scratch_bytes(Npad, form);
    if (form <= 2) {
        GD_CHECK_ARG(scratch && scratch_bytes >= per_image, "gd_pam_flash_bwd: scratch smaller than gd_pam_bwd_scratch_bytes (one image)");
        // the batch is walked in slices that fit the caller's scratch buffer
        const int slice = (int)(scratch_bytes / per_image < (size_t)B ? scratch_bytes / per_image : (size_t)B);
        for (int b0 = 0; b0 < B; b0 += slice) {
            const int nb = B - b0 < slice ? B - b0 : slice;
            const long o32 = (long)b0 * Npad * 32, oc = (long)b0 * Npad * Cp, on = (long)b0 * N;
            if (form <= 1) {
                const long oq = out_bs ? (long)b0 * out_bs : o32, ov = out_bs ? (long)b0 * out_bs : oc;
                const int rcode = gd_pam_bwd64_slice(q + o32, k + o32, kT + o32, v + oc, dO + oc, lse + on, delta + on, nb, N,
                                                     Npad, Cp, f16, pam_k64_vreg(), form == 1, dqn + oq, dkn + oq, dv + ov,
                                                     out_bs, scratch, stream);
                if (rcode) return rcode;
            } else {
                PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_bwd_dkv3_kernel<CT, true, 8>), dim3(Npad / 256, nb), dim3(512), 0, s,
                                                             q + o32, k + o32, kT + o32, v + oc, dO + oc, lse + on, delta + on,
                                                             N, Npad, dkn + o32, dv + oc, (unsigned short*)scratch));
                gd_pam_dq_reduce_launch(scratch, Npad / 256, Npad, nb, dqn + o32, stream);
            }
        }
        GD_LAUNCH_CHECK();
        return 0;
    }
    // two-kernel form: dK / dV (transpose-read kernel, 4 waves = 128 keys, two workgroups per CU), then dQ with a
    // query-parallel kernel that recomputes S and dP (8 waves, 128-key LDS-DMA tiles; kn perm16 along the keys)
    PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_bwd_dkv3_kernel<CT, false>), dim3(Npad / 128, B), dim3(256), 0, s, q, k,
                                                 kT, v, dO, lse, delta, N, Npad, dkn, dv, (unsigned short*)nullptr));
    PAM_DISPATCH_CT(Cp / 32, hipLaunchKernelGGL((pam_bwd_dq_kernel<CT, 8, 128>), dim3(Npad / 256, B), dim3(512), 0, s, q, k,
                                                 kT, v, dO, lse, delta, N, Npad, dqn));
    GD_LAUNCH_CHECK();
    return 0;
}
