// PAM backward, key-parallel, ONE wave per SIMD (generator.py:115-122 under autograd).
//
// A workgroup = 4 waves = 256 keys of one image; each wave owns 64 keys (two 32-key tiles) and keeps their dV^T
// (64 x Cp) and dK^T (64 x 32) in accumulators for the whole sweep over the queries -- the 512-register shape (one
// wave per SIMD, accumulators in the AGPR half).  Per 32-query tile a wave issues 60 MFMAs:
//     S = Q K^T (2+2), dP = dO V^T (2 x 2CT), dV^T += dO^T P (2 x 2CT), dK^T += Q^T dS (2+2), dQ^T part (4)
// and every A fragment read from LDS (Q rows, dO rows, dO^T / Q^T transpose reads) feeds TWO MFMAs, one per key tile:
// half the LDS bytes per MFMA of the 32-keys-per-wave kernel (pam_bwd_dkv3_kernel), and one barrier per 60 MFMAs.
//   * key on the lane: S and dP come out [query rows][key lanes], so P and dS are directly the B operands of the
//     dV^T / dK^T products (accumulator-as-operand); -lse*log2e and -delta enter as the accumulator input.
//   * Q [i][d], dO [i][c] and the row constants of a query tile arrive by LDS-DMA (global_load_lds_dwordx4) into a
//     3-slot ring, two tiles ahead, behind counted vmcnt waits and a raw s_barrier (one per tile).
//   * V rows of the wave's keys: VREG of its two key tiles are held in registers as B fragments, the rest in LDS.
//   * dQ: the wave's dS tiles are turned around through wave-private LDS (transpose read), multiplied by its K^T
//     rows (4 MFMAs sum over its 64 keys), the four waves' fp32 parts are exchanged through LDS and
//       ATOMIC : added with fp32 atomics (256 contiguous bytes per wave-instruction) into dq_acc (B, Npad, 32);
//                key blocks start their sweep at different query tiles so that adds spread over many rows;
//       !ATOMIC: stored as one bf16 part per key block (deterministic; pam_dq_reduce_kernel sums them).
// No masks: padded queries carry -1e30 as their -lse (P = 0), padded keys only touch padded outputs (the packs are
// zero filled).
#include <stdlib.h>
#include "pam_common.h"
#include "../../include/gandanet.h"

namespace {

using namespace pam;

// rc (B, Npad/32, 64): per 32-query tile [32 x -lse*log2e][32 x -delta]; padded queries: -1e30 / 0
__global__ __launch_bounds__(256) void pam_rowconst_kernel(const float* __restrict__ lse, const float* __restrict__ delta,
                                                          int N, int Npad, float* __restrict__ rc) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;
    float a = -1e30f, d = 0.f;
    if (i < N) {
        a = -lse[(long)b * N + i] * LOG2E;
        d = -delta[(long)b * N + i];
    }
    float* t = rc + ((long)b * (Npad / 32) + (i >> 5)) * 64 + (i & 31);
    t[0] = a;
    t[32] = d;
}

// dq_acc (B, Npad, 32) fp32 [query][d]  ->  dqn (B, 32, Npad) [d][query]
__global__ __launch_bounds__(256) void pam_dq_transpose_kernel(const float* __restrict__ acc, int Npad,
                                                              float* __restrict__ dqn, long dqn_bs) {
    __shared__ float tile[64][33];
    const int b = blockIdx.y, i0 = blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
        const int q = idx >> 5, d = idx & 31;
        tile[q][d] = acc[((long)b * Npad + i0 + q) * 32 + d];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {
        const int d = idx >> 6, q = idx & 63;
        dqn[(long)b * dqn_bs + (long)d * Npad + i0 + q] = tile[q][d];
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- transpose reads as inline asm -------------------------------------------------------------------------------
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr16 BUILTIN whenever an LDS-DMA is in flight
// (its LDS-DMA alias tracking has no scope information for that intrinsic), which would drain the DMA ring and the dQ
// atomics every tile.  The asm forms are invisible to that pass; their results are handed to the compiler only through
// tr_wait(), which names them as read-write operands (guide 5.7 form ii): no consumer can be scheduled above the wait.
// LDS returns in order, so lgkmcnt(N) with N = the transpose reads issued AFTER the batch being waited for is exact
// when nothing else was issued in between and merely stricter when the compiler slipped LDS operations of its own in.
struct TrFrag2 {          // the two A fragments (k-steps s = 0, 1) of one 32-row operand tile
    s16x4_t lo0, hi0, lo1, hi1;
};
__device__ __forceinline__ unsigned int lds_addr(const void* p) {
    return (unsigned int)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}
template <int OFF>
__device__ __forceinline__ s16x4_t tr_issue(unsigned int addr) {
    s16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// lo / hi: byte addresses of the lane's first row group and of the one 8 rows below; STEP: bytes between k-steps
template <int OFF, int STEP>
__device__ __forceinline__ TrFrag2 tr_issue2(unsigned int lo, unsigned int hi) {
    TrFrag2 f;
    f.lo0 = tr_issue<OFF>(lo);
    f.hi0 = tr_issue<OFF>(hi);
    f.lo1 = tr_issue<OFF + STEP>(lo);
    f.hi1 = tr_issue<OFF + STEP>(hi);
    return f;
}
template <int N>
__device__ __forceinline__ void tr_wait(TrFrag2& f) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.lo0), "+v"(f.hi0), "+v"(f.lo1), "+v"(f.hi1) : "n"(N) : "memory");
}
__device__ __forceinline__ bf16x8_t tr_frag(const s16x4_t& lo, const s16x4_t& hi) {
    const bf16x8_t f = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return f;
}
// plain 16-byte LDS reads issued / awaited by hand (same contract as tr_issue / tr_wait): the compiler sinks its own
// ds_read next to the first use when registers are tight, which exposes the LDS latency in front of every MFMA pair
template <int OFF>
__device__ __forceinline__ bf16x8_t lds_issue128(unsigned int addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4_t lds_issue128f(unsigned int addr) {
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ float lds_issue32f(unsigned int addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait128(bf16x8_t& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
}
template <int I> struct IC { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- VGPR-form MFMAs as inline asm (schedule variant bit 3) ----------------------------------------------------------
// With one wave per SIMD the function's register budget is 512, so hipcc selects the AGPR form for EVERY MFMA: the S and
// dP tiles, which only feed VALU code, then cost a v_accvgpr_write per accumulator-input register and a v_accvgpr_read
// per result register (128 of the ~250 VALU instructions of a tile).  The asm forms keep those four tiles in VGPRs and
// take the row constants straight from their registers as srcC.  The hazard recognizer does not look inside inline asm:
// the consumers of a result are held back by mfma_pad() (a data dependency + the software wait states the ISA asks for
// between an 8-pass XDL write and a VALU read of the same VGPR: 11).
template <bool F16>
__device__ __forceinline__ void mfma_v_first(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
    if constexpr (F16) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
template <bool F16>
__device__ __forceinline__ void mfma_v_zero(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {
    if constexpr (F16) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
template <bool F16>
__device__ __forceinline__ void mfma_v_acc(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {
    if constexpr (F16) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
// orders the consumers of a, b behind this point (and, with NOPS, behind the wait states of the last MFMA into them)
template <bool NOPS>
__device__ __forceinline__ void mfma_pad(f32x16_t& a, f32x16_t& b) {
    if constexpr (NOPS) asm("s_nop 10" : "+v"(a), "+v"(b));
    else asm("" : "+v"(a), "+v"(b));
}
// The first two k-steps of two accumulators that start from the same srcC tile c, as ONE statement: the srcC of an
// in-flight 32x32 MFMA must not be overwritten for 13 wait states (WAR), and the compiler -- which does not know that
// the asm reads c late -- would recycle c's registers right behind a single-MFMA statement (seen: an LDS load and a
// v_accvgpr_read landing in the row constants under the MFMAs that were still reading them).  Two further MFMAs
// (>= 16 quad-cycles) separate the last read of c from the end of this statement.
template <bool F16>
__device__ __forceinline__ void mfma_v_first2x2(f32x16_t& d0, f32x16_t& d1, const f32x16_t& c, const bf16x8_t& a0,
                                                const bf16x8_t& b00, const bf16x8_t& b10, const bf16x8_t& a1,
                                                const bf16x8_t& b01, const bf16x8_t& b11) {
    if constexpr (F16)
        asm("v_mfma_f32_32x32x16_f16 %0, %3, %4, %2\n\tv_mfma_f32_32x32x16_f16 %1, %3, %5, %2\n\t"
            "v_mfma_f32_32x32x16_f16 %0, %6, %7, %0\n\tv_mfma_f32_32x32x16_f16 %1, %6, %8, %1"
            : "=&v"(d0), "=&v"(d1)
            : "v"(c), "v"(a0), "v"(b00), "v"(b10), "v"(a1), "v"(b01), "v"(b11));
    else
        asm("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %2\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %5, %2\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %6, %8, %1"
            : "=&v"(d0), "=&v"(d1)
            : "v"(c), "v"(a0), "v"(b00), "v"(b10), "v"(a1), "v"(b01), "v"(b11));
}
// AGPR-form accumulate as asm: the dV^T steps of the hand-placed second half (schedule bit 5).  Source order is the
// schedule; mfma_tie_a() pins VALU chunks between two MFMA pairs (no instruction: a dependency through the pair's
// accumulators and the chunk's registers)
template <bool F16>
__device__ __forceinline__ void mfma_a_acc(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {
    // volatile: hipcc otherwise sinks the last steps' MFMAs into the dK^T / dQ^T steps behind them
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_tie_a(f32x16_t& acc0, f32x16_t& acc1, f32x16_t& x) {
    asm("" : "+a"(acc0), "+a"(acc1), "+v"(x));
}
template <bool NOPS>
__device__ __forceinline__ void mfma_pad4(f32x16_t& a, f32x16_t& b, f32x16_t& c, f32x16_t& d) {
    if constexpr (NOPS) asm("s_nop 10" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

template <int CT, bool F16, int VREG, bool ATOMIC, int ORDER = 0>
__global__ __launch_bounds__(256, 1) void pam_bwd_k64_kernel(
    const unsigned short* __restrict__ qt, const unsigned short* __restrict__ kt, const unsigned short* __restrict__ kn,
    const unsigned short* __restrict__ vt, const unsigned short* __restrict__ dot_, const float* __restrict__ rc,
    int Npad, float* __restrict__ dkn, float* __restrict__ dv, void* __restrict__ dq_out, long dk_bs, long dv_bs,
    unsigned int* __restrict__ dbg) {
    constexpr int CP = CT * 32;
    constexpr int DOLD = CP + 32;                  // dO rows, chunk-swizzled (do_off)
    constexpr int DROWCH = DOLD / 8;               // 16-byte chunks per dO row
    constexpr int QCH = 192;                       // Q region: 32 rows x 5 chunks (80-byte rows) = 160, padded to 3 pieces
    constexpr int DCH = 32 * DROWCH;               // = 128 (CT + 1): a whole number of 64-chunk pieces
    constexpr int SLOT_CH = QCH + DCH + 64;        // + one full DMA piece of row constants (16 chunks used: this tile's)
    constexpr int SLOT = SLOT_CH * 8;              // ring slot, in 16-bit elements
    constexpr int NPIECE = 3 + DCH / 64 + 1;
    constexpr int PPW = (NPIECE + 3) / 4;          // DMA wave-instructions per wave and tile (every wave issues PPW)
    constexpr int NSLOT = 3;
    constexpr int XLD = 36;                        // dS^T rows: 72 bytes (conflict-free 8-byte writes / transpose reads)
    constexpr int XW = 2 * 32 * XLD;               // per wave: the two key tiles' dS^T
    constexpr int QXLD = 36;                       // dQ exchange rows, floats
    constexpr int DLD = CP + 8;                    // V rows in LDS
    // bit 8 (DQ16): dQ without the cross-wave exchange.  The dS^T images of all four waves are double-buffered; after the
    // next iteration's barrier every wave forms ONE 16 x 16 sub-tile of the tile's dQ (queries 16 (wave & 1).., d columns
    // 16 (wave >> 1)..) over ALL 256 keys of the workgroup with eight v_mfma_f32_16x16x32 (A = dS by transpose reads of the
    // four images, B = the K rows of the 256 keys, held in registers) and adds it straight to dQ -- no fp32 exchange
    // buffer, no 16 single-dword reads and 12 adds per tile, no LDS write in front of the barrier.
    constexpr bool DQ16 = (ORDER & 256) != 0;
    // bit 9 (DV16): the dV^T and dK^T products (28 of the 60 MFMA-equivalents of a tile) on v_mfma_f32_16x16x32: the P / dS
    // fragments of a key tile are regrouped with v_permlane16_swap (8 + 8 per tile) into the B operands of its two 16-key
    // halves, the dO^T / Q^T operands come from the same transpose reads with other row addresses, the accumulators
    // become 16 x 16 tiles (same register count).  The kernel is power-managed: the 16-wide shape costs the
    // same cycles and less energy per FLOP.
    constexpr bool DV16 = (ORDER & 512) != 0;
    constexpr bool DV16_V = DV16 && !(ORDER & 2048), DV16_K = DV16 && !(ORDER & 1024);   // A/B: one product only
    constexpr int OFF_X = NSLOT * SLOT;
    constexpr int OFF_XQ = OFF_X + (DQ16 ? 2 : 1) * 4 * XW;
    constexpr int XQ_BUF = 4 * 32 * QXLD;          // floats per exchange buffer
    constexpr int OFF_V = OFF_XQ + (DQ16 ? 0 : 2 * XQ_BUF * 2);
    constexpr int VLDS = 2 - VREG;                 // key tiles per wave whose V rows live in LDS
    constexpr int V_ELEMS = 4 * VLDS * 32 * DLD;
    constexpr int KLD = 264;                       // DQ16: K^T image rows [d][256 keys + 8 pad] (528 B: conflict-free 16-byte column reads)
    constexpr int K_ELEMS = ((ORDER & 256) != 0) ? 32 * KLD : 0;
    constexpr int AOPS = (ATOMIC || (ORDER & 256) != 0) ? 4 : 1;           // VMEM operations of one dQ hand-over per wave
    // schedule variants (A/B switches, see the loop): bit 0 DQ_FIRST, bit 1 HANDOVER_MID, bit 2 DMA_LATE
    constexpr bool DQ_FIRST = (ORDER & 1) != 0, HANDOVER_MID = (ORDER & 2) != 0, DMA_LATE = (ORDER & 4) != 0;
    constexpr bool VFORM = (ORDER & 8) != 0;        // S / dP / dQ-part tiles through VGPR-form asm MFMAs
    // bit 4: dS = P (dP - delta) as 32 single v_mul_f32 (asm) instead of the 16 v_pk_mul_f32 hipcc's SLP vectoriser makes
    // of adjacent scalar multiplies -- packed f32 VALU issued while MFMAs are in flight is an anti-lever
    // (MI355X_MICROARCH.md, "price of one filler beside MFMAs": one v_pk_fma_f32 = +22 cycles over two v_fma_f32)
    constexpr bool NOPK = (ORDER & 16) != 0;
    // bit 5: hand-placed second half -- the dS arithmetic (32 multiplies, 16 packs, the dS^T writes: a 50-instruction
    // burst with the matrix pipe idle in the compiler's order) rides in the gaps of the first dV^T MFMA pairs, which only
    // need P; the first transpose reads are requested before it
    constexpr bool PH2 = (ORDER & 32) != 0;
    static_assert(!PH2 || VFORM, "the hand-placed second half continues the hand-placed dP phase");
    static_assert(!DQ16 || (VFORM && !PH2 && CT >= 4), "DQ16 rides in the hand-placed dP phase of the production schedule (>= 8 steps)");
    static_assert(!DV16 || (VFORM && !PH2 && !DQ_FIRST), "DV16 replaces the compiler-scheduled dV^T / dK^T steps of the production schedule");
    // bit 6 (diagnostic build only, tools/pam_stamps.py): s_memtime stamps at the segment seams of the tile loop; the
    // values are requested without a wait (SMEM returns through lgkmcnt: the loop-top lgkmcnt(0) covers them) and summed
    // per wave into dbg[(image, key block, wave)][8] = {wait+barrier, head, S+dP, dS, dV^T, dK^T+dQ^T, -, tiles}
    constexpr bool STAMP = (ORDER & 64) != 0;
    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0}, stw = 0;     // stw: between the counted waits and the barrier
    unsigned int sacc_t[6] = {0, 0, 0, 0, 0, 0}, sacc_w = 0;
    auto stamp = [&](auto ic) {
        if constexpr (STAMP) {
            unsigned long long& t = st[decltype(ic)::value];      // (a variable named only in an asm operand is not captured)
            asm volatile("s_memtime %0" : "=s"(t));
        }
    };
    static_assert(!HANDOVER_MID || DQ_FIRST, "the mid-iteration hand-over follows the early dQ steps");
    static_assert(DCH % 64 == 0 && (OFF_X % 8) == 0 && (OFF_XQ % 8) == 0 && (OFF_V % 8) == 0, "LDS carve");
    constexpr int OFF_K = OFF_V + (V_ELEMS ? V_ELEMS : 8);
    static_assert((OFF_K + K_ELEMS) * 2 <= 163840, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned short lds[OFF_K + (K_ELEMS ? K_ELEMS : 8)];   // the ONLY LDS object

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, kb = blockIdx.x;
    const int j0 = kb * 256 + wave * 64;
    const long nb = (long)b * Npad;
    const int nqt = Npad / 32;

    // ---- this wave's key-side operands (registers for the whole sweep) ----
    bf16x8_t kfB[2][2], knA[2][2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kfB[k2][s] = *reinterpret_cast<const bf16x8_t*>(kt + (nb + j0 + 32 * k2 + r) * 32 + s * 16 + 8 * h);
            // K^T rows as the A operand of dQ^T[d][i] += K^T[d][j] dS^T[j][i] (kn is perm16: one 16-byte read is an
            // accumulator-row-ordered k-step)
            knA[k2][s] = *reinterpret_cast<const bf16x8_t*>(kn + ((long)b * 32 + r) * Npad + j0 + 32 * k2 + s * 16 + 8 * h);
        }
    // DQ16: the K^T rows [d][key] of ALL 256 keys of the workgroup as an LDS image (kn is perm16 along the keys, which the
    // A-side row addresses below follow): the B operand (k = key, n = d) of the dQ sub-tiles is read from it per tile
    if constexpr (DQ16) {
        unsigned short* Kimg = lds + OFF_K;
        for (int c = tid; c < 32 * 32; c += 256) {
            const int row = c >> 5, ch = c & 31;
            *reinterpret_cast<u32x4_t*>(Kimg + row * KLD + ch * 8) =
                *reinterpret_cast<const u32x4_t*>(kn + ((long)b * 32 + row) * Npad + (long)kb * 256 + ch * 8);
        }
    }
    bf16x8_t vB[VREG > 0 ? VREG : 1][2 * CT];
#pragma unroll
    for (int k2 = 0; k2 < VREG; ++k2)
#pragma unroll
        for (int s = 0; s < 2 * CT; ++s)
            vB[k2][s] = *reinterpret_cast<const bf16x8_t*>(vt + (nb + j0 + 32 * k2 + r) * CP + s * 16 + 8 * h);
    unsigned short* Vw = lds + OFF_V + wave * (VLDS * 32 * DLD);   // wave-private V rows (key tiles VREG..1)
    if constexpr (VLDS > 0) {
        const unsigned short* vsrc = vt + (nb + j0 + 32 * VREG) * CP;
        for (int c = lane; c < VLDS * 32 * (CP / 8); c += 64) {
            const int row = c / (CP / 8), ch = c - row * (CP / 8);
            *reinterpret_cast<u32x4_t*>(Vw + row * DLD + ch * 8) = *reinterpret_cast<const u32x4_t*>(vsrc + (long)row * CP + ch * 8);
        }
    }
    // dQ exchange buffers start at zero: iteration 0 hands over an all-zero "previous tile"
    if constexpr (DQ16) {      // the dS^T images start at zero: iteration 0 forms the dQ of an all-zero "previous tile"
        unsigned int* xz = reinterpret_cast<unsigned int*>(lds + OFF_X);
        for (int c = tid; c < 2 * 4 * XW / 2; c += 256) xz[c] = 0u;
    } else {
        float* xq = reinterpret_cast<float*>(lds + OFF_XQ);
        for (int c = tid; c < 2 * XQ_BUF; c += 256) xq[c] = 0.f;
    }

    // ---- DMA plan: byte offset of this lane's chunk inside the tile's Q / dO / row-constant source ----
    unsigned int voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int piece = wave + 4 * i;
        if (piece >= NPIECE) piece = NPIECE - 2;           // filler: re-fetch a dO piece (same bytes, same place)
        const int c = piece * 64 + lane;
        if (piece < 3) {
            const int row = c / 5, part = c - row * 5;
            voff[i] = c < 160 ? (unsigned int)(row * 32 + (part < 4 ? part : 3) * 8) * 2u : 0u;
        } else if (piece < NPIECE - 1) {
            const int c2 = c - QCH;
            const int row = c2 / DROWCH, sl = c2 - row * DROWCH;
            const int chunk = sl < CP / 8 ? (sl ^ ((row >> 2) & 3)) : 0;
            voff[i] = (unsigned int)(row * CP + chunk * 8) * 2u;
        } else {
            voff[i] = (unsigned int)lane * 16u;             // 1 KiB = this tile's constants + the next three tiles' (unused)
        }
    }
    // wave-uniform source pointer of each of this wave's pieces for the NEXT tile to prefetch, and its per-tile
    // advance: they live in SGPRs and move by one add per piece and tile (no per-tile index arithmetic)
    const char* pbase[PPW];
    long padv[PPW];
    int pdst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int piece = wave + 4 * i;
        if (piece >= NPIECE) piece = NPIECE - 2;
        pdst[i] = piece * 512;
        if (piece < 3) {
            pbase[i] = reinterpret_cast<const char*>(qt + nb * 32);
            padv[i] = 32 * 32 * 2;
        } else if (piece < NPIECE - 1) {
            pbase[i] = reinterpret_cast<const char*>(dot_ + nb * CP);
            padv[i] = 32 * CP * 2;
        } else {
            pbase[i] = reinterpret_cast<const char*>(rc + (long)b * nqt * 64);
            padv[i] = 64 * 4;
        }
    }
    auto dma_seek = [&](int t) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) pbase[i] += (long)t * padv[i];
    };
    // issue the DMA of the tile the pointers stand on into ring slot `slot`, then step to tile `t + 1` (wrapping)
    long pwrap[PPW];                      // the step from the last tile back to tile 0 (select + add: no multiply per tile)
#pragma unroll
    for (int i = 0; i < PPW; ++i) pwrap[i] = -(long)(nqt - 1) * padv[i];
    auto dma_next = [&](int t, int slot) {
        const bool wrap = t + 1 == nqt;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            unsigned short* dst = lds + slot * SLOT + pdst[i];           // wave-uniform piece base
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pbase[i] + voff[i]),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            pbase[i] += wrap ? pwrap[i] : padv[i];
        }
    };

    // DV16: 16 x 16 accumulator tiles [channel tile][16-channel half][key tile][16-key half] / [16-d half][key tile][half]
    f32x4_acc_t dv16[DV16 ? CT : 1][2][2][2], dk16[2][2][2];
#pragma unroll
    for (int a_ = 0; a_ < (DV16 ? CT : 1); ++a_)
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_)
#pragma unroll
            for (int e = 0; e < 4; ++e) dv16[a_][i_ >> 2][(i_ >> 1) & 1][i_ & 1][e] = 0.f;
#pragma unroll
    for (int i_ = 0; i_ < 8; ++i_)
#pragma unroll
        for (int e = 0; e < 4; ++e) dk16[i_ >> 2][(i_ >> 1) & 1][i_ & 1][e] = 0.f;
    f32x16_t dvacc[2][CT], dkacc[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) dkacc[k2][e] = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) dvacc[k2][ct][e] = 0.f;
    }

    // key blocks that share an XCD (equal kb % 8 under round-robin placement) start close together (L2 reuse of the
    // streamed Q / dO tiles), 4 tiles apart (atomic adds of neighbours land in different rows); the 8 groups start an
    // eighth of the sweep apart
    int tcur = ATOMIC ? (int)(((long)(kb & 7) * (nqt / 8) + (long)(kb >> 3) * 4) % nqt) : 0;
    auto next_tile = [&](int t) { return t + 1 == nqt ? 0 : t + 1; };
    int tprev = tcur == 0 ? nqt - 1 : tcur - 1;

    unsigned short* Xw = lds + OFF_X + wave * XW;
    // per-lane LDS offsets (16-bit elements), fixed for the whole sweep
    const int li = lane & 15, g4 = (lane >> 4) & 1;
    const int swz_r = (r >> 2) & 3;
    const int off_qrow = r * B_QLD + 8 * h;                                   // Q row fragment: + 16 s
    const int off_do_e = r * DOLD + ((h ^ swz_r) << 3);                       // dO row fragment, even k-step s: + 32 (s >> 1)
    const int off_do_o = r * DOLD + (((2 + h) ^ swz_r) << 3);                 //                  odd k-step
    const int c2 = 2 * g4 + ((li & 3) >> 1);
    const int off_trd_lo = (4 * h + (li >> 2)) * DOLD + ((c2 ^ h) << 3) + 4 * (li & 1);            // dO^T: + 16 s DOLD + 32 ct
    const int off_trd_hi = (4 * h + (li >> 2) + 8) * DOLD + ((c2 ^ (h + 2)) << 3) + 4 * (li & 1);  //   rows + 8: swizzle h + 2
    const int off_trq = (4 * h + (li >> 2)) * B_QLD + 16 * g4 + 4 * (li & 3);                      // Q^T: + 16 s B_QLD (+ 8 B_QLD)
    // DV16: the A operand (16 rows x 32 queries) of a 16x16x32 MFMA by two transpose reads whose row addresses follow the
    // slot order of the regrouped P / dS fragments: lane group g = lane >> 4 takes the queries 16 (g & 1) + 4 (g >> 1) + {0..3}
    // (first read) and + 8 (second read); the chunk swizzle of the dO image is (row >> 2) & 3 = g >> 1 resp. (g >> 1) + 2
    const int row16 = 16 * ((lane >> 4) & 1) + 4 * (lane >> 5) + (li >> 2), p16 = li & 3, hh16 = lane >> 5;
    const int off16_d1[2] = {row16 * DOLD + (((0 + (p16 >> 1)) ^ hh16) << 3) + 4 * (p16 & 1),
                             row16 * DOLD + (((2 + (p16 >> 1)) ^ hh16) << 3) + 4 * (p16 & 1)};
    const int off16_d2[2] = {(row16 + 8) * DOLD + (((0 + (p16 >> 1)) ^ (hh16 + 2)) << 3) + 4 * (p16 & 1),
                             (row16 + 8) * DOLD + (((2 + (p16 >> 1)) ^ (hh16 + 2)) << 3) + 4 * (p16 & 1)};
    const int off16_q1 = row16 * B_QLD + 4 * p16, off16_q2 = (row16 + 8) * B_QLD + 4 * p16;      // + 16 (d half)
    const unsigned short* x_tr = Xw + (4 * h + (li >> 2)) * XLD + 16 * g4 + 4 * (li & 3);          // dS^T: + 32 k2 XLD + 16 s XLD
    unsigned short* x_wr = Xw + r * XLD + 4 * h;
    const unsigned short* v_rows = Vw + r * DLD + 8 * h;
    float* xq_base = reinterpret_cast<float*>(lds + OFF_XQ);

    // hand the previous tile's dQ over: sum the four waves' fp32 parts (all LDS reads first: ONE round trip), then
    // atomics / one bf16 part store
    auto dq_handover = [&](int buf, int tq) {
        const float* xr = xq_base + buf * XQ_BUF;
        if constexpr (ATOMIC) {
            float* acc = reinterpret_cast<float*>(dq_out) + (nb + (long)tq * 32) * 32;
            float v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int w = 0; w < 4; ++w) v[k][w] = xr[w * 32 * QXLD + (8 * wave + 2 * k + h) * QXLD + r];
#pragma unroll
            for (int k = 0; k < 4; ++k)      // one wave-instruction = two adjacent 128-byte rows
                atomicAdd(acc + (8 * wave + 2 * k + h) * 32 + r, (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]));
        } else {
            unsigned short* part = reinterpret_cast<unsigned short*>(dq_out) + (((long)b * (Npad / 256) + kb) * Npad + (long)tq * 32) * 32;
            const int i = 8 * wave + (lane >> 3), d4 = (lane & 7) * 4;
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) a += *reinterpret_cast<const f32x4_t*>(xr + w * 32 * QXLD + i * QXLD + d4);
            const u32x2_t o = {pack2<false>(a[0], a[1]), pack2<false>(a[2], a[3])};
            *reinterpret_cast<u32x2_t*>(part + i * 32 + d4) = o;
        }
    };

    // VFORM: the same hand-over in two halves -- the exchange-buffer reads are requested at the top of the iteration
    // (hand-issued: no wait next to them), the sums and the VMEM operations follow a few MFMA steps later
    struct HqRegs {
        float v[4][4];
        f32x4_t q[4];
    };
    auto hq_issue = [&](HqRegs& hq, int buf) {
        if constexpr (ATOMIC) {
            const unsigned int a = lds_addr(xq_base + buf * XQ_BUF + (8 * wave + h) * QXLD + r);
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value, k = i >> 2, w = i & 3;
                hq.v[k][w] = lds_issue32f<(w * 32 * QXLD + 2 * k * QXLD) * 4>(a);
            });
        } else {
            const unsigned int a = lds_addr(xq_base + buf * XQ_BUF + (8 * wave + (lane >> 3)) * QXLD + (lane & 7) * 4);
            static_for<0, 4>([&](auto ic) {
                constexpr int w = decltype(ic)::value;
                hq.q[w] = lds_issue128f<w * 32 * QXLD * 4>(a);
            });
        }
    };
    auto hq_commit = [&](HqRegs& hq, int tq) {
        if constexpr (ATOMIC) {
            float* acc = reinterpret_cast<float*>(dq_out) + (nb + (long)tq * 32) * 32;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                atomicAdd(acc + (8 * wave + 2 * k + h) * 32 + r, (hq.v[k][0] + hq.v[k][1]) + (hq.v[k][2] + hq.v[k][3]));
        } else {
            unsigned short* part = reinterpret_cast<unsigned short*>(dq_out) + (((long)b * (Npad / 256) + kb) * Npad + (long)tq * 32) * 32;
            const int i = 8 * wave + (lane >> 3), d4 = (lane & 7) * 4;
            const f32x4_t a = (hq.q[0] + hq.q[1]) + (hq.q[2] + hq.q[3]);
            const u32x2_t o = {pack2<false>(a[0], a[1]), pack2<false>(a[2], a[3])};
            *reinterpret_cast<u32x2_t*>(part + i * 32 + d4) = o;
        }
    };

    // DQ16: the A fragments (dS[query][key], 16 queries x 32 keys per k-step) of the PREVIOUS tile by transpose reads of all
    // four waves' dS^T images: lane group g = lane >> 4 takes the stored key positions 8g .. 8g+7 of a 32-key block, i.e. (kn
    // is perm16) keys base_g + {0..3} and base_g + 8 + {0..3} with base_g = 4 (g & 1) + 16 (g >> 1): two reads per k-step;
    // the B fragments (K[key][d]) are 16-byte reads of the K^T image.  Two batches of four k-steps (32 registers in flight).
    struct Dq16Regs {
        s16x4_t f[8];
        bf16x8_t kb[4];
    };
    const unsigned int a_dq16 = lds_addr(lds + OFF_X + (4 * ((lane >> 4) & 1) + 16 * (lane >> 5) + (li >> 2)) * XLD + 16 * (wave & 1) + 4 * (li & 3));
    const unsigned int a_k16 = lds_addr(lds + OFF_K + (16 * (wave >> 1) + (lane & 15)) * KLD + 8 * (lane >> 4));
    auto dq16_issue = [&](Dq16Regs& dr, int buf, auto batch) {
        constexpr int B0 = decltype(batch)::value * 4;
        const unsigned int a = a_dq16 + (unsigned int)buf * (4 * XW * 2);
        static_for<0, 4>([&](auto ic) {
            constexpr int kp = B0 + decltype(ic)::value;
            constexpr int off = ((kp >> 1) * XW + (kp & 1) * 32 * XLD) * 2;
            dr.f[2 * (kp - B0)] = tr_issue<off>(a);
            dr.f[2 * (kp - B0) + 1] = tr_issue<off + 8 * XLD * 2>(a);
            dr.kb[kp - B0] = lds_issue128<kp * 64>(a_k16);
        });
    };
    auto dq16_land = [&](Dq16Regs& dr) {          // a counted wait in front of this has covered the twelve reads
        asm volatile("" : "+v"(dr.f[0]), "+v"(dr.f[1]), "+v"(dr.f[2]), "+v"(dr.f[3]), "+v"(dr.f[4]), "+v"(dr.f[5]), "+v"(dr.f[6]),
                          "+v"(dr.f[7]), "+v"(dr.kb[0]), "+v"(dr.kb[1]), "+v"(dr.kb[2]), "+v"(dr.kb[3]));
    };
    auto dq16_mma = [&](Dq16Regs& dr, f32x4_t& acc, auto batch) {
        static_for<0, 4>([&](auto ic) {
            constexpr int kk = decltype(ic)::value;
            const bf16x8_t a = tr_frag(dr.f[2 * kk], dr.f[2 * kk + 1]);
            if constexpr (decltype(batch)::value == 0 && kk == 0) {
                if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(dr.kb[kk]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(dr.kb[kk]));
            } else {
                if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(dr.kb[kk]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(dr.kb[kk]));
            }
        });
    };
    auto dq16_commit = [&](f32x4_t& acc, int tq) {      // rows = queries 16 (wave & 1) + 4 (lane >> 4) + e, columns = d
        const int qrow = 16 * (wave & 1) + 4 * (lane >> 4), dcol = 16 * (wave >> 1) + (lane & 15);
        if constexpr (ATOMIC) {
            float* o = reinterpret_cast<float*>(dq_out) + (nb + (long)tq * 32 + qrow) * 32 + dcol;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(o + e * 32, acc[e]);      // one wave-instruction = four 64-byte row segments
        } else {
            unsigned short* part = reinterpret_cast<unsigned short*>(dq_out) + (((long)b * (Npad / 256) + kb) * Npad + (long)tq * 32 + qrow) * 32 + dcol;
#pragma unroll
            for (int e = 0; e < 4; ++e) part[e * 32] = (unsigned short)(pack2<false>(acc[e], 0.f) & 0xFFFFu);
        }
    };

    // the key-side operands above are complete: said with the BUILTIN so that the compiler's own wait-count pass knows it
    // (it cannot see a wait inside asm text and would otherwise put vmcnt(0) in front of their first use in the loop,
    // draining the DMA ring and the dQ atomics every tile)
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    dma_seek(tcur);
    int tpf = tcur;                          // the tile the DMA pointers stand on
    dma_next(tpf, 0);
    tpf = next_tile(tpf);
    dma_next(tpf, 1);
    tpf = next_tile(tpf);

    unsigned long long clk0 = 0, rt0 = 0;
    if constexpr (STAMP) {           // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz over the whole sweep
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0), "=s"(rt0) :: "memory");
    }
    for (int it = 0; it < nqt; ++it) {
        // tile `it` landed: this wave's pieces by its own counted wait, the other waves' by the barrier behind it.
        // Younger VMEM operations than DMA(it): [hand-over(it-3)] DMA(it+1) [hand-over(it-2)]
        if (it < 2) wait_vmcnt<PPW>();
        else wait_vmcnt<PPW + 2 * AOPS>();
        if constexpr (STAMP) {
            // the previous iteration's stamps have landed behind this wait; st[0] = the end stamp of the iteration before it
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st[1]), "+s"(st[2]), "+s"(st[3]), "+s"(st[4]), "+s"(st[5]), "+s"(st[6]), "+s"(stw) :: "memory");
            if (it > 1) {
                sacc_t[0] += (unsigned int)(st[1] - st[0]);          // loop back + counted waits + barrier
                sacc_w += (unsigned int)(stw - st[0]);               //   of which: loop back + the counted vmcnt / lgkmcnt waits
#pragma unroll
                for (int k = 1; k < 6; ++k) sacc_t[k] += (unsigned int)(st[k + 1] - st[k]);
            }
            st[0] = st[6];
            asm volatile("s_memtime %0" : "=s"(stw));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's exchange-buffer writes are in LDS
        }
        __builtin_amdgcn_s_barrier();
        stamp(IC<1>{});
        const int slot = it % NSLOT;
        // every LDS address below = (ring slot base) + (per-lane offset fixed for the whole sweep) + (compile-time
        // immediate): the XOR swizzle of the dO image only touches the low two chunk bits, so it folds into the
        // per-lane part
        const unsigned short* Qs = lds + slot * SLOT;
        const unsigned short* dOs = Qs + QCH * 8;
        const float* RC = reinterpret_cast<const float*>(dOs + DCH * 8);
        const unsigned short* q_rows = Qs + off_qrow;
        const unsigned short* do_rows_e = dOs + off_do_e;
        const unsigned short* do_rows_o = dOs + off_do_o;
        const unsigned short* do_tr_lo = dOs + off_trd_lo;
        const unsigned short* do_tr_hi = dOs + off_trd_hi;
        const unsigned short* q_tr = Qs + off_trq;

        // VFORM: the row constants and the Q rows are requested right behind the barrier, in front of the DMA issue
        f32x4_t rc4[8];
        bf16x8_t qa0, qa1;
        HqRegs hq;
        Dq16Regs dqr;
        f32x4_t dq4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (VFORM) {
            const unsigned int a_rc = lds_addr(RC + 4 * h), a_qr = lds_addr(q_rows);
            rc4[0] = lds_issue128f<0>(a_rc);
            rc4[1] = lds_issue128f<32>(a_rc);
            rc4[2] = lds_issue128f<64>(a_rc);
            rc4[3] = lds_issue128f<96>(a_rc);
            qa0 = lds_issue128<0>(a_qr);
            qa1 = lds_issue128<32>(a_qr);
            rc4[4] = lds_issue128f<128>(a_rc);
            rc4[5] = lds_issue128f<160>(a_rc);
            rc4[6] = lds_issue128f<192>(a_rc);
            rc4[7] = lds_issue128f<224>(a_rc);
        }
        // VMEM order per iteration (the vmcnt count above relies on it): DMA of tile it+2 into the slot of tile it-1
        // (every wave is past its reads of it: it is past this barrier), THEN the dQ hand-over of tile it-1
        // (VFORM issues both from inside the dP phase, in this order, under its MFMAs)
        if constexpr (!DMA_LATE && !VFORM) {
            dma_next(tpf, (it + 2) % NSLOT);
            tpf = next_tile(tpf);
        }
        if constexpr (!HANDOVER_MID && !VFORM) dq_handover((it + 1) & 1, tprev);    // buffer written in iteration it-1

        f32x16_t sacc[2], dpacc[2];
        u32x4_t pw[2][2];                              // VFORM: P fragments, packed chunk by chunk under the dP MFMAs
        if constexpr (VFORM) {
            constexpr int DPD = 2;                         // dO row fragments in flight ahead of the MFMAs
            const unsigned int a_de = lds_addr(do_rows_e), a_do = lds_addr(do_rows_o);
            bf16x8_t dq_[DPD + 1];
            auto do_issue = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                dq_[s % (DPD + 1)] = (s & 1) ? lds_issue128<(s >> 1) * 64>(a_do) : lds_issue128<(s >> 1) * 64>(a_de);
            };
            static_for<0, DPD>([&](auto sc) { do_issue(sc); });
            asm volatile("s_waitcnt lgkmcnt(%10)"
                         : "+v"(rc4[0]), "+v"(rc4[1]), "+v"(rc4[2]), "+v"(rc4[3]), "+v"(rc4[4]), "+v"(rc4[5]), "+v"(rc4[6]),
                           "+v"(rc4[7]), "+v"(qa0), "+v"(qa1)
                         : "n"(DPD)
                         : "memory");
            f32x16_t rcA, rcD;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rcA[4 * g + e] = rc4[g][e];
                    rcD[4 * g + e] = rc4[4 + g][e];
                }
            mfma_v_first2x2<F16>(sacc[0], sacc[1], rcA, qa0, kfB[0][0], kfB[1][0], qa1, kfB[0][1], kfB[1][1]);
            stamp(IC<2>{});
            if constexpr (DMA_LATE) {
                dma_next(tpf, (it + 2) % NSLOT);
                tpf = next_tile(tpf);
            }
            // dP = dO V^T - delta: 2 CT steps of two MFMAs; the dO row fragment of step s + 2 is requested before the
            // MFMAs of step s, and the exp / pack of P (eight chunks of four elements) is written between the steps it
            // should run under -- asm statements carry no latency for the scheduler, so the order here IS the schedule;
            // mfma_tie() pins it (no instruction, only a dependency through both operands)
            auto p_chunk = [&](auto cc) {          // P = exp2(S) and its 16-bit pack, four accumulator registers at a time
                constexpr int c = decltype(cc)::value, k2 = c >> 2, q = c & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc[k2][4 * q + e] = gd_exp2_fast(sacc[k2][4 * q + e]);
                pw[k2][q >> 1][2 * (q & 1)] = pack2<F16>(sacc[k2][4 * q], sacc[k2][4 * q + 1]);
                pw[k2][q >> 1][2 * (q & 1) + 1] = pack2<F16>(sacc[k2][4 * q + 2], sacc[k2][4 * q + 3]);
            };
            static_for<0, 2 * CT>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                auto v_frag = [&](int k2) {
                    if (k2 < VREG) return vB[k2 < VREG ? k2 : 0][s];
                    return *reinterpret_cast<const bf16x8_t*>(v_rows + (k2 - VREG) * 32 * DLD + s * 16);
                };
                if constexpr (s == 0) {
                    // steps 0 and 1 go out together at s == 1 (mfma_v_first2x2); meanwhile, under the S MFMAs, the
                    // exchange-buffer reads of the hand-over (buffer written in iteration it-1) and fragment 2 are requested
                    if constexpr (DQ16) dq16_issue(dqr, (it + 1) & 1, IC<0>{});
                    else hq_issue(hq, (it + 1) & 1);
                    if constexpr (DPD < 2 * CT) do_issue(IC<DPD>{});
                } else if constexpr (s == 1) {
                    // fragments 0 and 1 landed: the reads behind them are fragment 2 (if any)
                    // (behind them: the hand-over reads and fragment 2 -- more than the counter can express: 15 is stricter)
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(dq_[0]), "+v"(dq_[1]) : "n"(DQ16 ? 13 : ATOMIC ? 15 : (DPD < 2 * CT ? 5 : 4)) : "memory");     // DQ16: 12 batch reads + fragment 2 behind them
                    bf16x8_t v00, v10;
                    if (0 < VREG) v00 = vB[0][0];
                    else v00 = *reinterpret_cast<const bf16x8_t*>(v_rows + (0 - VREG) * 32 * DLD);
                    if (1 < VREG) v10 = vB[VREG > 1 ? 1 : 0][0];
                    else v10 = *reinterpret_cast<const bf16x8_t*>(v_rows + (1 - VREG) * 32 * DLD);
                    mfma_v_first2x2<F16>(dpacc[0], dpacc[1], rcD, dq_[0], v00, v10, dq_[1], v_frag(0), v_frag(1));
                    if constexpr (1 + DPD < 2 * CT) do_issue(IC<1 + DPD>{});     // into the ring slot of fragment 0
                } else {
                    if constexpr (s + DPD < 2 * CT) do_issue(IC<s + DPD>{});
                    // reads issued after fragment s (DQ16: batch 1's twelve reads sit behind fragments 3 and 4)
                    constexpr int behind = (DQ16 && (s == 3 || s == 4)) ? 14 : ((2 * CT - 1 - s) < DPD ? (2 * CT - 1 - s) : DPD);
                    lds_wait128<behind>(dq_[s % (DPD + 1)]);
                    const bf16x8_t da = dq_[s % (DPD + 1)];
                    mfma_v_acc<F16>(dpacc[0], da, v_frag(0));
                    mfma_v_acc<F16>(dpacc[1], da, v_frag(1));
                }
                // S is final four MFMAs (> 11 quad-cycles) behind its last write at s == 1: chunk c runs under step c + 1
                if constexpr (s >= 1 && s <= 8) {
                    mfma_pad4<false>(sacc[0], sacc[1], dpacc[0], dpacc[1]);
                    p_chunk(IC<s - 1>{});
                    mfma_pad4<false>(sacc[0], sacc[1], dpacc[0], dpacc[1]);
                }
                // DMA of tile it+2, then the hand-over's VMEM operations: the order the vmcnt count at the loop top relies on
                if constexpr (s == (2 * CT > 2 ? 2 : 2 * CT - 1)) {
                    dma_next(tpf, (it + 2) % NSLOT);
                    tpf = next_tile(tpf);
                }
                // hand the exchange-buffer values to the compiler as soon as they are known to have landed: they are older
                // than fragment 2, whose wait has just passed (CT == 1 has no fragment 2: waited for here)
                if constexpr (s == (2 * CT > 2 ? 2 : 1)) {
                    if constexpr (2 * CT <= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (DQ16) {
                        dq16_land(dqr);
                        dq16_mma(dqr, dq4, IC<0>{});                       // four 16-cycle MFMAs between dP steps
                        dq16_issue(dqr, (it + 1) & 1, IC<1>{});            // batch 1 into the same registers
                    } else if constexpr (ATOMIC)
                        asm volatile("" : "+v"(hq.v[0][0]), "+v"(hq.v[0][1]), "+v"(hq.v[0][2]), "+v"(hq.v[0][3]), "+v"(hq.v[1][0]),
                                          "+v"(hq.v[1][1]), "+v"(hq.v[1][2]), "+v"(hq.v[1][3]), "+v"(hq.v[2][0]), "+v"(hq.v[2][1]),
                                          "+v"(hq.v[2][2]), "+v"(hq.v[2][3]), "+v"(hq.v[3][0]), "+v"(hq.v[3][1]), "+v"(hq.v[3][2]),
                                          "+v"(hq.v[3][3]));
                    else
                        asm volatile("" : "+v"(hq.q[0]), "+v"(hq.q[1]), "+v"(hq.q[2]), "+v"(hq.q[3]));
                }
                if constexpr (DQ16 && s == 5) {             // fragment 5's wait (2 behind) has covered batch 1
                    dq16_land(dqr);
                    dq16_mma(dqr, dq4, IC<1>{});
                }
                if constexpr (DQ16 && s == 7) {
                    // the sub-tile is final two dP steps (>= 64 cycles) behind its last MFMA; the tie keeps the adds here
                    asm volatile("" : "+v"(dq4), "+v"(dpacc[0]));
                    dq16_commit(dq4, tprev);
                }
                if constexpr (!DQ16 && s == (2 * CT > 4 ? 4 : 2 * CT - 1)) hq_commit(hq, tprev);
            });
            static_for<(2 * CT - 1 < 8 ? 2 * CT - 1 : 8), 8>(p_chunk);      // narrow C: the chunks no step was left for
            // the wait states between the last dP MFMA and the first VALU read of dP; PH2 reads it two MFMAs later
            mfma_pad<!PH2>(dpacc[0], dpacc[1]);
            stamp(IC<3>{});
        } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(RC + 8 * g + 4 * h);
            const f32x4_t d = *reinterpret_cast<const f32x4_t*>(RC + 32 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sacc[0][4 * g + e] = a[e];
                sacc[1][4 * g + e] = a[e];
                dpacc[0][4 * g + e] = d[e];
                dpacc[1][4 * g + e] = d[e];
            }
        }
        // S = Q K^T - lse (log2 domain)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(q_rows + s * 16);
            sacc[0] = mfma16<F16>(qa, kfB[0][s], sacc[0]);
            sacc[1] = mfma16<F16>(qa, kfB[1][s], sacc[1]);
        }
        if constexpr (DMA_LATE) {
            dma_next(tpf, (it + 2) % NSLOT);
            tpf = next_tile(tpf);
        }
        // dP = dO V^T - delta
#pragma unroll
        for (int s = 0; s < 2 * CT; ++s) {
            const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(((s & 1) ? do_rows_o : do_rows_e) + (s >> 1) * 32);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                bf16x8_t vb;
                if (k2 < VREG) vb = vB[k2 < VREG ? k2 : 0][s];
                else vb = *reinterpret_cast<const bf16x8_t*>(v_rows + (k2 - VREG) * 32 * DLD + s * 16);
                dpacc[k2] = mfma16<F16>(da, vb, dpacc[k2]);
            }
        }
        }
        // ---- second half: dV^T += dO^T P (CT steps), dK^T += Q^T dS (1 step), dQ^T part (2 steps) -------------------
        // every A/B fragment below comes from a transpose read (element j of lane half h <-> query 16s + 8(j>>2) + 4h +
        // (j&3)), issued one step (4 reads = both k-steps of one tile) ahead of the MFMAs that consume it.  Step ids:
        // 0..CT-1 dV^T channel tile, CT dK^T, CT+1 / CT+2 dQ^T over the wave's key tile 0 / 1.
        const unsigned int a_dlo = lds_addr(do_tr_lo), a_dhi = lds_addr(do_tr_hi);
        const unsigned int a_q = lds_addr(q_tr), a_x = lds_addr(x_tr);
        TrFrag2 fb[CT + 3];
        const unsigned int a16_d1[2] = {lds_addr(dOs + off16_d1[0]), lds_addr(dOs + off16_d1[1])};
        const unsigned int a16_d2[2] = {lds_addr(dOs + off16_d2[0]), lds_addr(dOs + off16_d2[1])};
        const unsigned int a16_q1 = lds_addr(Qs + off16_q1), a16_q2 = lds_addr(Qs + off16_q2);
        auto issue = [&](auto idc) {
            constexpr int id = decltype(idc)::value;
            if constexpr (DV16_V && id < CT) {            // lo0 / hi0: channels 32 id + 0..15, lo1 / hi1: + 16..31
                fb[id].lo0 = tr_issue<id * 64>(a16_d1[0]);
                fb[id].hi0 = tr_issue<id * 64>(a16_d2[0]);
                fb[id].lo1 = tr_issue<id * 64>(a16_d1[1]);
                fb[id].hi1 = tr_issue<id * 64>(a16_d2[1]);
            } else if constexpr (DV16_K && id == CT) {    // Q^T: d 0..15 / 16..31
                fb[id].lo0 = tr_issue<0>(a16_q1);
                fb[id].hi0 = tr_issue<0>(a16_q2);
                fb[id].lo1 = tr_issue<32>(a16_q1);
                fb[id].hi1 = tr_issue<32>(a16_q2);
            } else if constexpr (id < CT) fb[id] = tr_issue2<id * 64, 16 * DOLD * 2>(a_dlo, a_dhi);
            else if constexpr (id == CT) fb[id] = tr_issue2<0, 16 * B_QLD * 2>(a_q, a_q + 8 * B_QLD * 2);
            else fb[id] = tr_issue2<(id - CT - 1) * 32 * XLD * 2, 16 * XLD * 2>(a_x, a_x + 8 * XLD * 2);
        };
        constexpr bool HOIST = PH2 || (ORDER & 128) != 0;   // bit 7: only the hoisted first transpose reads of bit 5
        if constexpr (HOIST) issue(IC<0>{});           // dO^T of channel tile 0: in flight under the dS arithmetic below
        bf16x8_t pf[2][2], dsf[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            if constexpr (VFORM) {
                pf[k2][0] = __builtin_bit_cast(bf16x8_t, pw[k2][0]);
                pf[k2][1] = __builtin_bit_cast(bf16x8_t, pw[k2][1]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[k2][e] = gd_exp2_fast(sacc[k2][e]);   // P
                pf[k2][0] = pack_frag<F16>(sacc[k2], 0);
                pf[k2][1] = pack_frag<F16>(sacc[k2], 1);
            }
        }
        // dS = P (dP - delta), its 16-bit pack and the dS^T image X[key][query] (this lane's 16 queries are 4 runs of 4),
        // four accumulator registers at a time: chunk c = key tile c >> 2, registers 4 (c & 3) ..
        u32x4_t dsw[2][2];
        unsigned short* x_wr_it = x_wr + (DQ16 ? (it & 1) * (4 * XW) : 0);       // DQ16: this tile's image of the double buffer
        auto ds_chunk = [&](auto cc) {
            constexpr int c = decltype(cc)::value, k2 = c >> 2, q = c & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NOPK) {
                    float t = dpacc[k2][4 * q + e];
                    asm("v_mul_f32 %0, %0, %1" : "+v"(t) : "v"(sacc[k2][4 * q + e]));
                    dpacc[k2][4 * q + e] = t;
                } else {
                    dpacc[k2][4 * q + e] *= sacc[k2][4 * q + e];
                }
            }
            dsw[k2][q >> 1][2 * (q & 1)] = pack2<F16>(dpacc[k2][4 * q], dpacc[k2][4 * q + 1]);
            dsw[k2][q >> 1][2 * (q & 1) + 1] = pack2<F16>(dpacc[k2][4 * q + 2], dpacc[k2][4 * q + 3]);
            if constexpr ((q & 1) == 1) {              // k-step q >> 1 of key tile k2 is complete
                const u32x4_t w = dsw[k2][q >> 1];
                const u32x2_t lo = {w.x, w.y}, hi = {w.z, w.w};
                *reinterpret_cast<u32x2_t*>(x_wr_it + k2 * 32 * XLD + 16 * (q >> 1)) = lo;
                *reinterpret_cast<u32x2_t*>(x_wr_it + k2 * 32 * XLD + 16 * (q >> 1) + 8) = hi;
                dsf[k2][q >> 1] = __builtin_bit_cast(bf16x8_t, w);
            }
        };
        if constexpr (DV16_V) {                           // P of each key tile -> the B operands of its two 16-key halves
            swap16_frags(pf[0][0], pf[0][1]);
            swap16_frags(pf[1][0], pf[1][1]);
        }
        if constexpr (!PH2) static_for<0, 8>(ds_chunk);
        if constexpr (DV16_K) {                           // dS likewise (the dS^T image above was written from the plain words)
            swap16_frags(dsf[0][0], dsf[0][1]);
            swap16_frags(dsf[1][0], dsf[1][1]);
        }
        stamp(IC<4>{});
        f32x16_t dqp;
#pragma unroll
        for (int e = 0; e < 16; ++e) dqp[e] = 0.f;
        auto compute = [&](auto idc) {
            constexpr int id = decltype(idc)::value;
            const bf16x8_t a0 = tr_frag(fb[id].lo0, fb[id].hi0), a1 = tr_frag(fb[id].lo1, fb[id].hi1);
            if constexpr (DV16_V && id < CT) {
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {        // (key tile, half): eight 16-cycle MFMAs on eight accumulators
                    dv16[id][0][i_ >> 1][i_ & 1] = mfma16x16<F16>(a0, pf[i_ >> 1][i_ & 1], dv16[id][0][i_ >> 1][i_ & 1]);
                    dv16[id][1][i_ >> 1][i_ & 1] = mfma16x16<F16>(a1, pf[i_ >> 1][i_ & 1], dv16[id][1][i_ >> 1][i_ & 1]);
                }
            } else if constexpr (DV16_K && id == CT) {
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    dk16[0][i_ >> 1][i_ & 1] = mfma16x16<F16>(a0, dsf[i_ >> 1][i_ & 1], dk16[0][i_ >> 1][i_ & 1]);
                    dk16[1][i_ >> 1][i_ & 1] = mfma16x16<F16>(a1, dsf[i_ >> 1][i_ & 1], dk16[1][i_ >> 1][i_ & 1]);
                }
            } else if constexpr (id < CT) {
                dvacc[0][id] = mfma16<F16>(a0, pf[0][0], dvacc[0][id]);
                dvacc[1][id] = mfma16<F16>(a0, pf[1][0], dvacc[1][id]);
                dvacc[0][id] = mfma16<F16>(a1, pf[0][1], dvacc[0][id]);
                dvacc[1][id] = mfma16<F16>(a1, pf[1][1], dvacc[1][id]);
            } else if constexpr (id == CT) {
                dkacc[0] = mfma16<F16>(a0, dsf[0][0], dkacc[0]);
                dkacc[1] = mfma16<F16>(a0, dsf[1][0], dkacc[1]);
                dkacc[0] = mfma16<F16>(a1, dsf[0][1], dkacc[0]);
                dkacc[1] = mfma16<F16>(a1, dsf[1][1], dkacc[1]);
            } else {
                // transpose read of X = the B operand (lane = query, k = key)
                dqp = mfma16<F16>(knA[id - CT - 1][0], a0, dqp);
                dqp = mfma16<F16>(knA[id - CT - 1][1], a1, dqp);
            }
        };
        auto xq_write = [&]() {
            float* xw = xq_base + (it & 1) * XQ_BUF + wave * 32 * QXLD + r * QXLD + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t v = {dqp[4 * g], dqp[4 * g + 1], dqp[4 * g + 2], dqp[4 * g + 3]};
                *reinterpret_cast<f32x4_t*>(xw + 8 * g) = v;
            }
        };
        constexpr int NSTEP = DQ16 ? CT + 1 : CT + 3;       // DQ16: the dQ^T steps live in the NEXT iteration's dP phase
        if constexpr (PH2) {
            // dV^T steps as asm MFMA pairs in source order = schedule; the eight dS chunks are spread over the gaps
            // behind the pairs of the first steps (NCH per step, half behind each pair)
            constexpr int NCH = (8 + CT - 1) / CT;
            static_for<0, CT>([&](auto ic) {
                constexpr int ct = decltype(ic)::value;
                issue(IC<ct + 1>{});
                tr_wait<4>(fb[ct]);
                const bf16x8_t a0 = tr_frag(fb[ct].lo0, fb[ct].hi0), a1 = tr_frag(fb[ct].lo1, fb[ct].hi1);
                mfma_a_acc<F16>(dvacc[0][ct], a0, pf[0][0]);
                mfma_a_acc<F16>(dvacc[1][ct], a0, pf[1][0]);
                static_for<0, (NCH + 1) / 2>([&](auto jc) {
                    constexpr int c = ct * NCH + decltype(jc)::value;
                    if constexpr (c < 8) {
                        mfma_tie_a(dvacc[0][ct], dvacc[1][ct], dpacc[c >> 2]);
                        ds_chunk(IC<c>{});
                        mfma_tie_a(dvacc[0][ct], dvacc[1][ct], dpacc[c >> 2]);
                    }
                });
                mfma_a_acc<F16>(dvacc[0][ct], a1, pf[0][1]);
                mfma_a_acc<F16>(dvacc[1][ct], a1, pf[1][1]);
                static_for<(NCH + 1) / 2, NCH>([&](auto jc) {
                    constexpr int c = ct * NCH + decltype(jc)::value;
                    if constexpr (c < 8) {
                        mfma_tie_a(dvacc[0][ct], dvacc[1][ct], dpacc[c >> 2]);
                        ds_chunk(IC<c>{});
                        mfma_tie_a(dvacc[0][ct], dvacc[1][ct], dpacc[c >> 2]);
                    }
                });
            });
            stamp(IC<5>{});
            static_for<CT, NSTEP>([&](auto ic) {
                constexpr int id = decltype(ic)::value;
                if constexpr (id + 1 < NSTEP) issue(IC<id + 1>{});
                tr_wait<(id + 1 < NSTEP ? 4 : 0)>(fb[id]);
                compute(IC<id>{});
                if constexpr (id == CT + 2) xq_write();
            });
        } else {
        // DQ_FIRST: the dQ^T steps (and the exchange-buffer write) go in front of the dV^T / dK^T steps, so that the
        // iteration ends on independent MFMAs instead of a dependent chain + an LDS write in front of the barrier
        auto step_id = [](int i) constexpr { return DQ_FIRST ? (i < 2 ? CT + 1 + i : i - 2) : i; };
        // transpose reads run LA steps ahead of their MFMAs (2 measured equal to 1: the waits are not what stalls)
        constexpr int LA = 1;
        if constexpr (!HOIST) static_for<0, (LA < NSTEP ? LA : NSTEP)>([&](auto ic) { issue(IC<step_id(decltype(ic)::value)>{}); });
        static_for<0, NSTEP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int id = step_id(i);
            if constexpr (i + LA < NSTEP) issue(IC<step_id(i + LA)>{});
            constexpr int ahead = (NSTEP - 1 - i) < LA ? (NSTEP - 1 - i) : LA;      // steps whose reads are behind this one
            if constexpr (i == CT) stamp(IC<5>{});
            tr_wait<4 * ahead>(fb[id]);
            compute(IC<id>{});
            if constexpr (id == CT + 2) {       // dQ^T part complete
                xq_write();
                if constexpr (HANDOVER_MID) dq_handover((it + 1) & 1, tprev);
            }
        });
        }
        stamp(IC<6>{});
        tprev = tcur;
        tcur = next_tile(tcur);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DQ16) {
        Dq16Regs dqr;
        f32x4_t dq4;
        dq16_issue(dqr, (nqt + 1) & 1, IC<0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dq16_land(dqr);
        dq16_mma(dqr, dq4, IC<0>{});
        dq16_issue(dqr, (nqt + 1) & 1, IC<1>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dq16_land(dqr);
        dq16_mma(dqr, dq4, IC<1>{});
        asm volatile("s_nop 10" : "+v"(dq4));
        dq16_commit(dq4, tprev);
    } else {
        dq_handover((nqt + 1) & 1, tprev);
    }
    if constexpr (STAMP) {
        if (dbg && lane == 0) {
            unsigned long long clk1, rt1;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1), "=s"(rt1) :: "memory");
            unsigned int* o = dbg + (((long)b * gridDim.x + kb) * 4 + wave) * 12;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = sacc_t[k];
            o[6] = sacc_w;
            o[7] = (unsigned int)(nqt - 2);
            o[8] = (unsigned int)(clk1 - clk0);
            o[9] = (unsigned int)(rt1 - rt0);
            o[10] = o[11] = 0;
        }
    }

    if constexpr (DV16) {
        // 16 x 16 tiles: row 4 (lane >> 4) + e = channel (d) inside the 16-block, column lane & 15 = key inside the half
        const int g16 = lane >> 4, c16 = lane & 15;
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) {
            const int j = j0 + 32 * (i_ >> 1) + 16 * (i_ & 1) + c16;
#pragma unroll
            for (int ct = 0; ct < (DV16_V ? CT : 0); ++ct)
#pragma unroll
                for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dv[(long)b * dv_bs + (long)(ct * 32 + 16 * cp + 4 * g16 + e) * Npad + j] = dv16[ct][cp][i_ >> 1][i_ & 1][e];
#pragma unroll
            for (int db = 0; db < (DV16_K ? 2 : 0); ++db)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dkn[(long)b * dk_bs + (long)(16 * db + 4 * g16 + e) * Npad + j] = dk16[db][i_ >> 1][i_ & 1][e] * LN2;
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const int j = j0 + 32 * k2 + r;
#pragma unroll
        for (int ct = 0; ct < (DV16_V ? 0 : CT); ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) dv[(long)b * dv_bs + (long)(ct * 32 + acc_row(e, h)) * Npad + j] = dvacc[k2][ct][e];
        if constexpr (!DV16_K)
#pragma unroll
            for (int e = 0; e < 16; ++e) dkn[(long)b * dk_bs + (long)acc_row(e, h) * Npad + j] = dkacc[k2][e] * LN2;   // Q^T was q * log2 e
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------
// scratch of one image: row constants (Npad/32 x 64 floats) + dQ accumulator (Npad x 32 floats, ATOMIC) or the bf16
// dQ parts (Npad/256 key blocks x Npad x 32, deterministic)
extern "C" size_t gd_pam_bwd64_scratch_bytes(int Npad, int deterministic) {
    const size_t rcb = (size_t)Npad * 2 * sizeof(float);
    const size_t dq = deterministic ? (size_t)(Npad / 256) * (size_t)Npad * 32 * sizeof(unsigned short)
                                    : (size_t)Npad * 32 * sizeof(float);
    return rcb + dq;
}

namespace {
template <int CT, bool F16, int VREG, bool ATOMIC>
void launch_k64(dim3 grid, hipStream_t s, const unsigned short* q, const unsigned short* k, const unsigned short* kT,
                const unsigned short* v, const unsigned short* dO, const float* rc, int Npad, float* dkn, float* dv,
                void* dq_out, long dk_bs, long dv_bs) {
    hipLaunchKernelGGL((pam_bwd_k64_kernel<CT, F16, VREG, ATOMIC, 8>), grid, dim3(256), 0, s, q, k, kT, v, dO, rc, Npad, dkn, dv, dq_out, dk_bs, dv_bs,
                       (unsigned int*)nullptr);
}
}  // namespace

extern "C" void gd_pam_dq_reduce_launch(const void* part, int KB, int Npad, int nb, float* dqn, void* stream);   // pam.hip

// bench tooling: schedule variant of the K64 kernel (0 = production = VGPR-form S / dP tiles with the hand-placed
// dP phase, 1 = the compiler-scheduled AGPR-form loop it replaced) and V-in-registers count (0 = default)
static int g_k64_order = 0, g_k64_vreg = 0;
static unsigned int* g_k64_dbg = nullptr;
extern "C" void gd_pam_k64_variant(int order, int vreg) {
    g_k64_order = order;
    g_k64_vreg = vreg;
}
// diagnostic builds (variants 5, 6): per-wave cycle sums of the tile loop's segments, (images x key blocks x 4 waves x 8) words
extern "C" void gd_pam_k64_debug(void* buf) { g_k64_dbg = (unsigned int*)buf; }

// one batch slice through the 64-keys-per-wave backward; scratch holds `images` images' worth
extern "C" int gd_pam_bwd64_slice(const void* qt, const void* kt, const void* kn, const void* vt, const void* dot_,
                                  const float* lse, const float* delta, int nb, int N, int Npad, int Cp, int f16,
                                  int vreg, int deterministic, float* dqn, float* dkn, float* dv, long out_bs,
                                  void* scratch, void* stream) {
    // out_bs: batch stride (elements) shared by dqn / dkn / dv when they are row blocks of ONE (B, rows, Npad) buffer;
    // 0 = three dense tensors
    const long dq_bs = out_bs ? out_bs : 32L * Npad, dk_bs = out_bs ? out_bs : 32L * Npad, dv_bs = out_bs ? out_bs : (long)Cp * Npad;
    hipStream_t s = (hipStream_t)stream;
    if (g_k64_vreg) vreg = g_k64_vreg;
    float* rc = reinterpret_cast<float*>(scratch);
    char* dq_scr = reinterpret_cast<char*>(scratch) + (size_t)nb * Npad * 2 * sizeof(float);
    hipLaunchKernelGGL(pam_rowconst_kernel, dim3(Npad / 256, nb), dim3(256), 0, s, lse, delta, N, Npad, rc);
    if (!deterministic) {
        if (hipMemsetAsync(dq_scr, 0, (size_t)nb * Npad * 32 * sizeof(float), s) != hipSuccess) {
            gd_set_error("gd_pam_flash_bwd: hipMemsetAsync of the dQ accumulator failed");
            return -2;
        }
    }
    const dim3 grid(Npad / 256, nb);
    const unsigned short *q = (const unsigned short*)qt, *k = (const unsigned short*)kt, *kT = (const unsigned short*)kn;
    const unsigned short *v = (const unsigned short*)vt, *dO = (const unsigned short*)dot_;
#define K64_ARGS grid, s, q, k, kT, v, dO, rc, Npad, dkn, dv, (void*)dq_scr, dk_bs, dv_bs
#define K64_CASE(CT_)                                                                                   \
    case CT_:                                                                                           \
        if (f16) {                                                                                      \
            if (deterministic) launch_k64<CT_, true, 2, false>(K64_ARGS);                               \
            else launch_k64<CT_, true, 2, true>(K64_ARGS);                                              \
        } else if (vreg == 2) {                                                                         \
            if (deterministic) launch_k64<CT_, false, 2, false>(K64_ARGS);                              \
            else launch_k64<CT_, false, 2, true>(K64_ARGS);                                             \
        } else {                                                                                        \
            if (deterministic) launch_k64<CT_, false, 1, false>(K64_ARGS);                              \
            else launch_k64<CT_, false, 1, true>(K64_ARGS);                                             \
        }                                                                                               \
        break;
    if (g_k64_order && Cp == 192 && !f16 && !deterministic) {     // schedule A/B variants (bench tooling only)
#define K64_ORD(O_)                                                                                          \
    case O_:                                                                                                 \
        if (vreg == 2) hipLaunchKernelGGL((pam_bwd_k64_kernel<6, false, 2, true, O_>), grid, dim3(256), 0, s, q, k, kT, v, dO, rc, Npad, dkn, dv, (void*)dq_scr, dk_bs, dv_bs, g_k64_dbg); \
        else hipLaunchKernelGGL((pam_bwd_k64_kernel<6, false, 1, true, O_>), grid, dim3(256), 0, s, q, k, kT, v, dO, rc, Npad, dkn, dv, (void*)dq_scr, dk_bs, dv_bs, g_k64_dbg); \
        break;
        switch (g_k64_order == 1 ? 0 : g_k64_order == 2 ? 24 : g_k64_order == 3 ? 56 : g_k64_order == 4 ? 8
                : g_k64_order == 5 ? 72 : g_k64_order == 6 ? 120 : g_k64_order == 7 ? 136 : g_k64_order == 8 ? 152
                : g_k64_order == 9 ? 264 : g_k64_order == 10 ? 328 : g_k64_order == 11 ? 520
                : g_k64_order == 14 ? 1544 : g_k64_order == 15 ? 2568 : -1) {
            K64_ORD(0)         // 1: the compiler-scheduled AGPR-form loop
            K64_ORD(8)         // 4: round-2 production (hand-placed dP phase)
            K64_ORD(24)        // 2: + unpacked dS multiplies
            K64_ORD(56)        // 3: + hand-placed second half
            K64_ORD(72)        // 5: production + segment stamps (diagnostic)
            K64_ORD(120)       // 6: hand-placed second half + segment stamps (diagnostic)
            K64_ORD(136)       // 7: production + first transpose reads hoisted above the dS arithmetic
            K64_ORD(152)       // 8: 7 + unpacked dS multiplies
            K64_ORD(264)       // 9: production + DQ16 (dQ sub-tiles over all 256 keys, no cross-wave exchange)
            K64_ORD(328)       // 10: 9 + segment stamps (diagnostic)
            // DV16 (measured 8 % SLOWER than production, profiles/r03_k64_dv16_ab.txt; the DQ16 + DV16 combination computes a
            // wrong dS for the first tile of a workgroup -- unresolved, not instantiated)
            K64_ORD(520)       // 11: production + DV16 (dV^T / dK^T on 16x16x32 MFMAs)
            K64_ORD(1544)      // 14: DV16 for dV^T only
            K64_ORD(2568)      // 15: DV16 for dK^T only
            default: gd_set_error("gd_pam_k64_variant: unknown order"); return -1;
        }
#undef K64_ORD
        hipLaunchKernelGGL(pam_dq_transpose_kernel, dim3(Npad / 64, nb), dim3(256), 0, s, (const float*)dq_scr, Npad, dqn, dq_bs);
        GD_LAUNCH_CHECK();
        return 0;
    }
    switch (Cp / 32) {
        K64_CASE(1) K64_CASE(2) K64_CASE(3) K64_CASE(4) K64_CASE(5) K64_CASE(6)
        default: gd_set_error("pam: Cp must be 32..192"); return -1;
    }
#undef K64_CASE
#undef K64_ARGS
    if (deterministic) {
        if (out_bs) { gd_set_error("gd_pam_flash_bwd: the deterministic dQ reduction writes a dense dqn (out_bs must be 0)"); return -1; }
        gd_pam_dq_reduce_launch(dq_scr, Npad / 256, Npad, nb, dqn, stream);
    }
    else
        hipLaunchKernelGGL(pam_dq_transpose_kernel, dim3(Npad / 64, nb), dim3(256), 0, s, (const float*)dq_scr, Npad, dqn, dq_bs);
    GD_LAUNCH_CHECK();
    return 0;
}
