// Microbenchmark (round 3, redo of overlap.hip): how many single-issue VALU fillers hide in the gap of a
// v_mfma_f32_32x32x16_bf16 when ONE wave per SIMD runs MFMAs and VALU in one instruction stream?
//
// overlap.hip (round 2) concluded "plain VALU and MFMA work of a SIMD add up".  Its disassembly shows why that was an
// artefact: hipcc -O3 SLP-packed the six adjacent scalar fmaf() of every gap into v_pk_fma_f32 (42 of them in the
// binary), and a packed f32 op beside an MFMA costs ~22 cycles more than the two scalar ops it replaces
// (MI355X_MICROARCH.md, "price of one filler beside MFMAs").  Here every filler is ONE inline-asm instruction, so the
// compiler can neither pack nor move it, and the kernel is sized to one wave per SIMD (256 threads, 96 KiB of LDS per
// workgroup -> one workgroup per CU), the shape of pam_bwd_k64_kernel.
//
//   filler kinds: 0 v_fma_f32   1 v_pk_fma_f32 (one per two scalar fmas' worth)   2 v_exp_f32   3 v_cvt_pk_bf16_f32
//                 4 s_nop 0     5 v_mul_f32 on INDEPENDENT registers (no chain)
//   K = fillers per MFMA gap (0..10).  Output: cycles per MFMA (s_memtime around the loop, median wave) and wall ms.
// hipcc --offload-arch=gfx950 -O3 -o overlap2 overlap2.hip && ./overlap2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KIND, int K, bool AGPR = false>
__global__ __launch_bounds__(256, 1) void kern(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ char lds_force[];          // 96 KiB dynamic: one workgroup per CU
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(((threadIdx.x * 37 + e * 11) % 97) * 0.01f - 0.4f);      // random-ish, full-range signs
        b[e] = (__bf16)(((threadIdx.x * 13 + e * 29) % 89) * 0.01f - 0.4f);
    }
    float v[12];
    f32x2 pv[6];
    for (int e = 0; e < 12; ++e) v[e] = threadIdx.x * 0.001f + e * 0.01f;
    for (int e = 0; e < 6; ++e) pv[e] = f32x2{threadIdx.x * 0.001f, e * 0.01f};
    const float c1 = 0.99999f, c2 = 1e-6f;
    unsigned int pk = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int e = 0; e < K; ++e) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e % 12]) : "v"(c1), "v"(c2));
                else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[e % 6]) : "v"(pv[(e + 3) % 6]));
                else if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[e % 12]));
                else if constexpr (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[e % 12]), "v"(v[(e + 1) % 12]));
                else if constexpr (KIND == 4) asm volatile("s_nop 0");
                else asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[e % 12]) : "v"(c1), "v"(c2));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 12; ++e) s += v[e];
    for (int e = 0; e < 6; ++e) s += pv[e].x + pv[e].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)pk + (lds_force[threadIdx.x] ? 0.f : 0.f);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int K, bool AGPR = false>
void run(float* d, unsigned long long* dc, int iters, const char* name) {
    hipFuncSetAttribute((const void*)kern<KIND, K, AGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<KIND, K, AGPR><<<256, 256, 96 * 1024>>>(d, dc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<KIND, K, AGPR><<<256, 256, 96 * 1024>>>(d, dc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), dc, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cpm = (double)h[512] / (4.0 * iters);
    printf("  %-18s K=%2d : %6.1f cycles per MFMA gap (median wave), %7.3f ms, %6.0f TF\n", name, K, cpm, ms,
           256.0 * 4 * 4.0 * iters * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12);
}

#define ROW(KIND, NAME) \
    run<KIND, 0>(d, dc, iters, NAME); run<KIND, 1>(d, dc, iters, NAME); run<KIND, 2>(d, dc, iters, NAME); run<KIND, 3>(d, dc, iters, NAME); \
    run<KIND, 4>(d, dc, iters, NAME); run<KIND, 5>(d, dc, iters, NAME); run<KIND, 6>(d, dc, iters, NAME); run<KIND, 7>(d, dc, iters, NAME); \
    run<KIND, 8>(d, dc, iters, NAME); run<KIND, 10>(d, dc, iters, NAME);

int main() {
    float* d;
    unsigned long long* dc;
    hipMalloc(&d, 256 * 256 * 4);
    hipMalloc(&dc, 1024 * 8);
    const int iters = 20000;
    printf("one wave per SIMD, 256 workgroups x 4 waves, %d x 4 MFMAs per wave, K asm fillers behind every MFMA\n", iters);
    printf("accumulators in AGPRs (the K64 backward's dV^T / dK^T form):\n");
    run<0, 0, true>(d, dc, iters, "v_fma_f32 / AGPR"); run<0, 2, true>(d, dc, iters, "v_fma_f32 / AGPR"); run<0, 4, true>(d, dc, iters, "v_fma_f32 / AGPR");
    run<0, 5, true>(d, dc, iters, "v_fma_f32 / AGPR"); run<0, 6, true>(d, dc, iters, "v_fma_f32 / AGPR"); run<0, 8, true>(d, dc, iters, "v_fma_f32 / AGPR");
    run<3, 2, true>(d, dc, iters, "v_cvt_pk / AGPR"); run<3, 4, true>(d, dc, iters, "v_cvt_pk / AGPR");
    printf("accumulators in VGPRs:\n");
    ROW(0, "v_fma_f32")
    ROW(5, "v_mul_f32 indep")
    ROW(1, "v_pk_fma_f32")
    ROW(2, "v_exp_f32")
    ROW(3, "v_cvt_pk_bf16_f32")
    ROW(4, "s_nop 0")
    return 0;
}
