// Practical MFMA ceiling probe: every wave issues independent v_mfma_f32_32x32x16_bf16 back to back on
// register-resident random operands (no memory traffic), NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
template <int NACC>
__global__ __launch_bounds__(512) void probe(float* out, int iters, unsigned seed) {
    f16v acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    bf8 A, B;
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u; A[j] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
        s = s * 1664525u + 1013904223u; B[j] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
    float t = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) t += acc[a][e];
    if (t == 12345.678f) out[0] = t;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((probe<4>), dim3(256 * 4), dim3(waves * 64), 0, 0, out, iters, 1234u + rep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 256.0 * 4 * waves * iters * 4 * 32.0 * 32 * 16 * 2;
            printf("waves/WG %d  %.3f ms  %.1f TFLOP/s\n", waves, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
