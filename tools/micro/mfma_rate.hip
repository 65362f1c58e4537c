// Microbenchmark: MFMA issue rate of 1 / 2 / 4 waves per SIMD (pure MFMA stream, NACC independent accumulators per wave).
//   hipcc --offload-arch=gfx950 -O3 -w -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, int SHAPE>
__global__ void k(float* out, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f); b[e] = (__bf16)(e * 0.01f); }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int SHAPE>
void run(float* d, int threads, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)(threads / 64) / 4.0 * NACC * iters;
    const double flop = (SHAPE == 32 ? 32768.0 : 16384.0) * NACC * iters * (threads / 64) * 256;
    printf("  %dx%d  %2d acc  %d wave(s)/SIMD: %7.3f ms  %6.1f ns per MFMA and SIMD  %7.1f TFLOP/s\n", SHAPE, SHAPE, NACC, threads / 256,
           ms, ms * 1e6 / mfma_per_simd, flop / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        run<4, 32>(d, threads, iters);
        run<8, 32>(d, threads, iters);
        run<4, 16>(d, threads, iters);
        run<8, 16>(d, threads, iters);
    }
    return 0;
}
