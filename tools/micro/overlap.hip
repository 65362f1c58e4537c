// Microbenchmark: do VALU instructions of one wave issue while another wave's (or the same wave's) MFMAs execute?
//   mode 0: all 8 waves MFMA only          mode 1: all 8 waves VALU (v_exp / v_fma) only
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU  mode 3: waves with even id MFMA, odd id VALU
//   mode 4: every wave interleaves 1 MFMA : K VALU in one instruction stream
// hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip && ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int TRANS>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    bool do_mfma, do_valu;
    if (MODE == 0) { do_mfma = true; do_valu = false; }
    else if (MODE == 1) { do_mfma = false; do_valu = true; }
    else if (MODE == 2) { do_mfma = wave < 4; do_valu = !do_mfma; }
    else if (MODE == 3) { do_mfma = (wave & 1) == 0; do_valu = !do_mfma; }
    else { do_mfma = do_valu = true; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f); b[e] = (__bf16)(e * 0.01f); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.01f + e;
    if (MODE == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 6; ++e) v[e] = TRANS ? __builtin_amdgcn_exp2f(v[e]) : fmaf(v[e], 1.0001f, 0.5f);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 6; ++e) v[e] = TRANS ? __builtin_amdgcn_exp2f(v[e]) : fmaf(v[e], 1.0001f, 0.5f);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int TRANS>
float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, TRANS>), dim3(blocks), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, TRANS>), dim3(blocks), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 1024 * 512 * 4);
    const int iters = 20000;
    for (int blocks : {64, 256}) {
        printf("blocks %d (one per CU), %d iterations of 4 MFMA / 24 VALU per wave\n", blocks, iters);
        printf("  fma : mfma only %.3f ms, valu only %.3f, split 0-3/4-7 %.3f, split even/odd %.3f, interleaved in every wave %.3f\n",
               run<0, 0>(d, blocks, iters), run<1, 0>(d, blocks, iters), run<2, 0>(d, blocks, iters), run<3, 0>(d, blocks, iters), run<4, 0>(d, blocks, iters));
        printf("  exp2: mfma only %.3f ms, valu only %.3f, split 0-3/4-7 %.3f, split even/odd %.3f, interleaved in every wave %.3f\n",
               run<0, 1>(d, blocks, iters), run<1, 1>(d, blocks, iters), run<2, 1>(d, blocks, iters), run<3, 1>(d, blocks, iters), run<4, 1>(d, blocks, iters));
    }
    return 0;
}
