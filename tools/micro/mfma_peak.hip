// mfma_peak.hip — what the matrix pipe of one MI355X sustains for a pure stream of fp32 / bf16 MFMAs (no memory traffic):
// the ceiling any GEMM inner loop can approach.  Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_32x32x2_f32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_16x16x4_f32(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_32x32x16_bf16(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int r = 0; r < 8; ++r) {
        a[r] = (__bf16)(a0 + threadIdx.x + r);
        b[r] = (__bf16)(b0 + r);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
double run(K kern, int blocks, int iters, float* d, double flop_per_mfma, int nacc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters / 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * (double)iters * nacc;
    return mfmas * flop_per_mfma / (ms * 1e-3) / 1e12;
}

int main() {
    float* d;
    hipMalloc(&d, 1024);
    const int iters = 20000;
    printf("| instruction | accumulators per wave | waves per SIMD | TFLOP/s |\n|---|---:|---:|---:|\n");
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        printf("| v_mfma_f32_32x32x2_f32 | 1 | %d | %.1f |\n", wps, run(k_32x32x2_f32<1>, blocks, iters, d, 4096, 1));
        printf("| v_mfma_f32_32x32x2_f32 | 2 | %d | %.1f |\n", wps, run(k_32x32x2_f32<2>, blocks, iters, d, 4096, 2));
        printf("| v_mfma_f32_32x32x2_f32 | 4 | %d | %.1f |\n", wps, run(k_32x32x2_f32<4>, blocks, iters, d, 4096, 4));
        printf("| v_mfma_f32_16x16x4_f32 | 4 | %d | %.1f |\n", wps, run(k_16x16x4_f32<4>, blocks, iters, d, 2048, 4));
        printf("| v_mfma_f32_16x16x4_f32 | 8 | %d | %.1f |\n", wps, run(k_16x16x4_f32<8>, blocks, iters, d, 2048, 8));
        printf("| v_mfma_f32_32x32x16_bf16 | 4 | %d | %.1f |\n", wps, run(k_32x32x16_bf16<4>, blocks, iters, d, 32768, 4));
        fflush(stdout);
    }
    return 0;
}
