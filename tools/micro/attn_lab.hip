// attn_lab — timing probes of the 16-bit attention kernel (attention.hip compiled here with S3_ATTN_PROBE): which part of
// a tile bounds it?  Results are garbage for probe != 0; probe 0 is the product kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DS3_ATTN_PROBE -Is3prl_amd/csrc tools/micro/attn_lab.hip -o tools/micro/attn_lab
#include "../../s3prl_amd/csrc/attention.hip"

#include <cstdio>
#include <vector>

namespace s3 {
Tuning g_tuning;
thread_local const Tuning* t_tuning = nullptr;
}  // namespace s3

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill16(unsigned short* p, long n, unsigned seed, float scale) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = ((int)(x & 0xffffff) - 0x800000) * (scale / 0x800000);
        p[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}

int main() {
    struct Shape { const char* name; int B, T, H; };
    const Shape shapes[] = {{"HuBERT-base 32 x 499 x 12 heads", 32, 499, 12}, {"HuBERT-large 32 x 499 x 16", 32, 499, 16},
                            {"WavLM-large 32 x 749 x 16 (no bias)", 32, 749, 16}};
    const int probes[] = {0, 1, 2, 4, 8, 16, 1 | 16, 2 | 4 | 8, 1 | 2 | 4 | 8 | 16};
    const char* names[] = {"product", "no staging after tile 0", "no softmax", "no P.V", "no Q.K", "no barrier", "no staging, no barrier",
                           "no softmax / MFMA", "empty loop"};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const long D = 64L * sh.H, rows = (long)sh.B * sh.T;
        unsigned short *qkv, *out;
        int* valid;
        CK(hipMalloc(&qkv, rows * 3 * D * 2));
        CK(hipMalloc(&out, rows * D * 2));
        CK(hipMalloc(&valid, sh.B * 4));
        std::vector<int> v(sh.B, sh.T);
        CK(hipMemcpy(valid, v.data(), sh.B * 4, hipMemcpyHostToDevice));
        fill16<<<2048, 256, 0, st>>>(qkv, rows * 3 * D, 7u, 1.0f);
        const double flops = 4.0 * sh.B * sh.H * (double)sh.T * sh.T * 64;
        printf("\n%s  (bf16, %.1f GFLOP)\n\n| probe | us per launch | TFLOP/s |\n|---|---:|---:|\n", sh.name, flops * 1e-9);
        for (size_t i = 0; i < sizeof(probes) / sizeof(probes[0]); ++i) {
            s3::AttnParams p{};
            p.qkv = qkv; p.out = out; p.valid = valid; p.B = sh.B; p.T = sh.T; p.H = sh.H; p.probe = probes[i];
            double best = 1e30;
            for (int r = 0; r < 3; ++r) {
                CK(s3::launch_attention(s3::BF16, p, st));
                CK(hipEventRecord(e0, st));
                for (int k = 0; k < 40; ++k) CK(s3::launch_attention(s3::BF16, p, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms / 40 < best ? ms / 40 : best;
            }
            printf("| %s | %.1f | %.0f |\n", names[i], best * 1e3, flops / best * 1e-9);
            fflush(stdout);
        }
        CK(hipFree(qkv)); CK(hipFree(out)); CK(hipFree(valid));
    }
    return 0;
}
