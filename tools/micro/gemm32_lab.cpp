// gemm32_lab — A/B bench of the GEMM variants of libs3enc.so on the shapes of the HuBERT-base 32 x 10 s forward (and the
// large models' layer shapes), through the C ABI (s3enc_op_gemm + the "gemm32_big" / "gemm16_tile" tuning keys):
//   fp32 (default):  mode 0 = gemm_kernel<float> (128x128 tile), 2 / 3 / 4 / 5 = the 256 / 192 / 128 / 64 x 128 tile of gemmt.hip,
//                    1 = the library's own choice.  Every mode's output is compared BIT FOR BIT with mode 0.
//   bf16 (argv[3] = bf16): mode 0 = gemm16.hip's 256x256 lock-step tile, the others gemmt.hip; outputs are compared with
//                    mode 0 by count of values that differ (different k order: rounding-level differences are expected), and
//                    the gemmt.hip heights with each other bit for bit.
//   Timing is interleaved rounds of HIP events on random operands (cdna_hip_programming.md rules 24 / 25).
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/gemm32_lab.cpp -Iinclude -Ls3prl_amd -ls3enc -Wl,-rpath,'$ORIGIN/../../s3prl_amd' -o tools/micro/gemm32_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "s3enc.h"

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed, float scale) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((int)(x & 0xffffff) - 0x800000) * (scale / 0x800000);
    }
}
__global__ void fill16_kernel(unsigned short* p, long n, unsigned seed, float scale) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = ((int)(x & 0xffffff) - 0x800000) * (scale / 0x800000);
        p[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}
// 16-bit outputs of two kernels with different k orders: count values further apart than `tol` relative (bf16: 2^-7)
__global__ void diff16_kernel(const unsigned short* a, const unsigned short* b, long n, unsigned long long* cnt) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) {
        const float x = __uint_as_float((unsigned)a[i] << 16), y = __uint_as_float((unsigned)b[i] << 16);
        c += !(fabsf(x - y) <= 0.02f * fmaxf(fabsf(x), fabsf(y)) + 1e-2f);
    }
    if (c) atomicAdd(cnt, c);
}
__global__ void diff32_kernel(const float* a, const float* b, long n, unsigned long long* cnt) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) c += !(fabsf(a[i] - b[i]) <= 2e-3f * fmaxf(fabsf(a[i]), fabsf(b[i])) + 1e-2f);
    if (c) atomicAdd(cnt, c);
}
__global__ void diff_kernel(const unsigned* a, const unsigned* b, long n, unsigned long long* cnt) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}

struct Shape {
    const char* name;
    int batches, M, N, K;
    long lda, a_bs;  // elements; lda < K = overlapping conv rows
    int act, residual, limit;
};

int main(int argc, char** argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 5;
    const char* only = argc > 2 && strcmp(argv[2], "-") ? argv[2] : nullptr;
    const bool h16 = false;  // (round 3 measured the tile kernel for bf16 too and dropped it: profiles/r03_gemm_tile_lab.md)
    const bool x3 = argc > 3 && !strcmp(argv[3], "x3");  // S3ENC_F32X3: fp32 operands, mode 0 = gemm_x3.hip's 256x256 tile
    const char* key = h16 ? "gemm16_tile" : (x3 ? "gemm_x3_tile" : "gemm32_big");
    const int dt = h16 ? S3ENC_BF16 : (x3 ? S3ENC_F32X3 : S3ENC_F32);
    if (x3) s3enc_set_tuning("x3_pack_cache", 1);  // keep the packed weight image across the repeated calls
    const int eb = h16 ? 2 : 4;
    std::vector<Shape> shapes = {
        {"conv1 32x15999x512x1536", 32, 15999, 512, 1536, 1024, 31999L * 512, 1, 0, 0},
        {"conv2 32x7999x512x1536", 32, 7999, 512, 1536, 1024, 15999L * 512, 1, 0, 0},
        {"conv4 32x1999x512x1536", 32, 1999, 512, 1536, 1024, 3999L * 512, 1, 0, 0},
        {"conv6 32x499x512x1024", 32, 499, 512, 1024, 1024, 999L * 512, 1, 0, 0},
        {"proj 32x499x768x512", 32, 499, 768, 512, 512, 499L * 512, 0, 0, 1},
        {"qkv 15968x2304x768", 1, 15968, 2304, 768, 768, 0, 0, 0, 0},
        {"out_proj 15968x768x768", 1, 15968, 768, 768, 768, 0, 0, 1, 0},
        {"fc1 15968x3072x768", 1, 15968, 3072, 768, 768, 0, 1, 0, 0},
        {"fc2 15968x768x3072", 1, 15968, 768, 3072, 3072, 0, 0, 1, 0},
        {"L.qkv 15968x3072x1024", 1, 15968, 3072, 1024, 1024, 0, 0, 0, 0},
        {"L.out 15968x1024x1024", 1, 15968, 1024, 1024, 1024, 0, 0, 1, 0},
        {"L.fc1 15968x4096x1024", 1, 15968, 4096, 1024, 1024, 0, 1, 0, 0},
        {"L.fc2 15968x1024x4096", 1, 15968, 1024, 4096, 4096, 0, 0, 1, 0},
        {"W.qkv 23968x3072x1024", 1, 23968, 3072, 1024, 1024, 0, 0, 0, 0},
        {"W.fc2 23968x1024x4096", 1, 23968, 1024, 4096, 4096, 0, 0, 1, 0},
        {"sq8k 8192^3", 1, 8192, 8192, 8192, 8192, 0, 0, 0, 0},
        {"small 1x499 qkv 499x2304x768", 1, 499, 2304, 768, 768, 0, 0, 0, 0},
    };
    const int modes[] = {0, 2, 3, 4, 5, 1};
    const int NM = 6;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long* d_cnt;
    CK(hipMalloc(&d_cnt, 8));
    printf("%s GEMMs, TFLOP/s\n\n| shape | act/res | %s (mode 0) | 256x128 | 192x128 | 128x128 | 64x128 | auto | vs mode 0 |\n|---|---|---:|---:|---:|---:|---:|---:|---|\n",
           h16 ? "bf16" : (x3 ? "fp32x3" : "fp32"), h16 ? "gemm16.hip 256x256" : (x3 ? "gemm_x3.hip 256x256" : "gemm.hip 128x128"));
    for (const Shape& s : shapes) {
        if (only) {  // comma-separated substrings
            bool hit = false;
            std::string o(only);
            size_t a = 0;
            while (a <= o.size()) {
                size_t b = o.find(',', a);
                if (b == std::string::npos) b = o.size();
                if (b > a && strstr(s.name, o.substr(a, b - a).c_str())) hit = true;
                a = b + 1;
            }
            if (!hit) continue;
        }
        const long a_rows = s.batches > 1 ? s.a_bs * s.batches + s.K : (long)(s.M - 1) * s.lda + s.K;
        const long a_elems = a_rows + 64;
        const long w_elems = (long)s.N * s.K;
        const long o_elems = (long)s.batches * s.M * s.N;
        float *A, *W, *bias, *res, *out, *ref;
        int* lim = nullptr;
        CK(hipMalloc(&A, a_elems * eb));
        CK(hipMalloc(&W, w_elems * eb));
        CK(hipMalloc(&bias, s.N * 4));
        CK(hipMalloc(&res, o_elems * 4));
        CK(hipMalloc(&out, o_elems * 4));
        CK(hipMalloc(&ref, o_elems * 4));
        if (h16) {
            fill16_kernel<<<2048, 256, 0, st>>>((unsigned short*)A, a_elems, 1u, 1.0f);
            fill16_kernel<<<2048, 256, 0, st>>>((unsigned short*)W, w_elems, 2u, 0.05f);
        } else {
            fill_kernel<<<2048, 256, 0, st>>>(A, a_elems, 1u, 1.0f);
            fill_kernel<<<2048, 256, 0, st>>>(W, w_elems, 2u, 0.05f);
        }
        fill_kernel<<<64, 256, 0, st>>>(bias, s.N, 3u, 0.5f);
        fill_kernel<<<2048, 256, 0, st>>>(res, o_elems, 4u, 1.0f);
        if (s.limit) {
            std::vector<int> h(s.batches);
            for (int b = 0; b < s.batches; ++b) h[b] = s.M - (b * 37) % (s.M / 2);
            CK(hipMalloc(&lim, s.batches * 4));
            CK(hipMemcpy(lim, h.data(), s.batches * 4, hipMemcpyHostToDevice));
        }
        CK(hipStreamSynchronize(st));
        auto run = [&](float* o) {
            // 16-bit modes as on the path: 16-bit output unless the GEMM feeds the fp32 residual stream
            const bool o16 = h16 && !s.residual && !s.limit;
            int rc = s3enc_op_gemm(dt, A, s.lda, s.a_bs, W, bias, s.M, s.N, s.K, s.batches, s.act, s.residual ? res : nullptr,
                                   lim, o16 ? nullptr : o, o16 ? (void*)o : nullptr, s.N, (long)s.M * s.N, st);
            if (rc) {
                printf("s3enc_op_gemm failed: %s\n", s3enc_last_error());
                exit(1);
            }
        };
        double best[NM];
        std::string bit;
        const double flops = 2.0 * s.batches * s.M * (double)s.N * s.K;
        for (int m = 0; m < NM; ++m) best[m] = 1e30;
        // correctness first: every mode against mode 0, whole output, bit for bit
        s3enc_set_tuning(key, 0);
        CK(hipMemsetAsync(ref, 0xff, o_elems * 4, st));
        run(ref);
        const bool o16 = h16 && !s.residual && !s.limit;
        float* first = nullptr;  // 16-bit: the first gemmt.hip height, the others must equal it bit for bit
        if (h16) CK(hipMalloc(&first, o_elems * 4));
        for (int m = 1; m < NM; ++m) {
            s3enc_set_tuning(key, modes[m]);
            CK(hipMemsetAsync(out, 0xff, o_elems * 4, st));
            run(out);
            auto count = [&](const float* x, const float* y, int how) {
                CK(hipMemsetAsync(d_cnt, 0, 8, st));
                const long words = o16 ? o_elems / 2 : o_elems;
                if (how == 0) diff_kernel<<<2048, 256, 0, st>>>((const unsigned*)x, (const unsigned*)y, words, d_cnt);
                else if (o16) diff16_kernel<<<2048, 256, 0, st>>>((const unsigned short*)x, (const unsigned short*)y, o_elems, d_cnt);
                else diff32_kernel<<<2048, 256, 0, st>>>(x, y, o_elems, d_cnt);
                unsigned long long c = 0;
                CK(hipMemcpyAsync(&c, d_cnt, 8, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                return c;
            };
            if (!h16) {
                const unsigned long long c = count(out, ref, 0);
                bit += c ? ("mode" + std::to_string(modes[m]) + ":" + std::to_string(c) + "DIFF ") : "";
            } else {
                const unsigned long long far = count(out, ref, 1);
                if (far) bit += "mode" + std::to_string(modes[m]) + ":" + std::to_string(far) + "FAR ";
                if (m == 1) CK(hipMemcpyAsync(first, out, o_elems * 4, hipMemcpyDeviceToDevice, st));
                else {
                    const unsigned long long c = count(out, first, 0);
                    if (c) bit += "mode" + std::to_string(modes[m]) + ":" + std::to_string(c) + "!=mode2 ";
                }
            }
        }
        if (first) CK(hipFree(first));
        if (bit.empty()) bit = h16 ? "close; heights identical" : "identical";
        const int iters = std::max(2, (int)((h16 ? 2e13 : (x3 ? 8e12 : 3e12)) / flops));  // ~20 ms per measurement
        for (int r = 0; r < rounds; ++r)
            for (int m = 0; m < NM; ++m) {
                s3enc_set_tuning(key, modes[m]);
                run(out);  // warm
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) run(out);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best[m] = std::min(best[m], (double)ms / iters);
            }
        printf("| %s | %d/%d |", s.name, s.act, s.residual);
        for (int m = 0; m < NM; ++m) printf(" %.1f |", flops / best[m] * 1e-9);
        printf(" %s |\n", bit.c_str());
        fflush(stdout);
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(res)); CK(hipFree(out)); CK(hipFree(ref));
        if (lim) CK(hipFree(lim));
    }
    s3enc_set_tuning("gemm32_big", 1);
    s3enc_set_tuning("gemm_x3_tile", 1);
    return 0;
}
