// gemm16_lab — timing probes of the 16-bit lock-step GEMM (gemm16.hip compiled here with S3_GEMM_PROBE): how much of a launch is
// the K loop, the epilogue's LDS / VALU part, and the global stores (+ their drain before the workgroup can retire)?
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DS3_GEMM_PROBE -Is3prl_amd/csrc tools/micro/gemm16_lab.hip -o tools/micro/gemm16_lab
#include "../../s3prl_amd/csrc/gemm16.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace s3 {
Tuning g_tuning;
thread_local const Tuning* t_tuning = nullptr;
int device_cus() { return 256; }
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

// `gemm16_lab cmp a b ...`: products only, one column per gemm16_big mode given (default: 3 = lock-step tile per workgroup, 7 = the
// persistent tile loop), interleaved rounds so that clock drift hits every column alike
static int compare(int argc, char** argv) {
    struct Shape { const char* name; int M, N, K, act, res; };
    const Shape shapes[] = {{"qkv 15968x2304x768", 15968, 2304, 768, 0, 0}, {"out_proj 15968x768x768", 15968, 768, 768, 0, 1},
                            {"fc1 15968x3072x768", 15968, 3072, 768, 1, 0}, {"fc2 15968x768x3072", 15968, 768, 3072, 0, 1},
                            {"conv2 255968x512x1536", 255968, 512, 1536, 1, 0}, {"L.qkv 15968x3072x1024", 15968, 3072, 1024, 0, 0},
                            {"L.fc1 15968x4096x1024", 15968, 4096, 1024, 1, 0}, {"L.fc2 15968x1024x4096", 15968, 1024, 4096, 0, 1}};
    // `cmp8 ...`: every tile loads tile 0's panels (variant bit 3: all operands L2-resident — what each shape's launch costs when
    // the memory side is free);  `cmpx ...`: fp16 operands with two-term weights (S3ENC_F16X2: W rows [hi | lo], 2K contraction)
    const bool shared = !strcmp(argv[1], "cmp8");
    const bool split = !strcmp(argv[1], "cmpx");
    int modes[8], nm = 0;
    for (int i = 2; i < argc && nm < 8; ++i) modes[nm++] = atoi(argv[i]);
    if (!nm) { modes[0] = 3; modes[1] = 7; nm = 2; }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("| shape (%s%s) |", split ? "fp16x2" : "bf16", shared ? ", every tile reads tile 0's panels" : "");
    for (int c = 0; c < nm; ++c) printf(" gemm16_big = %d, us | TFLOP/s |", modes[c]);
    printf("\n|---|");
    for (int c = 0; c < nm; ++c) printf("---:|---:|");
    printf("\n");
    for (const Shape& sh : shapes) {
        unsigned short *A, *W, *o16;
        float *bias, *res, *o32;
        const long mn = (long)sh.M * sh.N;
        CK(hipMalloc(&A, (long)sh.M * sh.K * 2 + 256));
        CK(hipMalloc(&W, (long)sh.N * sh.K * 2 * (split ? 2 : 1)));
        CK(hipMalloc(&o16, mn * 2));
        CK(hipMalloc(&o32, mn * 4));
        CK(hipMalloc(&res, mn * 4));
        CK(hipMalloc(&bias, sh.N * 4));
        unsigned char *w4 = nullptr, *w4s = nullptr;  // cmpx, columns >= 1000: the lo term as an MX-fp4 image (timing: random nibbles, scale 2^-12)
        if (split) {
            CK(hipMalloc(&w4, (long)sh.N * sh.K / 2 + 256));
            CK(hipMalloc(&w4s, (long)sh.N * sh.K / 32 + 256));
            fill16<<<2048, 256, 0, st>>>((unsigned short*)w4, (long)sh.N * sh.K / 4, 3u, 1.0f);
            CK(hipMemsetAsync(w4s, 115, (long)sh.N * sh.K / 32, st));
        }
        fill16<<<2048, 256, 0, st>>>(A, (long)sh.M * sh.K, 1u, 1.0f);
        fill16<<<2048, 256, 0, st>>>(W, (long)sh.N * sh.K * (split ? 2 : 1), 2u, 0.05f);
        CK(hipMemsetAsync(bias, 0, sh.N * 4, st));
        CK(hipMemsetAsync(res, 0, mn * 4, st));
        s3::GemmParams p{};
        p.A = A; p.lda = sh.K; p.a_bs = 0; p.W = W; p.bias = bias; p.M = sh.M; p.N = sh.N; p.K = sh.K; p.batches = 1;
        p.act = sh.act; p.residual = sh.res ? res : nullptr; p.row_limit = nullptr;
        p.out32 = sh.res ? o32 : nullptr; p.out16 = sh.res ? nullptr : (void*)o16; p.ldo = sh.N; p.o_bs = mn;
        p.variant = 3 | (shared ? 8 : 0);
        p.wsplit = split ? 1 : 0;
        double best[8];
        for (int c = 0; c < nm; ++c) best[c] = 1e30;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < nm; ++c) {
                s3::g_tuning.gemm16_big = modes[c] % 100;  // (column 107 = mode 7 with gemm16_pp = 1, 207: pp = 2, 307: pp = 3)
                s3::g_tuning.gemm16_pp = (modes[c] / 100) % 10;
                // column s0007 (s = 1 .. 25): mode 7 with every second workgroup of an XCD started 2 s microseconds late (round 6)
                p.variant = 3 | (shared ? 8 : 0) | (((modes[c] / 10000) & 0xff) << 8);
                const bool mx = split && modes[c] >= 1000 && !(sh.K & 127);  // column 1007: cmpx with the MX second term (gemm16.hip MXW)
                p.wsplit = split && !mx ? 1 : 0;
                p.ldw = split ? 2L * sh.K : 0;
                p.mxw = mx ? 1 : 0;
                p.W4 = w4;
                p.W4s = w4s;
                const int dt = split ? s3::F16 : s3::BF16;
                CK(s3::launch_gemm16_big(dt, p, st));
                CK(hipEventRecord(e0, st));
                for (int k = 0; k < 30; ++k) CK(s3::launch_gemm16_big(dt, p, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r && ms / 30 < best[c]) best[c] = ms / 30;   // round 0 warms the clocks
            }
        printf("| %s |", sh.name);
        for (int c = 0; c < nm; ++c) printf(" %.1f | %.0f |", best[c] * 1e3, 2.0 * sh.M * (double)sh.N * sh.K / best[c] * 1e-9);  // algorithmic FLOPs (fp16x2 runs twice the MFMAs)
        printf("\n");
        fflush(stdout);
        if (w4) { CK(hipFree(w4)); CK(hipFree(w4s)); }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(o16)); CK(hipFree(o32)); CK(hipFree(res)); CK(hipFree(bias));
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strncmp(argv[1], "cmp", 3)) return compare(argc, argv);
    // the probes live in the one-tile-per-workgroup kernels: 1 = 256x256 / 192x256 (the tiles the default, persistent mode 7 walks),
    // 4 = 128x256 ring, two per CU; under 3 / 7 the probe bits are ignored and every row would time the full product
    s3::g_tuning.gemm16_big = argc > 1 ? atoi(argv[1]) : 1;
    if (s3::g_tuning.gemm16_big == 3 || s3::g_tuning.gemm16_big == 7) {
        printf("the probes need a one-tile-per-workgroup mode (1, 2, 4, 5, 6); use `cmp` for the persistent loop\n");
        return 1;
    }
    printf("gemm16_big mode %d\n", s3::g_tuning.gemm16_big);
    struct Shape { const char* name; int M, N, K, act, res; };
    const Shape shapes[] = {{"qkv 15968x2304x768", 15968, 2304, 768, 0, 0}, {"out_proj 15968x768x768", 15968, 768, 768, 0, 1},
                            {"fc1 15968x3072x768", 15968, 3072, 768, 1, 0}, {"fc2 15968x768x3072", 15968, 768, 3072, 0, 1},
                            {"L.fc1 15968x4096x1024", 15968, 4096, 1024, 1, 0}};
    const int probes[] = {0, 16, 32, 64, 64 | 16};
    const char* names[] = {"product", "no global stores", "no epilogue", "no K loop", "no K loop, no stores"};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        unsigned short *A, *W, *o16;
        float *bias, *res, *o32;
        const long mn = (long)sh.M * sh.N;
        CK(hipMalloc(&A, (long)sh.M * sh.K * 2 + 256));
        CK(hipMalloc(&W, (long)sh.N * sh.K * 2));
        CK(hipMalloc(&o16, mn * 2));
        CK(hipMalloc(&o32, mn * 4));
        CK(hipMalloc(&res, mn * 4));
        CK(hipMalloc(&bias, sh.N * 4));
        fill16<<<2048, 256, 0, st>>>(A, (long)sh.M * sh.K, 1u, 1.0f);
        fill16<<<2048, 256, 0, st>>>(W, (long)sh.N * sh.K, 2u, 0.05f);
        CK(hipMemsetAsync(bias, 0, sh.N * 4, st));
        CK(hipMemsetAsync(res, 0, mn * 4, st));
        const double flops = 2.0 * sh.M * (double)sh.N * sh.K;
        printf("\n%s  bf16 (act %d, residual %d)\n\n| probe | us per launch | TFLOP/s |\n|---|---:|---:|\n", sh.name, sh.act, sh.res);
        for (size_t i = 0; i < sizeof(probes) / sizeof(probes[0]); ++i) {
            s3::GemmParams p{};
            p.A = A; p.lda = sh.K; p.a_bs = 0; p.W = W; p.bias = bias; p.M = sh.M; p.N = sh.N; p.K = sh.K; p.batches = 1;
            p.act = sh.act; p.residual = sh.res ? res : nullptr; p.row_limit = nullptr;
            p.out32 = sh.res ? o32 : nullptr; p.out16 = sh.res ? nullptr : (void*)o16; p.ldo = sh.N; p.o_bs = mn;
            p.variant = 3 | probes[i];
            double best = 1e30;
            for (int r = 0; r < 3; ++r) {
                CK(s3::launch_gemm16_big(s3::BF16, p, st));
                CK(hipEventRecord(e0, st));
                for (int k = 0; k < 40; ++k) CK(s3::launch_gemm16_big(s3::BF16, p, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms / 40 < best ? ms / 40 : best;
            }
            printf("| %s | %.1f | %.0f |\n", names[i], best * 1e3, flops / best * 1e-9);
            fflush(stdout);
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(o16)); CK(hipFree(o32)); CK(hipFree(res)); CK(hipFree(bias));
    }
    return 0;
}
