// mx_probe — operand layout / scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950 block-scaled MX) and of the
// v_cvt_scalef32_pk_{fp4,fp8}_f16 converters, checked against a CPU product, plus the matrix-pipe rate of an fp16 MFMA stream
// with MX corrections mixed in.  The ISA document is not on this box; the guides give the builtin names and the C/D map only.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/mx_probe.hip -o tools/micro/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FMT>
__global__ void one_mfma(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x16* d) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, FMT, FMT, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    d[threadIdx.x] = acc;
}

__global__ void cvt_table(const h2* src, const float* scale, int n, unsigned* o4, unsigned* o8) {
    const int i = threadIdx.x;
    if (i >= n) return;
    o4[4 * i + 0] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xAAAAAAAAu, src[i], scale[i], 0);
    o4[4 * i + 1] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xAAAAAAAAu, src[i], scale[i], 1);
    o4[4 * i + 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xAAAAAAAAu, src[i], scale[i], 2);
    o4[4 * i + 3] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xAAAAAAAAu, src[i], scale[i], 3);
    s2 old = {(short)0xAAAA, (short)0xAAAA};
    s2 r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, src[i], scale[i], false);
    s2 r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, src[i], scale[i], true);
    o8[2 * i + 0] = __builtin_bit_cast(unsigned, r0);
    o8[2 * i + 1] = __builtin_bit_cast(unsigned, r1);
}

// matrix-pipe rate: per "K step of 64" and accumulator block, NH fp16 32x32x16 MFMAs + NX MX MFMAs of format FMT (K = 64 each)
template <int NH, int NX, int FMT>
__global__ __launch_bounds__(512, 2) void rate(float* out, int iters, int seed) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.25f + 0.01f * ((threadIdx.x + i + seed) & 7)); b[i] = (_Float16)(0.5f - 0.02f * ((threadIdx.x * 3 + i) & 7)); }
    v8i xa, xb;
    for (int i = 0; i < 8; ++i) { xa[i] = 0x21432143 + threadIdx.x * 0x01010101 * (i + 1); xb[i] = 0x12341234 ^ (threadIdx.x * 0x00110011 * (i + 3)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
#pragma unroll
            for (int h = 0; h < NH; ++h) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[blk], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NX; ++x) acc[blk] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, acc[blk], FMT, FMT, 0, 127, 0, 127);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static const float FP4V[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static unsigned fp4_enc(float v) {  // exact values only
    unsigned s = v < 0 ? 8u : 0u;
    v = std::fabs(v);
    for (unsigned i = 0; i < 8; ++i) if (FP4V[i] == v) return s | i;
    return 0xF;
}
static unsigned fp8_enc(float v) {  // e4m3fn, exact small values only (normal range)
    if (v == 0.f) return 0;
    unsigned s = v < 0 ? 0x80u : 0u;
    v = std::fabs(v);
    int e;
    float m = std::frexp(v, &e);  // v = m * 2^e, m in [0.5, 1)
    m *= 2.f; e -= 1;             // m in [1, 2)
    unsigned mant = (unsigned)std::lround((m - 1.f) * 8.f);
    return s | ((unsigned)(e + 7) << 3) | mant;
}

template <int FMT>
static int layout_check(const char* name) {
    // logical operands with exactly representable entries and per-(row, 32-block) scales 2^-1 .. 2^2
    std::vector<float> A(32 * 64), B(64 * 32);
    int sA[32][2], sB[32][2];
    srand(1234 + FMT);
    for (auto& v : A) v = FP4V[rand() % 8] * ((rand() & 1) ? -1.f : 1.f);
    for (auto& v : B) v = FP4V[rand() % 8] * ((rand() & 1) ? -1.f : 1.f);
    for (int i = 0; i < 32; ++i) for (int h = 0; h < 2; ++h) { sA[i][h] = 126 + rand() % 4; sB[i][h] = 126 + rand() % 4; }
    std::vector<double> ref(32 * 32, 0.0);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)A[i * 64 + k] * std::ldexp(1.0, sA[i][k / 32] - 127) * B[k * 32 + j] * std::ldexp(1.0, sB[j][k / 32] - 127);
            ref[i * 32 + j] = s;
        }
    // hypotheses: which k does element e of lane (l31, h) hold, and in which nibble / byte order
    const char* hyp[] = {"k = 32 h + e, low nibble / byte first", "k = 32 h + e, high nibble first (fp4 only)", "k = 2 e + h (interleaved halves)",
                         "k = 16 (e / 8) + 8 h + (e % 8)  [the fp16 32x32x16 fragment order]"};
    v8i *da, *db; int *dsa, *dsb; f32x16* dd;
    CK(hipMalloc(&da, 64 * sizeof(v8i))); CK(hipMalloc(&db, 64 * sizeof(v8i)));
    CK(hipMalloc(&dsa, 64 * 4)); CK(hipMalloc(&dsb, 64 * 4)); CK(hipMalloc(&dd, 64 * sizeof(f32x16)));
    printf("\n### %s\n\n| hypothesis | max abs difference to the CPU product (max |ref| = ", name);
    double mref = 0; for (double v : ref) mref = std::fmax(mref, std::fabs(v));
    printf("%.1f) |\n|---|---:|\n", mref);
    for (int hy = 0; hy < 4; ++hy) {
        std::vector<unsigned> pa(64 * 8, 0), pb(64 * 8, 0);
        std::vector<int> psa(64), psb(64);
        for (int L = 0; L < 64; ++L) {
            const int r = L & 31, h = L >> 5;
            psa[L] = sA[r][h] * 0x01010101;  // the same scale in every byte: whichever byte op_sel picks
            psb[L] = sB[r][h] * 0x01010101;
            for (int e = 0; e < 32; ++e) {
                int k = hy == 2 ? 2 * e + h : hy == 3 ? 16 * (e / 8) + 8 * h + (e % 8) : 32 * h + e;
                // (for hypotheses 2 / 3 the scale blocks no longer coincide with a lane's elements: use scales of 1 for them)
                const float av = A[r * 64 + k], bv = B[k * 32 + r];
                if (FMT == 4) {
                    const int nib = hy == 1 ? (e ^ 1) : e;
                    pa[L * 8 + nib / 8] |= fp4_enc(av) << (4 * (nib % 8));
                    pb[L * 8 + nib / 8] |= fp4_enc(bv) << (4 * (nib % 8));
                } else {
                    pa[L * 8 + e / 4] |= fp8_enc(av) << (8 * (e % 4));
                    pb[L * 8 + e / 4] |= fp8_enc(bv) << (8 * (e % 4));
                }
            }
        }
        std::vector<double> want = ref;
        if (hy >= 2) {  // unscaled reference
            for (int L = 0; L < 64; ++L) psa[L] = psb[L] = 127 * 0x01010101;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0;
                    for (int k = 0; k < 64; ++k) s += (double)A[i * 64 + k] * B[k * 32 + j];
                    want[i * 32 + j] = s;
                }
        }
        CK(hipMemcpy(da, pa.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, pb.data(), 64 * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, psa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, psb.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(one_mfma<FMT>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        std::vector<float> out(64 * 16);
        CK(hipMemcpy(out.data(), dd, 64 * 64, hipMemcpyDeviceToHost));
        double md = 0;
        for (int L = 0; L < 64; ++L)
            for (int r = 0; r < 16; ++r) {
                const int col = L & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (L >> 5);
                md = std::fmax(md, std::fabs(out[L * 16 + r] - want[row * 32 + col]));
            }
        printf("| %s | %.4g |\n", hyp[hy], md);
    }
    return 0;
}

int main() {
    printf("# mx_probe: v_mfma_scale_f32_32x32x64_f8f6f4 and v_cvt_scalef32_pk_{fp4,fp8}_f16 on this device\n");
    if (layout_check<0>("fp8 e4m3 operands (cbsz = blgp = 0)")) return 1;
    if (layout_check<4>("fp4 e2m1 operands (cbsz = blgp = 4)")) return 1;
    {   // converter semantics
        const float vals[][2] = {{1.f, 3.f}, {0.3f, 6.f}, {100.f, -0.5f}, {0.74f, 0.76f}, {5.f, 7.f}, {-1.25f, 2.5f}, {1000.f, 0.01f}};
        const float scales[] = {1.f, 2.f, 0.5f, 16.f};
        std::vector<h2> src; std::vector<float> sc;
        for (auto& v : vals) for (float s : scales) { h2 x; x[0] = (_Float16)v[0]; x[1] = (_Float16)v[1]; src.push_back(x); sc.push_back(s); }
        const int n = (int)src.size();
        h2* ds; float* dsc; unsigned *d4, *d8;
        CK(hipMalloc(&ds, n * sizeof(h2))); CK(hipMalloc(&dsc, n * 4)); CK(hipMalloc(&d4, n * 16)); CK(hipMalloc(&d8, n * 8));
        CK(hipMemcpy(ds, src.data(), n * sizeof(h2), hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_table, dim3(1), dim3(64), 0, 0, ds, dsc, n, d4, d8);
        std::vector<unsigned> o4(n * 4), o8(n * 2);
        CK(hipMemcpy(o4.data(), d4, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(o8.data(), d8, n * 8, hipMemcpyDeviceToHost));
        printf("\n### v_cvt_scalef32_pk_fp4_f16 (old = 0xAAAAAAAA, byte select 0..3) and v_cvt_scalef32_pk_fp8_f16 (old = 0xAAAAAAAA, word select 0 / 1)\n\n");
        printf("| src.x | src.y | scale | fp4 sel 0 | sel 1 | sel 2 | sel 3 | fp8 lo word | hi word |\n|---:|---:|---:|---|---|---|---|---|---|\n");
        for (int i = 0; i < n; ++i)
            printf("| %g | %g | %g | %08x | %08x | %08x | %08x | %08x | %08x |\n", (float)src[i][0], (float)src[i][1], sc[i], o4[4 * i], o4[4 * i + 1],
                   o4[4 * i + 2], o4[4 * i + 3], o8[2 * i], o8[2 * i + 1]);
    }
    {   // rate
        float* out;
        CK(hipMalloc(&out, 256 * 2 * 512 * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("\n### matrix-pipe rate, 512 workgroups x 8 waves, 8 accumulator blocks per wave (per block and 64 k: NH fp16 32x32x16 + NX MX 32x32x64)\n\n");
        printf("| stream | ms | 64-k block steps / s (x 1e9) | relative to 4 + 4 fp16 (the two-term GEMM) |\n|---|---:|---:|---:|\n");
        const int iters = 2000;
        double base = 0;
        auto run = [&](const char* name, auto kern) -> int {
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                hipLaunchKernelGGL(kern, dim3(512), dim3(512), 0, 0, out, iters, r);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kern, dim3(512), dim3(512), 0, 0, out, iters, r);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double steps = 512.0 * 8 * 8 * iters / (best * 1e-3) * 1e-9;
            if (base == 0) base = steps;
            printf("| %s | %.3f | %.2f | %.2f |\n", name, best, steps, steps / base);
            return 0;
        };
        if (run("8 fp16 (two-term weights: hi + lo)", rate<8, 0, 0>)) return 1;
        if (run("4 fp16 (one-term)", rate<4, 0, 0>)) return 1;
        if (run("4 fp16 + 1 MX fp8", rate<4, 1, 0>)) return 1;
        if (run("4 fp16 + 1 MX fp6 (e2m3)", rate<4, 1, 2>)) return 1;
        if (run("4 fp16 + 1 MX fp4", rate<4, 1, 4>)) return 1;
        if (run("1 MX fp8 only", rate<0, 1, 0>)) return 1;
        if (run("1 MX fp4 only", rate<0, 1, 4>)) return 1;
    }
    return 0;
}
