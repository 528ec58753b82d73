// attn_lab — timing probes of the 16-bit attention kernel (attention.hip compiled here with S3_ATTN_PROBE): which part of
// a tile bounds it?  Results are garbage for probe != 0; probe 0 is the product kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DS3_ATTN_PROBE -Is3prl_amd/csrc tools/micro/attn_lab.hip -o tools/micro/attn_lab
#include "../../s3prl_amd/csrc/attention.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

namespace s3 {
Tuning g_tuning;
thread_local const Tuning* t_tuning = nullptr;
}  // namespace s3


// fp64 evaluation of sampled (batch, head, query) rows from the 16-bit operands the kernel read (its operand contract: q arrives
// pre-scaled, scores are base-2 logarithms) — every S3_ATTN_EXP build is held against this, not against another build
static float h16_to_f(unsigned short h, bool bf) {
    if (bf) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? ldexpf((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(m | 1024), e - 25);
    return s ? -v : v;
}
static void check_rows(const char* tag, const std::vector<unsigned short>& qkv, const std::vector<unsigned short>& out, const std::vector<int>& valid,
                       int B, int T, int H, bool bf, const std::vector<float>* table, const std::vector<float>* gate, int R) {
    const long D = 64L * H;
    double worst = 0, sum2 = 0, ref2 = 0;
    long n = 0;
    for (int sidx = 0; sidx < 192; ++sidx) {
        const int b = (sidx * 7) % B, h = (sidx * 5) % H;
        const int q = sidx < 8 ? (sidx & 1 ? T - 1 - sidx : sidx) : (int)((sidx * 2654435761u) % (unsigned)T);
        const int nv = valid[b];
        if (nv == 0) continue;
        std::vector<double> sc(nv);
        double mx = -1e300;
        for (int j = 0; j < nv; ++j) {
            double a = 0;
            for (int d = 0; d < 64; ++d)
                a += (double)h16_to_f(qkv[((long)b * T + q) * 3 * D + h * 64 + d], bf) * (double)h16_to_f(qkv[((long)b * T + j) * 3 * D + D + h * 64 + d], bf);
            if (table) {
                int rel = j - q; rel = rel < -R ? -R : rel > R ? R : rel;
                a += (double)(*gate)[((long)b * H + h) * T + q] * 1.44269504088896340736 * (double)(*table)[(long)h * (2 * R + 1) + R + rel];
            }
            sc[j] = a;
            mx = a > mx ? a : mx;
        }
        double l = 0;
        std::vector<double> o(64, 0.0);
        for (int j = 0; j < nv; ++j) {
            const double pj = exp2(sc[j] - mx);
            l += pj;
            for (int d = 0; d < 64; ++d) o[d] += pj * (double)h16_to_f(qkv[((long)b * T + j) * 3 * D + 2 * D + h * 64 + d], bf);
        }
        for (int d = 0; d < 64; ++d) {
            const double ref = o[d] / l, got = h16_to_f(out[((long)b * T + q) * D + h * 64 + d], bf);
            const double e = fabs(got - ref);
            worst = e > worst ? e : worst;
            sum2 += e * e; ref2 += ref * ref; ++n;
        }
    }
    printf("%s: %ld sampled outputs vs fp64: max abs err %.3e, rel Frobenius %.3e\n", tag, n, worst, sqrt(sum2 / (ref2 > 0 ? ref2 : 1)));
}

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

__global__ void fill32(float* p, long n, unsigned seed, float scale) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((int)(x & 0xffffff) - 0x800000) * (scale / 0x800000);
    }
}

// the fp32 kernel (the headline mode's attention): one-shot grid against the persistent form, bit-compared on a ragged batch
static int lab_f32(hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    struct Shape { const char* name; int B, T, H; bool bias; };
    const Shape shapes[] = {{"HuBERT-base 32 x 499 x 12 heads", 32, 499, 12, false}, {"HuBERT-large 32 x 499 x 16", 32, 499, 16, false},
                            {"WavLM-large 32 x 749 x 16, gated relative-position bias (R = 800)", 32, 749, 16, true}};
    printf("\n## fp32 kernel (attn_f32_kernel vs attn_f32p_kernel)\n\n| shape | GFLOP | one-shot grid: us | TFLOP/s | persistent: us | TFLOP/s | elements that differ (ragged batch) |\n|---|---:|---:|---:|---:|---:|---:|\n");
    for (const Shape& sh : shapes) {
        const long D = 64L * sh.H, rows = (long)sh.B * sh.T;
        const int R = 800;
        float *qkv, *out, *out2, *table = nullptr, *gate = nullptr;
        int* valid;
        CK(hipMalloc(&qkv, rows * 3 * D * 4));
        CK(hipMalloc(&out, rows * D * 4));
        CK(hipMalloc(&out2, rows * D * 4));
        CK(hipMalloc(&valid, sh.B * 4));
        std::vector<int> v(sh.B, sh.T);
        for (int b = 1; b < sh.B; b += 3) v[b] = sh.T - (b * 37) % (sh.T - 1);
        CK(hipMemcpy(valid, v.data(), sh.B * 4, hipMemcpyHostToDevice));
        fill32<<<2048, 256, 0, st>>>(qkv, rows * 3 * D, 11u, 1.0f);
        if (sh.bias) {
            std::vector<float> tb((size_t)sh.H * (2 * R + 1)), gt((size_t)sh.B * sh.H * sh.T);
            for (size_t i = 0; i < tb.size(); ++i) tb[i] = 0.01f * (float)((int)(i * 2654435761u % 401) - 200);
            for (size_t i = 0; i < gt.size(); ++i) gt[i] = 0.5f + 0.001f * (float)(i * 40503u % 997);
            CK(hipMalloc(&table, tb.size() * 4));
            CK(hipMalloc(&gate, gt.size() * 4));
            CK(hipMemcpy(table, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(gate, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
        }
        s3::AttnParams p{};
        p.qkv = qkv; p.valid = valid; p.B = sh.B; p.T = sh.T; p.H = sh.H; p.bias_table = table; p.table_R = R; p.gate = gate;
        std::vector<unsigned> h0(rows * D), h1(rows * D);
        s3::g_tuning.attn_persist = 0; p.out = out;
        CK(hipMemsetAsync(out, 0xff, rows * D * 4, st));
        CK(s3::launch_attention(s3::F32, p, st));
        s3::g_tuning.attn_persist = 1; p.out = out2;
        CK(hipMemsetAsync(out2, 0xff, rows * D * 4, st));
        CK(s3::launch_attention(s3::F32, p, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h0.data(), out, rows * D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), out2, rows * D * 4, hipMemcpyDeviceToHost));
        long diff = 0;
        for (long i = 0; i < rows * D; ++i) diff += h0[i] != h1[i];
        for (int i = 0; i < sh.B; ++i) v[i] = sh.T;
        CK(hipMemcpy(valid, v.data(), sh.B * 4, hipMemcpyHostToDevice));
        const double flops = 4.0 * sh.B * sh.H * (double)sh.T * sh.T * 64;
        double res[2];
        p.out = out;
        for (int persist = 0; persist < 2; ++persist) {
            s3::g_tuning.attn_persist = persist;
            double best = 1e30;
            for (int r = 0; r < 3; ++r) {
                CK(s3::launch_attention(s3::F32, p, st));
                CK(hipEventRecord(e0, st));
                for (int k = 0; k < 20; ++k) CK(s3::launch_attention(s3::F32, p, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms / 20 < best ? ms / 20 : best;
            }
            res[persist] = best;
        }
        printf("| %s | %.1f | %.1f | %.0f | %.1f | %.0f | %ld of %ld |\n", sh.name, flops * 1e-9, res[0] * 1e3, flops / res[0] * 1e-9, res[1] * 1e3,
               flops / res[1] * 1e-9, diff, rows * D);
        fflush(stdout);
        CK(hipFree(qkv)); CK(hipFree(out)); CK(hipFree(out2)); CK(hipFree(valid));
        if (table) CK(hipFree(table));
        if (gate) CK(hipFree(gate));
    }
    return 0;
}

int main() {
    struct Shape { const char* name; int B, T, H; bool bias; };
    const Shape shapes[] = {{"HuBERT-base 32 x 499 x 12 heads", 32, 499, 12, false}, {"HuBERT-large 32 x 499 x 16", 32, 499, 16, false},
                            {"WavLM-large 32 x 749 x 16 (no bias)", 32, 749, 16, false},
                            {"WavLM-large 32 x 749 x 16, gated relative-position bias (R = 800)", 32, 749, 16, true}};
    const bool quick = getenv("QUICK") != nullptr;  // the product rows only (the S3_ATTN_EXP builds)
    const int probes[] = {0, 1, 2, 4, 8, 16, 1 | 16, 2 | 4 | 8, 1 | 2 | 4 | 8 | 16};
    const char* names[] = {"product", "no staging after tile 0", "no softmax", "no P.V", "no Q.K", "no barrier", "no staging, no barrier",
                           "no softmax / MFMA", "empty loop"};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    if (!getenv("SKIP_F32") && lab_f32(st, e0, e1)) return 1;
    printf("\n## 16-bit kernel (attn_h16_kernel vs attn_h16p_kernel), with the timing probes\n\n");
    for (const Shape& sh : shapes) {
        const long D = 64L * sh.H, rows = (long)sh.B * sh.T;
        const int R = 800;
        unsigned short *qkv, *out, *out2;
        int* valid;
        float *table = nullptr, *gate = nullptr;
        CK(hipMalloc(&qkv, rows * 3 * D * 2));
        CK(hipMalloc(&out, rows * D * 2));
        CK(hipMalloc(&out2, rows * D * 2));
        CK(hipMalloc(&valid, sh.B * 4));
        std::vector<int> v(sh.B, sh.T);
        for (int b = 1; b < sh.B; b += 3) v[b] = sh.T - (b * 37) % (sh.T - 1);  // ragged: a third of the batch is padded
        CK(hipMemcpy(valid, v.data(), sh.B * 4, hipMemcpyHostToDevice));
        fill16<<<2048, 256, 0, st>>>(qkv, rows * 3 * D, 7u, 1.0f);
        std::vector<float> tb, gt;
        if (sh.bias) {
            tb.resize((size_t)sh.H * (2 * R + 1)); gt.resize((size_t)sh.B * sh.H * sh.T);
            for (size_t i = 0; i < tb.size(); ++i) tb[i] = 0.01f * (float)((int)(i * 2654435761u % 401) - 200);
            for (size_t i = 0; i < gt.size(); ++i) gt[i] = 0.5f + 0.001f * (float)(i * 40503u % 997);
            CK(hipMalloc(&table, tb.size() * 4));
            CK(hipMalloc(&gate, gt.size() * 4));
            CK(hipMemcpy(table, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(gate, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
        }
        const double flops = 4.0 * sh.B * sh.H * (double)sh.T * sh.T * 64;
        CK(hipStreamSynchronize(st));
        std::vector<unsigned short> hq(rows * 3 * D);
        CK(hipMemcpy(hq.data(), qkv, rows * 3 * D * 2, hipMemcpyDeviceToHost));
        // the persistent form must reproduce the one-shot grid bit for bit (ragged batch: the padded keys are masked in both)
        {
            s3::AttnParams p{};
            p.qkv = qkv; p.valid = valid; p.B = sh.B; p.T = sh.T; p.H = sh.H; p.bias_table = table; p.table_R = R; p.gate = gate;
            std::vector<unsigned short> h0(rows * D), h1(rows * D);
            for (int dt : {(int)s3::BF16, (int)s3::F16}) {
                s3::g_tuning.attn_persist = 0; p.out = out;
                CK(hipMemsetAsync(out, 0xff, rows * D * 2, st));
                CK(s3::launch_attention(dt, p, st));
                s3::g_tuning.attn_persist = 1; p.out = out2;
                CK(hipMemsetAsync(out2, 0xff, rows * D * 2, st));
                CK(s3::launch_attention(dt, p, st));
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(h0.data(), out, rows * D * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h1.data(), out2, rows * D * 2, hipMemcpyDeviceToHost));
                long diff = 0;
                for (long i = 0; i < rows * D; ++i) diff += h0[i] != h1[i];
                printf("%s  %s: persistent vs one-shot grid, %ld of %ld output elements differ\n", sh.name, dt == s3::BF16 ? "bf16" : "f16", diff, rows * D);
                check_rows(dt == s3::BF16 ? "  one-shot bf16" : "  one-shot f16", hq, h0, v, sh.B, sh.T, sh.H, dt == s3::BF16, sh.bias ? &tb : nullptr, sh.bias ? &gt : nullptr, R);
                check_rows(dt == s3::BF16 ? "  persistent bf16" : "  persistent f16", hq, h1, v, sh.B, sh.T, sh.H, dt == s3::BF16, sh.bias ? &tb : nullptr, sh.bias ? &gt : nullptr, R);
            }
        }
        for (int i = 0; i < sh.B; ++i) v[i] = sh.T;
        CK(hipMemcpy(valid, v.data(), sh.B * 4, hipMemcpyHostToDevice));
        printf("\n%s  (bf16, %.1f GFLOP, equal lengths)\n\n| probe | one-shot grid: us | TFLOP/s | persistent: us | TFLOP/s |\n|---|---:|---:|---:|---:|\n", sh.name, flops * 1e-9);
        for (size_t i = 0; i < (quick ? 1 : sizeof(probes) / sizeof(probes[0])); ++i) {
            s3::AttnParams p{};
            p.qkv = qkv; p.out = out; p.valid = valid; p.B = sh.B; p.T = sh.T; p.H = sh.H; p.probe = probes[i];
            p.bias_table = table; p.table_R = R; p.gate = gate;
            double res[2];
            for (int persist = 0; persist < 2; ++persist) {
                s3::g_tuning.attn_persist = persist;
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
                res[persist] = best;
            }
            printf("| %s | %.1f | %.0f | %.1f | %.0f |\n", names[i], res[0] * 1e3, flops / res[0] * 1e-9, res[1] * 1e3, flops / res[1] * 1e-9);
            fflush(stdout);
        }
        CK(hipFree(qkv)); CK(hipFree(out)); CK(hipFree(out2)); CK(hipFree(valid));
        if (table) CK(hipFree(table));
        if (gate) CK(hipFree(gate));
    }
    return 0;
}
