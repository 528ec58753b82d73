// mx_gemm_lab — GEMM-level prototype of "the second weight term on the MX pipe" (profiles/r04_mx_probe.md), NOT part of libs3enc:
//   C = A (fp16) x (W_hi (fp16) + W_lo)^T,   W_lo = W - fp16(W)
// with W_lo either a second fp16 term (what S3ENC_F16X2 does today: the K loop runs over 2K, A staged twice) or an MX-fp4 image
// (per 32 k one E8M0 scale + 32 e2m1 nibbles) multiplied against an MX-fp4 image of A by ONE v_mfma_scale_f32_32x32x64_f8f6f4
// per 64 k and accumulator block, in the same K step as the four fp16 MFMAs.  Same tile, wave layout, LDS-DMA staging and
// swizzle as gemm16.hip's 192 x 256 kernel (one tile per workgroup, 2 stages of 64 k); a plain fp32 epilogue (this is a K-loop
// experiment).  Checked against a float64 product on the host; timed on the path's shapes.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/mx_gemm_lab.hip -o tools/micro/mx_gemm_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int WTM = 96, BM = 2 * WTM, BN = 256, NTHR = 512, ROWB = 128, MI = WTM / 32;
constexpr int A16 = BM * ROWB, W16 = BN * ROWB, A4B = BM * 32, W4B = BN * 32;
constexpr int NLA = BM / 64, NLB = BN / 64, NL = NLA + NLB, PASS = NTHR * 16;

struct P {
    const char* A;      // (M, K) fp16
    const char* W;      // two-term: (N, 2K) fp16 rows [hi | lo];  MX: (N, K) fp16 hi
    const char* A4;     // MX: (M, K/32, 16 bytes) e2m1 nibbles, element e of a block in nibble e
    const uint8_t* A4s; // MX: (M, K/32) E8M0
    const char* W4;
    const uint8_t* W4s;
    float* C;           // (M, N) fp32
    int M, N, K, store;
};

// MODE 0: two fp16 terms; 1: MX second term, A image staged from memory (a producer wrote it); 2: MX second term, the A image is
// built IN the kernel from the fp16 fragments the lane already holds (block max, scale, 16 converts per 32 values) — no producer,
// no A-side staging.  The fragment of step q of lane (row, half) is the 16-byte slot 4 half + q of the row's 128-byte K step, i.e.
// k = 32 half + 8 q + j: the lane's 32 values over the four steps ARE block `half` in natural order, so W_lo needs no re-packing.
template <int MODE>
__global__ __launch_bounds__(NTHR, 2) void gemm(P p) {
    constexpr bool MX = MODE != 0, CONV = MODE == 2;
    constexpr int STAGE = A16 + W16 + (MX ? (CONV ? 0 : A4B) + W4B : 0);
    constexpr int OFF_W4 = A16 + W16 + (CONV ? 0 : A4B);
    constexpr int SCB = (BM + BN) * 4;  // scale bytes per buffer: 4 blocks (= 2 K steps) per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3, half = lane >> 5, l31 = lane & 31;
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / n_tiles) * BM, n0 = (blockIdx.x % n_tiles) * BN;
    const long kbytes = (long)p.K * 2, wk = MX ? kbytes : 2 * kbytes;
    const int nk = (int)(wk / ROWB);
    const int kblocks = p.K / 32;

    const int ps = tid & 7, lr = tid >> 3, ls = ps ^ ((lr >> 1) & 7);
    const char *a_ptr[NLA], *w_ptr[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        int r = m0 + lr + 64 * i;
        r = r < p.M ? r : p.M - 1;
        a_ptr[i] = p.A + (long)r * kbytes + ls * 16;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        int r = n0 + lr + 64 * i;
        r = r < p.N ? r : p.N - 1;
        w_ptr[i] = p.W + (long)r * wk + ls * 16;
    }
    // MX tiles: lane t fills 16 bytes (one block) of row t >> 1; scales: lane t one dword (4 blocks) of operand row t
    const char *a4_ptr = nullptr, *w4_ptr = nullptr, *sc_ptr = nullptr;
    if (MX) {
        int ra = m0 + (tid >> 1);
        ra = ra < p.M ? ra : p.M - 1;
        a4_ptr = p.A4 + ((long)ra * kblocks + (tid & 1)) * 16;
        int rw = n0 + (tid >> 1);
        rw = rw < p.N ? rw : p.N - 1;
        w4_ptr = p.W4 + ((long)rw * kblocks + (tid & 1)) * 16;
        if (tid < BM) {
            int r = m0 + tid;
            r = r < p.M ? r : p.M - 1;
            sc_ptr = (const char*)p.A4s + (long)r * kblocks;
        } else {
            int r = n0 + tid - BM;
            r = r < p.N ? r : p.N - 1;
            sc_ptr = (const char*)p.W4s + (long)r * kblocks;
        }
    }
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    const unsigned lds_sc = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 2 * STAGE + wave * 256);
    auto dma16 = [&](const char* g, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    };
    auto dma4 = [&](const char* g, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    };
    auto issue = [&](int kt, int stage) {
        const long kb = (long)kt * ROWB;
        const long kba = kb >= kbytes ? kb - kbytes : kb;  // (two-term: the lo half re-reads A from its start)
        const unsigned sa = lds_base + stage * STAGE;
#pragma unroll
        for (int pc = 0; pc < NLA; ++pc) dma16(a_ptr[pc] + kba, sa + pc * PASS);
#pragma unroll
        for (int pc = 0; pc < NLB; ++pc) dma16(w_ptr[pc] + kb, sa + A16 + pc * PASS);
        if (MX) {
            if (!CONV && wave < BM / 32) dma16(a4_ptr + (long)kt * 32, sa + A16 + W16);  // 2 lanes per row: BM rows = BM / 32 waves
            dma16(w4_ptr + (long)kt * 32, sa + OFF_W4);
            if (!(kt & 1) && wave < (BM + BN) / 64) dma4(sc_ptr + 2 * kt, lds_sc + ((kt >> 1) & 1) * SCB);  // blocks 2 kt .. 2 kt + 3
        }
    };
    auto barrier_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    const int swz = (l31 >> 1) & 7;
    const int a_row0 = (wr * WTM + l31) * ROWB, w_row0 = A16 + (wc * 64 + l31) * ROWB;
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    barrier_all();
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + (kt & 1) * STAGE;
        uint4 keep[CONV ? MI : 1][4];  // CONV: the lane's 32 values of each row block (4 q steps x 8 halves)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int so = ((half * 4 + q) ^ swz) << 4;
            uint4 fa[MI], fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const uint4*)(st + w_row0 + j * 32 * ROWB + so);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *(const uint4*)(st + a_row0 + i * 32 * ROWB + so);
            if (q == 0 && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (CONV) keep[i][q] = fa[i];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j], 0, 0, 0);
            }
        }
        if (MX) {
            const char* sc = smem + 2 * STAGE + ((kt >> 1) & 1) * SCB + (kt & 1) * 2 + half;
            uint4 a4[MI], b4[2];
            int sa[MI], sb[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wr * WTM + i * 32 + l31;
                if constexpr (CONV) {
                    // block max of the 32 halves (|x| as bit patterns: positive fp16 order like unsigned integers), scale 2^e with
                    // max / 2^e <= 6, sixteen pair conversions
                    unsigned mx = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned d[4] = {keep[i][q].x, keep[i][q].y, keep[i][q].z, keep[i][q].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const unsigned a = d[c] & 0x7fff7fffu;
                            mx = max(mx, max(a & 0xffffu, a >> 16));
                        }
                    }
                    const float amax = (float)__builtin_bit_cast(_Float16, (unsigned short)mx);
                    int ex = amax > 0.f ? __builtin_amdgcn_frexp_expf(amax * (1.f / 6.f)) : -127;
                    ex = ex < -127 ? -127 : ex;
                    const float scf = __builtin_amdgcn_ldexpf(1.f, ex);
                    unsigned o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {  // elements 8 q .. 8 q + 7 -> dword q
                        const h2 v0 = __builtin_bit_cast(h2, keep[i][q].x), v1 = __builtin_bit_cast(h2, keep[i][q].y);
                        const h2 v2 = __builtin_bit_cast(h2, keep[i][q].z), v3 = __builtin_bit_cast(h2, keep[i][q].w);
                        unsigned w_ = 0;
                        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, v0, scf, 0);
                        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, v1, scf, 1);
                        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, v2, scf, 2);
                        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w_, v3, scf, 3);
                        o[q] = w_;
                    }
                    a4[i] = make_uint4(o[0], o[1], o[2], o[3]);
                    sa[i] = (ex + 127) * 0x01010101;
                } else {
                    a4[i] = *(const uint4*)(st + A16 + W16 + row * 32 + half * 16);
                    sa[i] = (int)(*(const uint8_t*)(sc + row * 4)) * 0x01010101;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wc * 64 + j * 32 + l31;
                b4[j] = *(const uint4*)(st + OFF_W4 + row * 32 + half * 16);
                sb[j] = (int)(*(const uint8_t*)(sc + (BM + row) * 4)) * 0x01010101;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const v8i av = {(int)a4[i].x, (int)a4[i].y, (int)a4[i].z, (int)a4[i].w, 0, 0, 0, 0};
                    const v8i bv = {(int)b4[j].x, (int)b4[j].y, (int)b4[j].z, (int)b4[j].w, 0, 0, 0, 0};
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 4, 4, 0, sa[i], 0, sb[j]);
                }
        }
        barrier_all();
    }
    if (p.store) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, n = n0 + wc * 64 + j * 32 + l31;
                    if (m < p.M && n < p.N) p.C[(long)m * p.N + n] = acc[i][j][r];
                }
    } else {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 12345.678f) p.C[0] = s;
    }
}

// MX-fp4 image of a (rows, K) fp16 matrix: per row and 32-k block an E8M0 scale 2^e with max|x| / 2^e <= 6 and 32 e2m1 nibbles
__global__ void mx_pack(const _Float16* x, long rows, int K, uint4* data, uint8_t* scales) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int kblocks = K / 32;
    if (idx >= rows * kblocks) return;
    const int blk = (int)(idx % kblocks);
    const _Float16* row = x + (idx / kblocks) * K;
    _Float16 src[32];
    for (int e = 0; e < 32; ++e) src[e] = row[32 * blk + e];
    float amax = 0.f;
    for (int e = 0; e < 32; ++e) amax = fmaxf(amax, fabsf((float)src[e]));
    int ex = amax > 0.f ? (int)ceilf(log2f(amax / 6.f)) : -127;
    ex = ex < -127 ? -127 : ex;
    const float sc = ldexpf(1.f, ex);
    unsigned d[4] = {0, 0, 0, 0};
    for (int e = 0; e < 32; e += 2) {
        h2 v;
        v[0] = src[e];
        v[1] = src[e + 1];
        const int byte = e / 2;
        switch (byte & 3) {
            case 0: d[byte >> 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(d[byte >> 2], v, sc, 0); break;
            case 1: d[byte >> 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(d[byte >> 2], v, sc, 1); break;
            case 2: d[byte >> 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(d[byte >> 2], v, sc, 2); break;
            default: d[byte >> 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(d[byte >> 2], v, sc, 3); break;
        }
    }
    data[idx] = make_uint4(d[0], d[1], d[2], d[3]);
    scales[idx] = (uint8_t)(ex + 127);
}

static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

struct Dev {
    char *A = nullptr, *W2 = nullptr, *Whi = nullptr, *Wlo16 = nullptr, *A4 = nullptr, *W4 = nullptr;
    uint8_t *A4s = nullptr, *W4s = nullptr;
    float *C = nullptr, *Wf = nullptr;
};

__device__ inline float hash_normal(unsigned long i, unsigned seed) {  // ~N(0, 1): sum of four uniforms
    float s = 0.f;
    for (int r = 0; r < 4; ++r) {
        unsigned x = (unsigned)(i * 4 + r) * 2654435761u + seed + (unsigned)(i >> 30) * 97u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        s += (x & 0xffffff) * (1.f / 0x1000000);
    }
    return (s - 2.f) * 1.7320508f;
}
__global__ void gen_a(_Float16* A, long n, int K, unsigned seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        A[i] = (_Float16)(hash_normal(i, seed) * (((i % K) % 193) == 7 ? 50.f : 1.f));  // a few outlier channels
}
__global__ void gen_w(_Float16* hi, _Float16* lo, _Float16* w2, float* wf, long N, int K, unsigned seed) {
    const float ws = rsqrtf((float)K);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N * K; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 40503u + seed * 7u;
        x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
        const float w = hash_normal(i, seed ^ 0x9e3779b9u) * ws * ((x & 63) == 0 ? 8.f : 1.f);  // heavy-tailed
        const _Float16 h = (_Float16)w, l = (_Float16)(w - (float)h);
        const long n = i / K, k = i % K;
        hi[i] = h;
        lo[i] = l;
        w2[n * 2 * K + k] = h;
        w2[n * 2 * K + K + k] = l;
        if (wf) wf[i] = w;
    }
}

static int setup(int M, int N, int K, Dev& d, std::vector<float>* a_host, std::vector<float>* w_host, unsigned seed) {
    const size_t na = (size_t)M * K, nw = (size_t)N * K;
    CK(hipMalloc(&d.A, na * 2 + 256)); CK(hipMalloc(&d.W2, nw * 4)); CK(hipMalloc(&d.Whi, nw * 2)); CK(hipMalloc(&d.Wlo16, nw * 2));
    CK(hipMalloc(&d.A4, na / 2)); CK(hipMalloc(&d.A4s, na / 32)); CK(hipMalloc(&d.W4, nw / 2)); CK(hipMalloc(&d.W4s, nw / 32));
    CK(hipMalloc(&d.C, (size_t)M * N * 4));
    if (w_host) CK(hipMalloc(&d.Wf, nw * 4));
    hipLaunchKernelGGL(gen_a, dim3(2048), dim3(256), 0, 0, (_Float16*)d.A, (long)na, K, seed);
    hipLaunchKernelGGL(gen_w, dim3(2048), dim3(256), 0, 0, (_Float16*)d.Whi, (_Float16*)d.Wlo16, (_Float16*)d.W2, d.Wf, (long)N, K, seed + 1);
    const long ba = (long)na / 32, bw = (long)nw / 32;
    hipLaunchKernelGGL(mx_pack, dim3((unsigned)((ba + 255) / 256)), dim3(256), 0, 0, (const _Float16*)d.A, (long)M, K, (uint4*)d.A4, d.A4s);
    hipLaunchKernelGGL(mx_pack, dim3((unsigned)((bw + 255) / 256)), dim3(256), 0, 0, (const _Float16*)d.Wlo16, (long)N, K, (uint4*)d.W4, d.W4s);
    CK(hipDeviceSynchronize());
    if (a_host) {
        std::vector<uint16_t> h(na);
        CK(hipMemcpy(h.data(), d.A, na * 2, hipMemcpyDeviceToHost));
        a_host->resize(na);
        for (size_t i = 0; i < na; ++i) (*a_host)[i] = h2f(h[i]);
    }
    if (w_host) {
        w_host->resize(nw);
        CK(hipMemcpy(w_host->data(), d.Wf, nw * 4, hipMemcpyDeviceToHost));
    }
    return 0;
}
static void release(Dev& d) {
    for (void* q : {(void*)d.A, (void*)d.W2, (void*)d.Whi, (void*)d.Wlo16, (void*)d.A4, (void*)d.W4, (void*)d.A4s, (void*)d.W4s, (void*)d.C, (void*)d.Wf}) (void)hipFree(q);
}

template <int MODE>
static int launch(const Dev& d, int M, int N, int K, int store) {
    constexpr bool MX = MODE != 0, CONV = MODE == 2;
    constexpr int STAGE = A16 + W16 + (MX ? (CONV ? 0 : A4B) + W4B : 0);
    const int lds = 2 * STAGE + (MX ? 2 * (BM + BN) * 4 : 0);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)gemm<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); set = true; }
    P p{d.A, MX ? d.Whi : d.W2, d.A4, d.A4s, d.W4, d.W4s, d.C, M, N, K, store};
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm<MODE>, dim3(tiles), dim3(NTHR), lds, 0, p);
    return 0;
}
static int launch_mode(int mode, const Dev& d, int M, int N, int K, int store) {
    return mode == 0 ? launch<0>(d, M, N, K, store) : mode == 1 ? launch<1>(d, M, N, K, store) : launch<2>(d, M, N, K, store);
}

int main() {
    printf("# mx_gemm_lab: fp16 activations x (fp16 hi + second weight term), 192 x 256 tile, 2 stages of 64 k, one tile per workgroup\n");
    {   // correctness against a float64 product
        const int M = 400, N = 520, K = 768;
        Dev d;
        std::vector<float> a, w;
        if (setup(M, N, K, d, &a, &w, 1)) return 1;
        std::vector<double> ref((size_t)M * N), ref_hi((size_t)M * N);
        double nrm = 0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double s = 0, sh = 0;
                for (int k = 0; k < K; ++k) {
                    s += (double)a[(size_t)m * K + k] * w[(size_t)n * K + k];
                    sh += (double)a[(size_t)m * K + k] * (double)(float)(_Float16)w[(size_t)n * K + k];
                }
                ref[(size_t)m * N + n] = s;
                ref_hi[(size_t)m * N + n] = sh;
                nrm += s * s;
            }
        std::vector<float> c((size_t)M * N);
        auto err = [&](const std::vector<double>& r) {
            double e = 0;
            for (size_t i = 0; i < c.size(); ++i) e += (c[i] - r[i]) * (c[i] - r[i]);
            return std::sqrt(e / nrm);
        };
        printf("\n## relative error of the product against float64 (M = %d, N = %d, K = %d; outlier activation channels, heavy-tailed weights)\n\n", M, N, K);
        printf("| weights | error vs A x W | error vs A x fp16(W) |\n|---|---:|---:|\n");
        const char* vn[3] = {"two fp16 terms (K loop over 2K)", "fp16 + MX-fp4 second term, A image staged (a producer wrote it)",
                             "fp16 + MX-fp4 second term, A image built in the kernel from the fp16 fragments"};
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(d.C, 0xff, c.size() * 4));
            if (launch_mode(mode, d, M, N, K, 1)) return 1;
            CK(hipMemcpy(c.data(), d.C, c.size() * 4, hipMemcpyDeviceToHost));
            printf("| %s | %.3e | %.3e |\n", vn[mode], err(ref), err(ref_hi));
        }
        {   // one term: the fp16 error floor of the weights (host)
            double e = 0;
            for (size_t i = 0; i < ref.size(); ++i) e += (ref_hi[i] - ref[i]) * (ref_hi[i] - ref[i]);
            printf("| one fp16 term (host, float64 sums) | %.3e | 0 |\n", std::sqrt(e / nrm));
        }
        release(d);
    }
    {   // timing
        struct Shape { const char* name; int M, N, K; };
        const Shape shapes[] = {{"qkv 15968x2304x768", 15968, 2304, 768}, {"fc1 15968x3072x768", 15968, 3072, 768}, {"fc2 15968x768x3072", 15968, 768, 3072},
                                {"conv1-like 127984x512x1536", 127984, 512, 1536}, {"L.fc1 15968x4096x1024", 15968, 4096, 1024}};
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("\n## time per launch, us (algorithmic TFLOP/s = 2 M N K / t): K loop only (no stores) | with the plain fp32 epilogue\n\n");
        printf("| shape | two fp16 terms | MX, A image staged | speed-up | MX, A image built in the kernel | speed-up | two terms, stores | MX staged, stores | speed-up | MX in-kernel, stores | speed-up |\n"
               "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n");
        for (const Shape& sh : shapes) {
            Dev d;
            if (setup(sh.M, sh.N, sh.K, d, nullptr, nullptr, 7)) return 1;
            double t[2][3];
            for (auto& row : t) for (double& v : row) v = 1e30;
            for (int store = 0; store < 2; ++store)
                for (int r = 0; r < 3; ++r)
                    for (int v = 0; v < 3; ++v) {
                        if (launch_mode(v, d, sh.M, sh.N, sh.K, store)) return 1;
                        CK(hipEventRecord(e0, 0));
                        for (int k = 0; k < 20; ++k)
                            if (launch_mode(v, d, sh.M, sh.N, sh.K, store)) return 1;
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r && ms / 20 < t[store][v]) t[store][v] = ms / 20;
                    }
            const double fl = 2.0 * sh.M * (double)sh.N * sh.K;
            printf("| %s |", sh.name);
            for (int store = 0; store < 2; ++store) {
                printf(" %.1f (%.0f) |", t[store][0] * 1e3, fl / t[store][0] * 1e-9);
                for (int v = 1; v < 3; ++v) printf(" %.1f (%.0f) | %.2f |", t[store][v] * 1e3, fl / t[store][v] * 1e-9, t[store][0] / t[store][v]);
            }
            printf("\n");
            fflush(stdout);
            release(d);
        }
    }
    return 0;
}
