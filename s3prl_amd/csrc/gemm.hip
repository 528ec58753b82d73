// gemm.hip — the one dense-contraction kernel of the encoder, written for gfx950 MFMA.
//
//   out[b][m][n] = epilogue( sum_k A[b][m][k] * W[n][k] )
//
// It serves every GEMM-shaped op of the path (SURVEY §2.3 K5,K6,K8,K13,K15,K16,K17):
//   * conv1..6 of the feature extractor as an *implicit* GEMM: activations are kept channel-last
//     (B, L, C), so the k*C inputs of output frame t are the contiguous span starting at frame s*t —
//     a GEMM whose A rows overlap (lda = s*C < K = k*C).  No im2col buffer exists.
//   * post_extract_proj, q|k|v (one N=3D GEMM), out_proj, fc1, fc2.
// Epilogue (fused, fp32): + bias, erf-GELU, + fp32 residual, zeroing of padded frames, and a dual
// store: fp32 (residual stream / LayerNorm input) and/or the 16-bit operand of the next GEMM.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 wave64 as 2x2, 64x64 per wave = 2x2
// MFMA 32x32 tiles, 64 accumulator VGPRs).  A K-tile is 128 BYTES of every row for both operands
// (32 fp32 / 64 bf16|f16), staged global -> VGPR -> LDS with a register double buffer, LDS double
// buffered, one barrier per K-tile.  LDS rows are 128 B with the 16-B slot XOR-swizzled by
// ((row>>1)&7) so that every ds_read_b128 fragment read is bank-conflict free (MI355X_MICROARCH §LDS).
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32) or v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate.
// The k index inside a K-tile is permuted identically for A and W (each half-wave owns one
// contiguous 64-byte half of the row) so fragments are read as 16-byte vectors.
#include "kernels.h"

namespace s3 {

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;  // tile rows / cols, bytes of K per row per stage
constexpr int STAGE_BYTES = (BM + BN) * ROWB;  // 32 KiB
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // 64 KiB

template <typename T> struct Mma;
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_tag> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<f16_tag> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
    typedef typename Cvt<T>::store_t store_t;
    constexpr int EB = sizeof(store_t);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int n_tiles = (p.N + BN - 1) / BN;
    const int tm = blockIdx.x / n_tiles, tn = blockIdx.x % n_tiles;
    const int b = blockIdx.y;
    const int m0 = tm * BM, n0 = tn * BN;

    const long lda_b = p.lda * EB;
    const long kbytes = (long)p.K * EB;  // bytes of one full row of K
    const char* Ab = (const char*)p.A + (long)b * p.a_bs * EB;
    const char* Wb = (const char*)p.W;

    // ---- loader: thread owns 16-B slot `ls` of rows lr, lr+32, lr+64, lr+96 of both operand tiles ----
    const int ls = tid & 7;
    const int lr = tid >> 3;
    const char* a_ptr[4];
    const char* w_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = m0 + lr + 32 * i;
        ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + lr + 32 * i;
        rw = rw < p.N ? rw : p.N - 1;
        a_ptr[i] = Ab + (long)ra * lda_b + ls * 16;
        w_ptr[i] = Wb + (long)rw * kbytes + ls * 16;
    }
    const int st_off = lr * ROWB + ((ls ^ ((lr >> 1) & 7)) << 4);  // + i*32*ROWB

    uint4 ga[4], gw[4];
    auto load_tile = [&](int kt) {
        const long kb = (long)kt * ROWB;
        const bool ok = kb + ls * 16 < kbytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ga[i] = ok ? *(const uint4*)(a_ptr[i] + kb) : make_uint4(0, 0, 0, 0);
            gw[i] = ok ? *(const uint4*)(w_ptr[i] + kb) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sw = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(uint4*)(sa + st_off + i * 32 * ROWB) = ga[i];
            *(uint4*)(sw + st_off + i * 32 * ROWB) = gw[i];
        }
    };

    // ---- fragment addresses ----
    const int swz = (l31 >> 1) & 7;
    const int a_row0 = (wr * 64 + l31) * ROWB;
    const int w_row0 = BM * ROWB + (wc * 64 + l31) * ROWB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (int)((kbytes + ROWB - 1) / ROWB);
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* st = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int so = ((half * 4 + q) ^ swz) << 4;
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const uint4*)(st + a_row0 + i * 32 * ROWB + so);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const uint4*)(st + w_row0 + j * 32 * ROWB + so);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc[i][j][r] is (row = wr*64+i*32 + (r&3)+8*(r>>2)+4*half, col = wc*64+j*32+l31) ----
    const int limit = p.row_limit ? p.row_limit[b] : p.M;
    const long ob = (long)b * p.o_bs;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + l31;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bias;
                if (p.act) v = gelu_erf(v);
                const long o = ob + (long)m * p.ldo + n;
                if (p.residual) v += p.residual[o];
                if (m >= limit) v = 0.f;
                if (p.out32) p.out32[o] = v;
                if (p.out16) ((store_t*)p.out16)[o] = Cvt<T>::to(v);
            }
        }
    }
}

}  // namespace

hipError_t launch_gemm(int dtype, const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.batches <= 0) return hipSuccess;
    const int eb = dtype == F32 ? 4 : 2;
    // 16-byte vector loads: row starts and K must be 16-byte granular
    if (((p.lda * eb) & 15) || ((p.a_bs * eb) & 15) || (((long)p.K * eb) & 15) || (((uintptr_t)p.A) & 15) ||
        (((uintptr_t)p.W) & 15))
        return hipErrorInvalidValue;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), p.batches);
    dim3 block(256);
    hipError_t e;
    switch (dtype) {
        case F32:
            e = hipFuncSetAttribute((const void*)gemm_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(gemm_kernel<float>, grid, block, GEMM_LDS, stream, p);
            break;
        case BF16:
            e = hipFuncSetAttribute((const void*)gemm_kernel<bf16_tag>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(gemm_kernel<bf16_tag>, grid, block, GEMM_LDS, stream, p);
            break;
        case F16:
            e = hipFuncSetAttribute((const void*)gemm_kernel<f16_tag>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(gemm_kernel<f16_tag>, grid, block, GEMM_LDS, stream, p);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace s3
