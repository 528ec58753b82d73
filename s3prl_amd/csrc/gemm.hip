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
// MFMA 32x32 tiles, 64 accumulator VGPRs).  A K-stage is ROWB = 64 (default) or 128 BYTES of every row of
// both operands, double buffered in LDS, one barrier per stage; 64-byte stages need 32 KiB of LDS so four or five
// workgroups share a CU (VGPR-limited), which measured +5..10 % (fp32) / +20..40 % (bf16) over 128-byte stages.
// Staging is, when K is a multiple of the stage (default), LDS-DMA (global_load_lds_dwordx4 from inline asm: no staging
// VGPRs — 95 instead of 119, five waves per SIMD — no ds_write pass, +4..6 % on every shape of the path), otherwise
// global -> VGPR -> ds_write (register double buffer; handles a ragged K tail).
// LDS rows are XOR-swizzled at 16-byte granularity (slot ^= f(row)) so every ds_read_b128 fragment read is
// bank-conflict free (MI355X_MICROARCH §LDS); with LDS-DMA the LDS image is lane-linear, so the same
// permutation is applied to the per-lane SOURCE address instead (cdna_hip_programming.md rule 21).
// Workgroups are mapped XCD-aware: every XCD (private L2) gets a contiguous range of (batch, m-tile, n-tile).
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32) or v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate.
// The k index inside a K-tile is permuted identically for A and W (each half-wave owns one
// contiguous 64-byte half of the row) so fragments are read as 16-byte vectors.
#include "kernels.h"

namespace s3 {

namespace {

constexpr int BM = 128, BN = 128;  // tile rows / cols
// ROWB = bytes of K per row per LDS stage (128: 64 KiB of LDS, 2 workgroups/CU; 64: 32 KiB, VGPR-limited 3/CU)

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

template <typename T, int ROWB, bool GLDS, bool VEC>
__global__ __launch_bounds__(256, ROWB == 128 ? 2 : (GLDS ? 5 : 3)) void gemm_kernel(GemmParams p) {
    typedef typename Cvt<T>::store_t store_t;
    constexpr int EB = sizeof(store_t);
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int SLOTS = ROWB / 16;          // 16-byte slots per row per stage (8 or 4)
    constexpr int SMASK = SLOTS - 1;
    constexpr int SSH = ROWB == 128 ? 1 : 2;  // swizzle: slot ^= (row >> SSH) & SMASK
    constexpr int RPT = 256 / SLOTS;          // rows covered by one pass of the 256 loader threads (32 or 64)
    constexpr int NLD = BM / RPT;             // 16-byte loads per thread per operand per stage (4 or 2)
    constexpr int NQ = SLOTS / 2;             // fragment steps per stage (4 or 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware tile order: workgroup w runs on XCD w%8 (each XCD has a private 4 MiB L2), so give every XCD a
    // CONTIGUOUS range of the (batch, m-tile, n-tile) sequence with n fastest: the n-tiles that share an A row panel
    // and the m-tiles that share W run back to back on the same L2 (bijective for any grid size).
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m_tiles = (p.M + BM - 1) / BM;
    int tile;
    {
        const int nwg = gridDim.x, wg = blockIdx.x;
        if (p.variant & 4) {
            tile = wg;
        } else {
            const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wg & 7, loc = wg >> 3;
            tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        }
    }
    const int tn = tile % n_tiles;
    const int tmb = tile / n_tiles;
    const int tm = tmb % m_tiles, b = tmb / m_tiles;
    const int m0 = tm * BM, n0 = tn * BN;

    const long lda_b = p.lda * EB;
    const long ka_bytes = (long)p.K * EB;                  // bytes of A's K extent
    const long kbytes = p.wsplit ? 2 * ka_bytes : ka_bytes;  // bytes of the contraction ([hi | lo] weights: launcher checks K % stage)
    const long ldw_b = p.ldw ? p.ldw * EB : kbytes;          // W row stride
    const char* Ab = (const char*)p.A + (long)b * p.a_bs * EB;
    const char* Wb = (const char*)p.W;
    const int nk = (int)((kbytes + ROWB - 1) / ROWB);

    // ---- loader: thread owns one 16-B slot of rows lr, lr+RPT, ... of both operand tiles.  Consecutive lanes fill
    //      consecutive 16-B slots of LDS (lane-linear: what LDS-DMA requires); the slot each lane FETCHES is the
    //      swizzle-inverse, so the LDS image ends up swizzled either way. ----
    const int ps = tid & SMASK;   // physical slot
    const int lr = tid / SLOTS;   // row within a pass
    const int ls = ps ^ ((lr >> SSH) & SMASK);  // logical slot (RPT is a multiple of 16: same for every pass)
    const char* a_ptr[NLD];
    const char* w_ptr[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int ra = m0 + lr + RPT * i;
        ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + lr + RPT * i;
        rw = rw < p.N ? rw : p.N - 1;
        a_ptr[i] = Ab + (long)ra * lda_b + ls * 16;
        w_ptr[i] = Wb + (long)rw * ldw_b + ls * 16;
    }
    const int st_off = lr * ROWB + (ps << 4);  // + i*RPT*ROWB ; == tid*16 + i*4096

    // ---- fragment addresses ----
    const int swz = (l31 >> SSH) & SMASK;
    const int a_row0 = (wr * 64 + l31) * ROWB;
    const int w_row0 = BM * ROWB + (wc * 64 + l31) * ROWB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int stage) {
        const char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int so = ((half * NQ + q) ^ swz) << 4;
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
    };

    if constexpr (GLDS) {
        // LDS-DMA staging: each wave instruction lands 64 x 16 B = 1 KiB contiguously at a wave-uniform LDS base.
        // Issued from inline asm (M0 = destination written in the same statement): through the builtin hipcc treats the
        // DMA as a pending LDS write it cannot disambiguate and drains it (vmcnt(0)) before the first fragment read of
        // the stage being multiplied, which serialises load and compute; here completion is waited for by hand.
        const unsigned lds0 =
            __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
        auto dma = [&](const char* gsrc, unsigned dst) {
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" S3_GLDS_MOD "\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(gsrc), "s"(dst)
                : "memory");
        };
        auto issue = [&](int kt, int stage) {
            const long kb = (long)kt * ROWB;
            const long kba = kb >= ka_bytes ? kb - ka_bytes : kb;  // wsplit: the lo half re-reads A
            const unsigned sa = lds0 + stage * STAGE_BYTES;  // == st_off - lane*16
            const unsigned sw = sa + BM * ROWB;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                dma(a_ptr[i] + kba, sa + i * 4096);
                dma(w_ptr[i] + kb, sw + i * 4096);
            }
        };
        auto stage_barrier = [&]() {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        issue(0, 0);
        stage_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
            compute(kt & 1);
            stage_barrier();  // my DMA landed + my fragment reads done, then everybody's
        }
    } else {
        uint4 ga[NLD], gw[NLD];
        auto load_tile = [&](int kt) {
            const long kb = (long)kt * ROWB;
            const long kba = kb >= ka_bytes ? kb - ka_bytes : kb;
            const bool ok = kb + ls * 16 < kbytes;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                ga[i] = ok ? *(const uint4*)(a_ptr[i] + kba) : make_uint4(0, 0, 0, 0);
                gw[i] = ok ? *(const uint4*)(w_ptr[i] + kb) : make_uint4(0, 0, 0, 0);
            }
        };
        auto store_tile = [&](int stage) {
            char* sa = smem + stage * STAGE_BYTES;
            char* sw = sa + BM * ROWB;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                *(uint4*)(sa + st_off + i * RPT * ROWB) = ga[i];
                *(uint4*)(sw + st_off + i * RPT * ROWB) = gw[i];
            }
        };
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1);
            compute(kt & 1);
            if (kt + 1 < nk) store_tile((kt + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: acc[i][j][r] is (row = wr*64+i*32 + (r&3)+8*(r>>2)+4*half, col = wc*64+j*32+l31) ----
    const int limit = p.row_limit ? p.row_limit[b] : p.M;
    const long ob = (long)b * p.o_bs;
    if constexpr (VEC) {
        // vector epilogue (launcher-checked: N, ldo, o_bs multiples of 4, 16-byte aligned pointers): the accumulators go
        // through a wave-private LDS transpose (32 x 64 fp32 per step; the stage buffers are free after the last
        // barrier) so that bias / GELU / residual run on row-contiguous float4s and every global access is a 16-byte
        // (fp32) or 8-byte (16-bit) vector — 16 store instructions per wave instead of 64
        float* stg = (float*)(smem + wave * 8192);
        const int c4 = (lane & 15) * 4;
        const int n = n0 + wc * 64 + c4;
        const bool n_ok = n < p.N;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && n_ok) bias4 = *(const float4*)(p.bias + n);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = t * 4 + (lane >> 4);
                float4 v = *(const float4*)(stg + row * 64 + c4);
                const int m = m0 + wr * 64 + i * 32 + row;
                if (m < p.M && n_ok) {
                    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                    if (p.act) {
                        if (!std::is_same<T, float>::value || p.act == 2) {
                            gelu_fast4(v);
                        } else {
                            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                        }
                    }
                    const long o = ob + (long)m * p.ldo + n;
                    if (p.residual) {
                        const float4 rs = *(const float4*)(p.residual + o);
                        v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
                    }
                    if (m >= limit) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.out32) *(float4*)(p.out32 + o) = v;
                    if (p.out16) {
                        ushort4 h;
                        h.x = Cvt<T>::to(v.x); h.y = Cvt<T>::to(v.y); h.z = Cvt<T>::to(v.z); h.w = Cvt<T>::to(v.w);
                        *(ushort4*)((store_t*)p.out16 + o) = h;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
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
                if (p.act) v = (p.act == 2) ? gelu_fast(v) : gelu_mode<T>(v);
                const long o = ob + (long)m * p.ldo + n;
                if (p.residual) v += p.residual[o];
                if (m >= limit) v = 0.f;
                if (p.out32) p.out32[o] = v;
                if (p.out16) ((store_t*)p.out16)[o] = Cvt<T>::to(v);
            }
        }
    }
    }
}

}  // namespace
namespace {

template <typename T, int ROWB, bool GLDS, bool VEC>
hipError_t gemm_go(const GemmParams& p, dim3 grid, hipStream_t stream) {
    constexpr int lds0 = 2 * (BM + BN) * ROWB;
    static_assert(lds0 >= 4 * 8192, "epilogue staging must fit");
    // tuning key "gemm_lds_pad" (occupancy probe): unused extra LDS so that fewer workgroups fit a CU
    const int lds = lds0 + tuning().gemm_lds_pad;
    hipError_t e = ensure_dynamic_lds<gemm_kernel<T, ROWB, GLDS, VEC>>(lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_kernel<T, ROWB, GLDS, VEC>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

// variant bits: 1 = 64-byte K stages (else 128), 2 = LDS-DMA staging, 4 = no XCD-aware tile order (A/B only),
//               16 = vector epilogue (set by the launcher when the alignment allows), 32 = force the scalar epilogue
template <typename T, bool VEC>
hipError_t gemm_variant2(const GemmParams& p, dim3 grid, hipStream_t stream, bool k_aligned64, bool k_aligned128) {
    const bool small = p.variant & 1;
    const bool glds = (p.variant & 2) && (small ? k_aligned64 : k_aligned128);  // DMA cannot zero-fill a ragged K tail
    if (small) return glds ? gemm_go<T, 64, true, VEC>(p, grid, stream) : gemm_go<T, 64, false, VEC>(p, grid, stream);
    return glds ? gemm_go<T, 128, true, VEC>(p, grid, stream) : gemm_go<T, 128, false, VEC>(p, grid, stream);
}
template <typename T>
hipError_t gemm_variant(const GemmParams& p, dim3 grid, hipStream_t stream, bool k_aligned64, bool k_aligned128) {
    return (p.variant & 16) ? gemm_variant2<T, true>(p, grid, stream, k_aligned64, k_aligned128)
                            : gemm_variant2<T, false>(p, grid, stream, k_aligned64, k_aligned128);
}

}  // namespace


hipError_t launch_gemm(int dtype, const GemmParams& p0, hipStream_t stream) {
    GemmParams p = p0;
    if (p.variant < 0) p.variant = tuning().gemm_variant;  // default 3: 64-byte stages + LDS-DMA (register staging for a ragged K)
    if (p.M <= 0 || p.N <= 0 || p.batches <= 0) return hipSuccess;
    if (dtype == F32 && p.act == 1 && tuning().gelu32 == 1) p.act = 2;  // fp32 products, the one-transcendental GELU
    if (p.res_ln_stats && (dtype == F32 || !p.res_ln_g || !p.res_ln_b || !gemm16_res_ln_ok(dtype, p))) return hipErrorInvalidValue;
    if (p.wsplit) {
        if (dtype == F32) return hipErrorInvalidValue;
        if (!p.ldw) p.ldw = 2L * p.K;
        if (((long)p.K * 2) & 127) p.wsplit = 0;  // the lo half must start on a stage boundary: else this GEMM runs on hi alone
        // the lo term as an MX-fp4 image (round 5): the contraction runs over the hi half only, the image supplies the rest
        if (p.wsplit && p.mxw && gemm16_mx_eligible(dtype, p)) {
            p.wsplit = 0;
            return launch_gemm16_big(dtype, p, stream);
        }
    }
    p.mxw = 0;
    if (dtype == F32 && gemm_x3_eligible(p)) {
        if (gemm_tile_eligible(3, p)) return launch_gemm_tile(3, p, stream);
        return launch_gemm_x3(p, stream);
    }
    if (gemm_tile_eligible(dtype, p)) return launch_gemm_tile(dtype, p, stream);  // gemmt.hip: the default of the path's shapes
    if (gemm16_big_eligible(dtype, p)) return launch_gemm16_big(dtype, p, stream);
    p.variant &= 7 | 32;  // bit 5 (tuning): force the scalar epilogue
    {   // vector epilogue when every output / residual / bias access can be a 16-byte (8-byte for 16-bit) vector
        const uintptr_t al = (uintptr_t)p.out32 | (uintptr_t)p.residual | (uintptr_t)p.bias;
        const bool ok = !(p.N & 3) && !(p.ldo & 3) && !(p.o_bs & 3) && !(al & 15) && !(((uintptr_t)p.out16) & 7);
        if (ok && !(p.variant & 32)) p.variant |= 16;
    }
    const int eb = dtype == F32 ? 4 : 2;
    // 16-byte vector loads: row starts and K must be 16-byte granular
    if (((p.lda * eb) & 15) || ((p.a_bs * eb) & 15) || (((long)p.K * eb) & 15) || (((uintptr_t)p.A) & 15) ||
        (((uintptr_t)p.W) & 15))
        return hipErrorInvalidValue;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batches);
    if (p.ldw && (((p.ldw * eb) & 15))) return hipErrorInvalidValue;
    const bool a64 = (((long)p.K * eb) & 63) == 0, a128 = (((long)p.K * eb) & 127) == 0;
    switch (dtype) {
        case F32: return gemm_variant<float>(p, grid, stream, a64, a128);
        case BF16: return gemm_variant<bf16_tag>(p, grid, stream, a64, a128);
        case F16: return gemm_variant<f16_tag>(p, grid, stream, a64, a128);
    }
    return hipErrorInvalidValue;
}

}  // namespace s3
