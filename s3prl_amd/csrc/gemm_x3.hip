// gemm_x3.hip — fp32-class GEMM on the 16-bit matrix pipe ("bf16x3" split-precision emulation).
//
//   out[b][m][n] = epilogue( sum_k A[b][m][k] * W[n][k] )        (same contract and fp32 operands as gemm.hip)
//
// gfx950 has no TF32-like MFMA: exact fp32 products (v_mfma_f32_32x32x2_f32) run at 1/16 of the bf16 rate.  Here every
// fp32 operand is split into two bf16 numbers, x = hi + lo + r with |r| <= 2^-17 |x| (hi = bf16(x), lo = bf16(x - hi)),
// and the product is rebuilt from three bf16 MFMAs with fp32 accumulation,
//     a*w  ~=  a_hi*w_hi + a_lo*w_hi + a_hi*w_lo          (dropped: a_lo*w_lo ~ 2^-18, truncation r ~ 2^-17)
// i.e. a relative error of ~1e-5 per product (random sign) at 3/16 of the fp32 MFMA cost.  No range is lost (bf16 has
// the fp32 exponent).  It is an OPT-IN mode (compute_dtype S3ENC_F32X3): activations, residual stream, norms and the
// softmax stay exactly as in the fp32 mode — only the matrix products change: the GEMMs (here), the positional conv
// (posconv.hip) and the two products of the attention (attention.hip: attn_x3_kernel), all with the same split.
//   * A (activations) stays fp32 in memory and in LDS; the split happens on the fragment registers
//     (v_cvt_pk_bf16_f32 + 2 v_sub + a shift / mask per pair: ~100 VALU per 24 MFMAs, hidden beside them).
//   * W is split once at pack time into a "pair-packed" image with the same bytes and the same addressing as fp32:
//     per row, every group of 8 k-values is 16 bytes of hi followed by 16 bytes of lo.
//   * tile / pipeline as gemm16.hip mode 1: 256x256 (or 192x256) tile, 8 waves, one workgroup per CU, two LDS stages of
//     128 bytes per row (= 32 k here), LDS-DMA from inline asm interleaved with the MFMA steps, source-side XOR
//     swizzle; 48 MFMAs per wave between barriers.
//   * epilogue: wave-private LDS transpose, float4 stores, fp32 residual; GELU in the one-transcendental form of the
//     16-bit modes (common.h gelu_fast: fp32 rounding level; libm erff costs three times the VALU).
// Requirements (else the launcher falls back to the exact kernel): K % 32 == 0, N / ldo / o_bs % 4 == 0, 16-byte
// alignment, M and N >= 128, the pair-packed weights present.
#include <type_traits>

#include "kernels.h"

namespace s3 {

namespace {

constexpr int XBN = 256, XROWB = 128;  // tile columns; bytes per row per stage (32 fp32 k-values)

__device__ __forceinline__ f32x16 mma_bf16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int WTM>
__global__ __launch_bounds__(512, 2) void gemm_x3_kernel(GemmParams p) {
    asm volatile("" ::: "v255");  // the whole register file of the SIMD: no foreign wave beside these 16-bit MFMAs (see gemm16.hip)
    constexpr int BM = 2 * WTM, MI = WTM / 32;
    constexpr int A_BYTES = BM * XROWB, B_BYTES = XBN * XROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NLA = BM / 64, NLB = XBN / 64, NL = NLA + NLB;  // LDS-DMA pieces per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;

    const int n_tiles = (p.N + XBN - 1) / XBN;
    const int m_tiles = (p.M + BM - 1) / BM;
    int tile;
    {
        const int nwg = gridDim.x, wg = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wg & 7, loc = wg >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    }
    const int tn = tile % n_tiles, tmb = tile / n_tiles;
    const int tm = tmb % m_tiles, b = tmb / m_tiles;
    const int m0 = tm * BM, n0 = tn * XBN;

    const long lda_b = p.lda * 4, kbytes = (long)p.K * 4;
    const char* Ab = (const char*)p.A + (long)b * p.a_bs * 4;
    const char* Wb = (const char*)p.W_x3;
    const int nk = (int)(kbytes / XROWB);

    const int ps = tid & 7, lr = tid >> 3;  // physical 16-byte slot, row within a 64-row pass
    const int ls = ps ^ ((lr >> 1) & 7);    // logical slot fetched (source-side swizzle)
    const char* a_ptr[NLA];
    const char* w_ptr[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        int ra = m0 + lr + 64 * i;
        ra = ra < p.M ? ra : p.M - 1;
        a_ptr[i] = Ab + (long)ra * lda_b + ls * 16;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        int rw = n0 + lr + 64 * i;
        rw = rw < p.N ? rw : p.N - 1;
        w_ptr[i] = Wb + (long)rw * kbytes + ls * 16;
    }
    const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
    auto dma = [&](const char* gsrc, unsigned dst) {
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" S3_GLDS_MOD "\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(dst)
            : "memory");
    };
    auto issue_piece = [&](int pc, int kt, int stage) {
        const long kb = (long)kt * XROWB;
        const unsigned sa = lds_base + stage * STAGE_BYTES;
        if (pc < NLA) dma(a_ptr[pc] + kb, sa + pc * 8192);
        else dma(w_ptr[pc - NLA] + kb, sa + A_BYTES + (pc - NLA) * 8192);
    };
    auto barrier_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    const int swz = (l31 >> 1) & 7;
    const int a_row0 = (wr * WTM + l31) * XROWB;
    const int w_row0 = A_BYTES + (wc * 64 + l31) * XROWB;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // a stage holds 32 k = two 16-deep MFMA steps; the pieces of the next stage are issued beside the first step
    auto compute = [&](int stage, bool pf, int kt_pf, int stage_pf) {
        const char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // this half-wave's 8 k-values of step q: logical 16-byte slots 4q + 2*half and +1
            const int s0 = ((4 * q + 2 * half) ^ swz) << 4, s1 = ((4 * q + 2 * half + 1) ^ swz) << 4;
            uint4 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {  // W is pair-packed: slot 2g = hi, slot 2g + 1 = lo of k-group g
                bh[j] = *(const uint4*)(st + w_row0 + j * 32 * XROWB + s0);
                bl[j] = *(const uint4*)(st + w_row0 + j * 32 * XROWB + s1);
            }
            float4 ax[MI][2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ax[i][0] = *(const float4*)(st + a_row0 + i * 32 * XROWB + s0);
                ax[i][1] = *(const float4*)(st + a_row0 + i * 32 * XROWB + s1);
            }
            if (pf && q == 0) {
#pragma unroll
                for (int pc = 0; pc < NL; ++pc) issue_piece(pc, kt_pf, stage_pf);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                uint4 ah, al;
                split8(ax[i][0], ax[i][1], ah, al);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = mma_bf16(al, bh[j], acc[i][j]);
                    acc[i][j] = mma_bf16(ah, bl[j], acc[i][j]);
                    acc[i][j] = mma_bf16(ah, bh[j], acc[i][j]);
                }
            }
        }
    };

#pragma unroll
    for (int pc = 0; pc < NL; ++pc) issue_piece(pc, 0, 0);
    barrier_all();
    for (int kt = 0; kt < nk; ++kt) {
        compute(kt & 1, kt + 1 < nk, kt + 1, (kt + 1) & 1);
        barrier_all();
    }

    // ---- epilogue through a wave-private LDS transpose (fp32 out only) ----
    float* stg = (float*)(smem + wave * 8192);
    const int limit = p.row_limit ? p.row_limit[b] : p.M;
    const long ob = (long)b * p.o_bs;
    const int c4 = (lane & 15) * 4;
    const int n = n0 + wc * 64 + c4;
    const bool n_ok = n < p.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) bias4 = *(const float4*)(p.bias + n);
    auto epilogue = [&](auto act_c, auto res_c) {
        constexpr bool ACT = decltype(act_c)::value, RES = decltype(res_c)::value;
        // the residual rows of a 32-row block are loaded one block AHEAD, before that block's LDS transpose: in the forward the
        // residual is cold and a load issued right before its use stalled every block for a memory round trip (see gemm16.hip)
        float4 rsb[2][RES ? 8 : 1];
        auto res_load = [&](int i, float4* dst) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int m = m0 + wr * WTM + i * 32 + t * 4 + (lane >> 4);
                dst[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < p.M && n_ok) dst[t] = *(const float4*)(p.residual + ob + (long)m * p.ldo + n);
            }
        };
        if constexpr (RES) res_load(0, rsb[0]);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if constexpr (RES) {
                if (i + 1 < MI) res_load(i + 1, rsb[(i + 1) & 1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = t * 4 + (lane >> 4);
                float4 v = *(const float4*)(stg + row * 64 + c4);
                const int m = m0 + wr * WTM + i * 32 + row;
                if (m < p.M && n_ok) {
                    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                    if (ACT) {
                        gelu_fast4(v);
                    }
                    const long o = ob + (long)m * p.ldo + n;
                    if (RES) {
                        const float4 rs = rsb[i & 1][t];
                        v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
                    }
                    if (m >= limit) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *(float4*)(p.out32 + o) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    using TT = std::true_type;
    using FF = std::false_type;
    const bool a = p.act != 0, r = p.residual != nullptr;
    if (a && !r) epilogue(TT{}, FF{});
    else if (!a && r) epilogue(FF{}, TT{});
    else if (!a && !r) epilogue(FF{}, FF{});
    else epilogue(TT{}, TT{});
}

template <int WTM>
hipError_t x3_go(const GemmParams& p, hipStream_t stream) {
    constexpr int BM = 2 * WTM;
    constexpr int lds = 2 * (BM + XBN) * XROWB;
    static_assert(lds >= 8 * 8192, "epilogue staging must fit");
    hipError_t e = ensure_dynamic_lds<gemm_x3_kernel<WTM>>(lds);
    if (e != hipSuccess) return e;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + XBN - 1) / XBN) * p.batches);
    hipLaunchKernelGGL(gemm_x3_kernel<WTM>, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

}  // namespace

bool gemm_x3_eligible(const GemmParams& p) {
    if (!p.W_x3 || !p.out32 || p.out16) return false;
    if ((p.K & 31) || (p.N & 3) || (p.ldo & 3) || (p.o_bs & 3)) return false;
    if (((p.lda * 4) & 15) || ((p.a_bs * 4) & 15)) return false;
    const uintptr_t al = (uintptr_t)p.A | (uintptr_t)p.W_x3 | (uintptr_t)p.out32 | (uintptr_t)p.residual | (uintptr_t)p.bias;
    if (al & 15) return false;
    return p.N >= 128;  // any M (ragged rows are clamped / masked): a row's arithmetic must not depend on the batch it sits in
}

hipError_t launch_gemm_x3(const GemmParams& p, hipStream_t stream) {
    // 256- or 192-row tiles, whichever leaves fewer idle CU-rounds (see gemm16.hip)
    const long nt = (p.N + 255) / 256;
    const long t256 = ((p.M + 255) / 256) * nt * p.batches, t192 = ((p.M + 191) / 192) * nt * p.batches;
    const long c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;
    return c192 * 11 < c256 * 10 ? x3_go<96>(p, stream) : x3_go<128>(p, stream);
}

// Host: fp32 (N, K) row-major -> pair-packed bf16 image of the same size: per row, per group of 8 k: 8 x hi, 8 x lo.
void pack_x3(const float* w, long N, long K, std::vector<uint16_t>& out) {
    auto bf = [](float f) -> uint16_t {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    out.assign((size_t)N * K * 2, 0);
    for (long r = 0; r < N; ++r)
        for (long g = 0; g < K / 8; ++g)
            for (int e = 0; e < 8; ++e) {
                const float x = w[r * K + g * 8 + e];
                const uint16_t h = bf(x);
                uint32_t hu = ((uint32_t)h) << 16;
                float hf;
                memcpy(&hf, &hu, 4);
                out[(size_t)r * K * 2 + g * 16 + e] = h;
                out[(size_t)r * K * 2 + g * 16 + 8 + e] = bf(x - hf);
            }
}

}  // namespace s3
