// gemmt.hip — the path's GEMM on a (256 | 192 | 128 | 64) x 128 output tile: the default kernel of the exact-fp32 mode
// (launch_gemm falls back to gemm.hip's 128x128 kernel for a ragged K or unaligned operands) and, for shapes with few tiles,
// of the split-precision mode S3ENC_F32X3.  One template over the arithmetic: a 64-byte K step is 16 fp32 = four
// v_mfma_f32_32x32x2_f32 per accumulator tile, or — x3 — one bf16 MFMA depth = three v_mfma_f32_32x32x16_bf16 on split
// operands; everything else — LDS image, swizzle, DMA, schedule — is shared.
//
//   out[b][m][n] = epilogue( sum_k A[b][m][k] * W[n][k] )        (conv1-6 as strided-row GEMMs, post_extract_proj, q|k|v,
//                                                                  out_proj, fc1, fc2 — wav2vec2_model.py:2910-2925,3306-3320)
//
// Why this shape (profiles/r02_mfma_peak.md, r02_gemm_yardstick.md): gemm_kernel<float> loses ~14 % of the matrix pipe and the
// loss scales with the L2 -> LDS staging bytes per MFMA; round 2's 256x256 tile halved the bytes but ran ONE lock-step workgroup
// per CU, so every stage boundary (wait, barrier, four M0-juggling DMA issues, fragment read latency) and every prologue /
// 256 KiB epilogue drained the pipe.  Here:
//   * 4 waves per workgroup, one per SIMD, each owning a (32 TM) x 64 accumulator block (TM = 4: 128 VGPRs, 193 in all;
//     TM = 3: 152; TM = 2: 108): two / three / four independent workgroups share a CU, so a SIMD always has other waves whose
//     MFMAs fill one's barrier, prologue, GELU / store epilogue and DMA waits (vmcnt is per wave and in order: a wave's own
//     epilogue stores never sit in front of its next DMA — it exits).
//   * one 64-byte K step = 8 TM MFMAs per wave in two halves; the fragments of the second half / of the next step's first half
//     are read while the previous half multiplies (two register sets), and the ONE wait + barrier of a step sits between the
//     halves with MFMAs ready on both sides of it: the pipe sees the barrier as one instruction slot, not as a restart.
//   * staging is LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... lds): per DMA one SGPR offset (k) and one VGPR
//     offset computed once per tile — no 64-bit address arithmetic in the loop, no M0 save / restore (nothing else uses M0).
//   * same 64-byte K steps, same lane -> k assignment, same MFMA (v_mfma_f32_32x32x2_f32) and the same per-accumulator k order
//     as gemm_kernel<float>: results are BIT-IDENTICAL to the 128x128 kernel whatever tile is chosen (tests/test_ops_gpu.py),
//     so a row's rounding does not depend on the batch it sits in (the shard == full-batch property of DESIGN §7).
//   * the tile height is chosen per GEMM (tile_efficiency below; measured in profiles/r03_gemm32_lab.md): 128 rows unless
//     another height divides the shape over the 256 CUs better (q|k|v of HuBERT-base 32 x 10 s: 192 rows = 5.9 tiles per CU)
//     or K is long enough for the 256-row tile's lower staging traffic to pay (K >= 6144); 64 rows when there are fewer
//     tiles than CUs.  Measured (same box, random operands): 128x128 kernel of round 2 -> this kernel, TFLOP/s: conv1 134.5
//     -> 145.6, q|k|v 128.7 -> 137.8, out_proj 121.7 -> 130.5, fc1 126.8 -> 135.5, fc2 130.5 -> 140.9, 8192^3 136.9 -> 151.1.
#include <type_traits>

#include "kernels.h"

namespace s3 {

namespace {

constexpr int BN = 128, ROWB = 64;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct x3_tag {};  // S3ENC_F32X3: fp32 operands in memory, three bf16 MFMAs per product on split operands (gemm_x3.hip's arithmetic)
template <typename T> struct TileElem { typedef typename Cvt<T>::store_t type; };
template <> struct TileElem<x3_tag> { typedef float type; };

template <typename T> struct TileMma;
template <> struct TileMma<float> {  // four dependent MFMAs (k order x, y, z, w — the order of gemm_kernel<float>)
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};
template <> struct TileMma<bf16_tag> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <typename T, int TM>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(GemmParams p) {
    typedef typename TileElem<T>::type store_t;
    constexpr int EB = sizeof(store_t);
    constexpr bool X3 = std::is_same<T, x3_tag>::value;
    constexpr int BM = 64 * TM;
    constexpr int PLANE = (BM + BN) * ROWB;  // one 64-byte K step of both operand tiles (24 / 20 / 16 KiB)
    constexpr int NPASS = (BM + BN) / 64;    // 1 KiB DMA pieces per wave per step (6 / 5 / 4); passes < APASS fetch A rows
    constexpr int APASS = BM / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware tile order (as gemm.hip): every XCD gets a contiguous range of the (batch, m-tile, n-tile) sequence, n fastest
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m_tiles = (p.M + BM - 1) / BM;
    int tile;
    {
        const int nwg = gridDim.x, wg = blockIdx.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wg & 7, loc = wg >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    }
    const int tn = tile % n_tiles;
    const int tmb = tile / n_tiles;
    const int tm = tmb % m_tiles, b = tmb / m_tiles;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = (p.K * EB) >> 6;  // launcher: K is a multiple of the 64-byte step

    // ---- LDS-DMA sources.  A wave instruction lands 64 x 16 B = 16 rows of one operand at a wave-uniform LDS base; wave w
    //      fetches rows 16 w + 64 i + lane / 4 of A (passes i < APASS) and of W.  The LDS image is lane-linear, so the
    //      XOR swizzle of the 16-byte slots is applied to the SOURCE slot each lane fetches (cdna_hip_programming.md rule 21).
    //      Rows past M / N are clamped to the last row (their results are never stored).
    i32x4 rsrc_a, rsrc_w;
    {
        const char* ab = (const char*)p.A + (long)b * p.a_bs * EB;
        const unsigned long ua = (unsigned long)ab, uw = (unsigned long)(X3 ? p.W_x3 : p.W);
        rsrc_a = (i32x4){(int)__builtin_amdgcn_readfirstlane((unsigned)ua), (int)__builtin_amdgcn_readfirstlane((unsigned)(ua >> 32)),
                         -1, 0x00020000};
        rsrc_w = (i32x4){(int)__builtin_amdgcn_readfirstlane((unsigned)uw), (int)__builtin_amdgcn_readfirstlane((unsigned)(uw >> 32)),
                         -1, 0x00020000};
    }
    const int src_slot = (lane & 3) ^ ((lane >> 4) & 3);  // logical slot of the row (row >> 2) & 3 == (lane >> 4) & 3
    unsigned voff[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        if (i < APASS) {
            int r = m0 + 16 * wave + 64 * i + (lane >> 2);
            r = r < p.M ? r : p.M - 1;
            voff[i] = (unsigned)r * (unsigned)(p.lda * EB) + src_slot * 16;
        } else {
            int r = n0 + 16 * wave + 64 * (i - APASS) + (lane >> 2);
            r = r < p.N ? r : p.N - 1;
            voff[i] = (unsigned)r * (unsigned)(p.K * EB) + src_slot * 16;
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    // one DMA: M0 = destination (written in the same statement; nothing else in this kernel reads M0), the k offset rides in
    // the scalar offset.  Completion is waited for by hand (vmcnt) — hipcc does not count inline-asm memory operations.
    auto dma = [&](unsigned vo, const i32x4& rs, unsigned dst, unsigned soff) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :
                     : "s"(dst), "v"(vo), "s"(rs), "s"(soff)
                     : "memory");
    };
    auto issue = [&](int kt, int slot) {
        const unsigned kb = (unsigned)kt * ROWB;
        const unsigned d0 = lds0 + slot * PLANE;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) dma(voff[i], i < APASS ? rsrc_a : rsrc_w, d0 + i * 4096, kb);
    };

    // ---- fragment addresses: row l31 of each 32-row block, 16-byte slot (half * 2 + q) ^ swizzle(row) ----
    const int swz = (l31 >> 2) & 3;
    const char* fa_base = smem + (wr * (TM * 32) + l31) * ROWB;
    const char* fb_base = smem + (BM + wc * 64 + l31) * ROWB;
    const int so0 = ((half * 2) ^ swz) << 4, so1 = ((half * 2 + 1) ^ swz) << 4;

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Frag {
        uint4 a[TM], b[2];
    };
    auto read = [&](Frag& f, int slot, int so) {
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *(const uint4*)(fa_base + slot * PLANE + i * 32 * ROWB + so);
#pragma unroll
        for (int j = 0; j < 2; ++j) f.b[j] = *(const uint4*)(fb_base + slot * PLANE + j * 32 * ROWB + so);
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nk > 1) issue(1, 1);
    if constexpr (!X3) {
        auto mma = [&](const Frag& f, int i, int j) { TileMma<T>::run(f.a[i], f.b[j], acc[i][j]); };
        auto mma_all = [&](const Frag& f) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(f, i, j);
        };
        // ---- K loop.  Step kt multiplies plane kt (LDS slot kt & 1):
        //        read F1 = (kt, second half) | MFMAs on F0 = (kt, first half) | wait: plane kt+1 landed (mine), my reads of
        //        plane kt done | barrier (everybody's) | first tile of F1 | DMA plane kt+2 into plane kt's slot | read F0 =
        //        (kt+1, first half) | rest of F1.                                                                        ----
        Frag f0, f1;
        read(f0, 0, so0);
        auto step = [&](int kt, int slot) {
            read(f1, slot, so1);
            mma_all(f0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(f1, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < nk) issue(kt + 2, slot);
            if (kt + 1 < nk) read(f0, slot ^ 1, so0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (i | j) mma(f1, i, j);
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            step(kt, 0);
            step(kt + 1, 1);
        }
        if (kt < nk) step(kt, 0);
    } else {
        // ---- split-precision K loop.  A 64-byte step is 16 k = ONE bf16 MFMA depth: this half-wave's 8 k-values are the
        //      two 16-byte slots 2 half, 2 half + 1 of a row — 8 fp32 of A (split into bf16 hi / lo on the registers:
        //      common.h split8) and, in the pair-packed weight image, 8 x hi then 8 x lo of W.  Per accumulator tile
        //      a_lo*w_hi + a_hi*w_lo + a_hi*w_hi (the order of gemm_x3.hip).  The whole next plane is read into a second
        //      register set right after the step's one barrier, under this step's remaining MFMAs.                    ----
        auto mma_row = [&](const Frag& lo_half, const Frag& hi_half, int i) {  // (slots 2 half | 2 half + 1) of plane
            uint4 ah, al;
            split8(__builtin_bit_cast(float4, lo_half.a[i]), __builtin_bit_cast(float4, hi_half.a[i]), ah, al);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                TileMma<bf16_tag>::run(al, lo_half.b[j], acc[i][j]);   // a_lo * w_hi   (slot 2g of W = hi)
                TileMma<bf16_tag>::run(ah, hi_half.b[j], acc[i][j]);   // a_hi * w_lo   (slot 2g + 1 = lo)
                TileMma<bf16_tag>::run(ah, lo_half.b[j], acc[i][j]);   // a_hi * w_hi
            }
        };
        Frag f0, f1, g0, g1;
        read(f0, 0, so0);
        read(f1, 0, so1);
        auto step = [&](int kt, int slot, Frag& c0, Frag& c1, Frag& n0_, Frag& n1_) {
            mma_row(c0, c1, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < nk) issue(kt + 2, slot);
            if (kt + 1 < nk) {
                read(n0_, slot ^ 1, so0);
                read(n1_, slot ^ 1, so1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < TM; ++i) mma_row(c0, c1, i);
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            step(kt, 0, f0, f1, g0, g1);
            step(kt + 1, 1, g0, g1, f0, f1);
        }
        if (kt < nk) step(kt, 0, f0, f1, g0, g1);
    }
    // the last step's barrier ordered every wave's fragment reads before this point: the planes are free for the epilogue

    // ---- epilogue: acc[i][j][r] is (row = wr*32 TM + i*32 + (r&3) + 8*(r>>2) + 4*half, col = wc*64 + j*32 + l31); each
    //      32 x 64 block goes through a wave-private LDS transpose so that bias / GELU / residual run on row-contiguous
    //      vectors and every global access is a 16-byte vector ----
    const int limit = p.row_limit ? p.row_limit[b] : p.M;
    const long ob = (long)b * p.o_bs;
    float* stg = (float*)(smem + wave * 8192);
    {
        // fp32 output (exact-fp32 and split-precision modes): 16 lanes x 4 columns per row (gemm.hip's vector epilogue)
        const int c4 = (lane & 15) * 4;
        const int n = n0 + wc * 64 + c4;
        const bool n_ok = n < p.N;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && n_ok) bias4 = *(const float4*)(p.bias + n);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = t * 4 + (lane >> 4);
                float4 v = *(const float4*)(stg + row * 64 + c4);
                const int m = m0 + wr * (TM * 32) + i * 32 + row;
                if (m < p.M && n_ok) {
                    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                    if (p.act) {
                        if (X3 || p.act == 2) {
                            gelu_fast4(v);  // the packed one-transcendental form (act 2: asked for in fp32 too)
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
                    *(float4*)(p.out32 + o) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

template <typename T, int TM>
hipError_t go(const GemmParams& p, hipStream_t stream) {
    constexpr int BM = 64 * TM;
    constexpr int plane2 = 2 * (BM + BN) * ROWB;
    constexpr int lds = plane2 > 4 * 8192 ? plane2 : 4 * 8192;  // 48 / 40 / 32 / 32 KiB: the workgroups per CU are VGPR-limited
    hipError_t e = ensure_dynamic_lds<gemm_tile_kernel<T, TM>>(lds);
    if (e != hipSuccess) return e;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batches);
    hipLaunchKernelGGL((gemm_tile_kernel<T, TM>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}
template <typename T>
hipError_t go_tm(int tm, const GemmParams& p, hipStream_t stream) {
    switch (tm) {
        case 4: return go<T, 4>(p, stream);
        case 3: return go<T, 3>(p, stream);
        case 2: return go<T, 2>(p, stream);
        case 1: return go<T, 1>(p, stream);
    }
    return hipErrorInvalidValue;
}

// Relative MFMA-time efficiency of a tile height on a shape (profiles/r03_gemm32_lab.md is the data behind the factors):
// useful rows / columns over what the tiles cover, times how evenly the tiles divide over the CUs (what bounds the GEMM is
// the CU that gets the most), times the measured steady-state factor of the height — on this part MORE co-resident workgroups
// (128 rows: four per CU, 192: three, 256: two) beat FEWER staged bytes per MFMA until K is very long: their prologues,
// GELU / store epilogues and barriers overlap, and a taller tile only pays once its K loop dwarfs them.
double tile_efficiency(const GemmParams& p, int tm, int cus) {
    const long bm = 64L * tm;
    const long mt = (p.M + bm - 1) / bm, nt = (p.N + BN - 1) / BN;
    const long tiles = mt * nt * p.batches;
    const double cover = ((double)p.M * p.N) / ((double)mt * bm * nt * BN);
    const long per_cu = (tiles + cus - 1) / cus;
    const double balance = (double)tiles / ((double)per_cu * cus);
    double steady = 1.0;  // tm == 2
    if (tm == 1) steady = 0.90;
    if (tm == 3) steady = 0.988;
    if (tm == 4) steady = p.K >= 6144 ? 1.01 : 0.965;
    return cover * balance * steady;
}

}  // namespace

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            cus = 256;
    }
    return cus;
}

int gemm_tile_pick(int mode, const GemmParams& p) {
    const int cus = device_cus();
    if (mode >= 2) return 6 - mode;  // 2 -> TM 4, 3 -> TM 3, 4 -> TM 2, 5 -> TM 1
    int best = 0;
    double best_e = 0.0;
    for (int tm = 2; tm <= 4; ++tm) {  // ties go to the smaller tile
        const double e = tile_efficiency(p, tm, cus);
        if (e > best_e + 1e-9) {
            best_e = e;
            best = tm;
        }
    }
    if (best_e < 0.5) {  // fewer tiles than CUs (a few short utterances): 64-row tiles double the workgroups
        const double e = tile_efficiency(p, 1, cus);
        if (e > best_e + 1e-9) best = 1;
    }
    return best;
}

// dtype F32, or 3 = S3ENC_F32X3 (fp32 operands, the pair-packed p.W_x3).  fp32 output only, vector-epilogue alignment, K a
// multiple of the 64-byte step, operands addressable with 32-bit offsets.  The 16-bit operand modes do NOT come here: the
// same template instantiated for bf16 measured 15-40 % behind gemm16.hip's lock-step 256x256 tile on every shape of the
// path, with or without a staggered start of the co-resident workgroups (profiles/r03_gemm_tile_lab.md) — at the 16-bit
// MFMA rate a 128-byte LDS fragment read per 32 cycles of MFMA is what bounds a 64-column wave tile.
bool gemm_tile_eligible(int dtype, const GemmParams& p) {
    if (dtype != F32 && dtype != 3) return false;
    const int mode = dtype == F32 ? tuning().gemm32_big : tuning().gemm_x3_tile;
    if (!mode || !p.out32 || p.out16) return false;
    if (dtype == 3 && !p.W_x3) return false;
    if ((p.K & 15) || (p.N & 3) || (p.ldo & 3) || (p.o_bs & 3) || ((p.lda * 4) & 15) || ((p.a_bs * 4) & 15)) return false;
    const uintptr_t al = (uintptr_t)p.A | (uintptr_t)(dtype == 3 ? p.W_x3 : p.W) | (uintptr_t)p.out32 | (uintptr_t)p.residual |
                         (uintptr_t)p.bias;
    if (al & 15) return false;
    if (p.M < 64 || p.N < 128) return false;
    const unsigned long a_span = ((unsigned long)(p.M - 1) * (unsigned long)p.lda + (unsigned long)p.K) * 4ul;
    const unsigned long w_span = (unsigned long)p.N * (unsigned long)p.K * 4ul;
    if (a_span >= (1ul << 32) - 64 || w_span >= (1ul << 32) - 64) return false;
    if (dtype == 3 && mode == 1) {
        // split-precision mode: the lock-step 256x256 tile of gemm_x3.hip wins wherever its tiles fill the chip (its wave
        // tile reads half the LDS bytes per MFMA); this kernel — bit-identical to it — takes the shapes with fewer 256x256
        // tiles than half the CUs (conv6 of a 32 x 10 s batch: 250 -> 288 TF; a single 10 s utterance's q|k|v: 39 -> 81 TF)
        const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.batches;
        if (t256 * 2 > device_cus()) return false;
    }
    return gemm_tile_pick(mode, p) != 0;
}

hipError_t launch_gemm_tile(int dtype, const GemmParams& p, hipStream_t stream) {
    const int tm = gemm_tile_pick(dtype == F32 ? tuning().gemm32_big : tuning().gemm_x3_tile, p);
    return dtype == 3 ? go_tm<x3_tag>(tm, p, stream) : go_tm<float>(tm, p, stream);
}

}  // namespace s3
