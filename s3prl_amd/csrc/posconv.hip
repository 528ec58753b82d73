// posconv.hip — the convolutional relative-position embedding:
//   out = x + GELU( SamePad( Conv1d(D, D, k=128, padding=64, groups=16)(x) ) + bias )
// make_conv_pos + SamePad (wav2vec2_model.py:2937-2953,1797-1808; WavLM.py:539-552); the weight-norm
// (w = g * v / ||v||, dim=2) is folded at pack time.
//
// Per group this is a Toeplitz GEMM: out[t][co] = sum_{j<K} sum_{ci<Dg} x[t + j - K/2][ci] * w[co][ci][j].
// One workgroup = (batch b, group g, 128 output frames).  The (128 + K - 1) x Dg input window of the group is staged
// ONCE in LDS (the halo is re-used by all 128 taps — the A operand of tap j is just the window shifted by j rows), the
// Dg x Dg weight slice of tap j streams through a double-buffered LDS tile shared by the 4 waves, and each wave
// accumulates 32 frames x Dg channels with v_mfma_f32_16x16x4_f32 (exact fp32): 2 x Dg/16 independent accumulator chains
// per wave (the 16x16x4 shape needs >= 6-8 in flight to approach its rate: profiles/r02_mfma_peak.md) and every weight
// fragment read from LDS feeds two frame tiles.  Residual add, bias and erf-GELU are fused in the epilogue; x is read
// once and the (B,T,D) result written once.  (Round 3: 64 -> 128 frames per workgroup = half the weight stream per output,
// and the XCD-aware work map below keeps each group's 1.2 MB weight slice in ONE L2.)
#include "kernels.h"

namespace s3 {
namespace {

constexpr int PC_TM = 128;  // output frames per workgroup (4 waves x 2 frame tiles of 16)
constexpr int PC_FT = 2;    // 16-frame tiles per wave

// XCD-aware work map (1-D grid; workgroup w runs on XCD w % 8, each with a private 4 MiB L2).  Every workgroup of a group
// streams that group's whole weight slice (Dg x Dg x K: 1.2 MB fp32 at HuBERT-base); with the plain (frame tile, group,
// batch) grid the workgroups of ONE group are spread over all eight XCDs and every L2 keeps re-fetching all 16 slices
// (round 2: 2.4 GB fetched per launch for 117 MB algorithmic, L2 hit 0.50).  Here XCD x owns the groups g = x (mod 8):
// two slices (2.4 MB) stay L2-resident while the XCD sweeps their (batch, frame tile) workgroups.  Used when G % 8 == 0
// (the released models: G = 16), else the plain order.
struct PcWork {
    int b, g, tile;
    bool live;
};
__device__ __forceinline__ PcWork pc_work(const PosConvParams& p, int ntiles) {
    PcWork w;
    const int wg = blockIdx.x;
    if (p.G & 7) {
        w.tile = wg % ntiles;
        const int r = wg / ntiles;
        w.g = r % p.G;
        w.b = r / p.G;
        w.live = w.b < p.B;
        return w;
    }
    const int per = ntiles * p.B, xcd = wg & 7, local = wg >> 3;
    w.g = xcd + 8 * (local / per);
    const int rem = local % per;
    w.b = rem / ntiles;
    w.tile = rem % ntiles;
    w.live = w.g < p.G;
    return w;
}

template <int DG>
__global__ __launch_bounds__(256) void posconv_kernel(PosConvParams p) {
    constexpr int NT = DG / 16;       // 16-wide output-channel tiles
    constexpr int NCC = DG / 16;      // 16-deep input-channel chunks
    constexpr int RS = DG + 8;        // LDS row stride of the x window (floats): +8 makes the float4 A reads conflict-free (+4: 2-way)
    constexpr int WSZ = DG * DG;      // floats per tap
    constexpr int NWV = (WSZ / 4 + 255) / 256;  // float4 per thread per tap
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = p.K;
    const int rows = PC_TM + K - 1;
    float* xs = lds;
    float* wl = lds + ((rows * RS + 3) & ~3);

    const PcWork wk = pc_work(p, (p.T + PC_TM - 1) / PC_TM);
    if (!wk.live) return;
    const int b = wk.b, g = wk.g;
    const int t0 = wk.tile * PC_TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int pad = p.pad >= 0 ? p.pad : K / 2;

    const float* xg = p.x + (long)b * p.T * p.D + g * DG;
    for (int idx = tid; idx < rows * (DG / 4); idx += 256) {
        const int rr = idx / (DG / 4), c4 = idx % (DG / 4);
        const int ts = t0 + rr - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ts >= 0 && ts < p.T) v = *(const float4*)(xg + (long)ts * p.D + 4 * c4);
        *(float4*)(xs + rr * RS + 4 * c4) = v;
    }
    const f32x4* wg = (const f32x4*)((const float*)p.w + (long)g * K * WSZ);
    f32x4 wreg[NWV];  // native vectors: an array of HIP float4 structs held across the loop is left in scratch memory
    auto wload = [&](int j) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = tid + 256 * i;
            if (e < WSZ / 4) wreg[i] = wg[(long)j * (WSZ / 4) + e];
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = tid + 256 * i;
            if (e < WSZ / 4) *(f32x4*)(wl + buf * WSZ + 4 * e) = wreg[i];
        }
    };
    wload(0);
    wstore(0);
    __syncthreads();

    f32x4 acc[PC_FT][NT];
#pragma unroll
    for (int f = 0; f < PC_FT; ++f)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[f][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* xrow = xs + (wave * (16 * PC_FT) + l15) * RS + 4 * kk;  // frame tile f: + 16 f rows
    for (int j = 0; j < K; ++j) {
        if (j + 1 < K) wload(j + 1);
        const float* wb = wl + (j & 1) * WSZ + l15 * 16 + 4 * (kk ^ pc_w_swizzle(l15));  // (the pack stored slot kk there)
        const float* xa = xrow + j * RS;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            float4 a[PC_FT];
#pragma unroll
            for (int f = 0; f < PC_FT; ++f) a[f] = *(const float4*)(xa + f * 16 * RS + 16 * cc);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 w = *(const float4*)(wb + (cc * DG + n * 16) * 16);
                // per accumulator the k order is x, y, z, w of chunk cc, chunks and taps ascending — as before: a frame's
                // result does not depend on the frame tile it sits in
#pragma unroll
                for (int f = 0; f < PC_FT; ++f) {
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].x, w.x, acc[f][n], 0, 0, 0);
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].y, w.y, acc[f][n], 0, 0, 0);
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].z, w.z, acc[f][n], 0, 0, 0);
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].w, w.w, acc[f][n], 0, 0, 0);
                }
            }
        }
        if (j + 1 < K) wstore((j + 1) & 1);
        __syncthreads();
    }

    // D layout of 16x16: col = lane&15 (channel), row = 4*(lane>>4) + reg (frame)
#pragma unroll
    for (int f = 0; f < PC_FT; ++f)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int c = g * DG + n * 16 + l15;
            const float bias = p.bias[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + wave * (16 * PC_FT) + 16 * f + 4 * kk + r;
                if (t < p.T) {
                    const long o = ((long)b * p.T + t) * p.D + c;
                    p.out[o] = p.plain ? acc[f][n][r] + bias : p.x[o] + gelu_erf(acc[f][n][r] + bias);
                }
            }
        }
}

template <int DG>
hipError_t pc_launch(const PosConvParams& p, hipStream_t s) {
    const int rows = PC_TM + p.K - 1;
    const size_t lds = (size_t)(((rows * (DG + 8) + 3) & ~3) + 2 * DG * DG) * sizeof(float);
    hipError_t e = ensure_dynamic_lds<posconv_kernel<DG>>((int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(((p.T + PC_TM - 1) / PC_TM) * p.G * p.B));  // 1-D: pc_work maps it XCD-aware (G % 8 == 0: same count)
    hipLaunchKernelGGL(posconv_kernel<DG>, grid, dim3(256), lds, s, p);
    return hipGetLastError();
}


// ---- 16-bit operand modes ------------------------------------------------------------------------------------------
// Same contraction on v_mfma_f32_16x16x32_{bf16,f16}.  Per group the Toeplitz GEMM is an IMPLICIT GEMM with
// overlapping rows, exactly like the strided convs of gemm.hip:  out[t][co] = sum_k A[t][k] * W[co][k] with the k axis
// flattened as k = j*Dg + ci, W packed [G][Dg][K*Dg] and A[t][k] = window[(t + j)][ci] = the 16-bit window read as one
// flat array at element t*Dg + k — so a 32-deep MFMA step may straddle taps and Dg = 48 needs no padding.
// One workgroup = (batch, group, 256 output frames); wave w owns frames [64w, 64w+64) x all Dg channels
// (4 x Dg/16 accumulator tiles: every W fragment read from LDS feeds 4 MFMAs).  The fp32 input window is converted
// once and kept in LDS chunk-major ([Dg/8][rows] 16-byte units: the 16 frames of an A fragment are 16 consecutive
// slots -> conflict-free for any Dg); W streams through a double-buffered, XOR-swizzled LDS tile of 128 k per step
// (global -> VGPR prefetch -> ds_write, one barrier per 128 k).  Bias, GELU and the fp32 residual are fused.
constexpr int P16_TM = 256;   // output frames per workgroup
constexpr int P16_KC = 128;   // k per W stage (16 x 16-byte chunks per row)

template <typename T> struct Mma16x16;
template <> struct Mma16x16<bf16_tag> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16x16<f16_tag> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// X3: split-precision variant of the fp32x3 mode (S3ENC_F32X3): the fp32 window is split into a bf16 hi and a bf16 lo
// plane, W arrives as a hi and a lo image ([G][Dg][K*Dg] each, lo after hi), every product is
// a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (~1e-5 relative, see gemm_x3.hip); 128 frames per workgroup (two planes of LDS).
template <typename T, int DG, bool X3>
__global__ __launch_bounds__(256) void posconv16_kernel(PosConvParams p) {
    constexpr int NT = DG / 16;  // 16-wide output-channel tiles
    constexpr int CH = DG / 8;   // 16-byte chunks per frame
    constexpr int NLW = DG * 16 / 256;  // 16-byte W pieces per thread per stage and plane
    constexpr int WBUF = DG * 256;      // bytes per W stage and plane
    constexpr int NS = X3 ? 2 : 1;      // operand planes (hi, lo)
    constexpr int TMF = X3 ? 128 : P16_TM;  // output frames per workgroup
    constexpr int MT = TMF / 64;            // 16-frame M tiles per wave
    extern __shared__ __attribute__((aligned(16))) char lds16[];
    const int K = p.K;
    const int ROWS = (TMF + K - 1 + 15) & ~15;  // window rows, padded so chunk planes start on the same bank
    const size_t WIN = (size_t)CH * ROWS * 16;  // bytes of one window plane
    char* win = lds16;
    char* wl = lds16 + NS * WIN;  // [plane][buffer][WBUF]

    const PcWork wk = pc_work(p, (p.T + TMF - 1) / TMF);
    if (!wk.live) return;
    const int b = wk.b, g = wk.g;
    const int t0 = wk.tile * TMF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kgrp = lane >> 4;
    const int pad = p.pad >= 0 ? p.pad : K / 2;
    const long Ktot = (long)K * DG;

    // ---- W stage loader: piece e = (row n, chunk c); LDS slot of chunk c in row n is c ^ (n & 15).
    //      (macros, not lambdas: a by-reference capture of the register array ends up in scratch memory) ----
    const char* wg = (const char*)p.w + (long)g * DG * Ktot * 2;
    const long plane_bytes = (long)p.G * DG * Ktot * 2;  // X3: the lo image follows the hi image
    u32x4 wreg[NS][NLW];  // native vectors: an array of HIP uint4 structs held across the loop is left in scratch memory
#define P16_WLOAD(kc_)                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < NLW; ++i_) {  \
        const int e_ = tid + 256 * i_, n_ = e_ >> 4, c_ = e_ & 15;                                          \
        wreg[s_][i_] = *(const u32x4*)(wg + s_ * plane_bytes + ((long)n_ * Ktot + (long)(kc_) * P16_KC + c_ * 8) * 2); \
    }
#define P16_WSTORE(buf_)                                                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < NLW; ++i_) {  \
        const int e_ = tid + 256 * i_, n_ = e_ >> 4, c_ = e_ & 15;                                          \
        *(u32x4*)(wl + (s_ * 2 + (buf_)) * WBUF + (n_ * 16 + (c_ ^ (n_ & 15))) * 16) = wreg[s_][i_];       \
    }
    P16_WLOAD(0)

    // ---- input window: fp32 -> 16-bit (X3: hi and lo planes), chunk-major ----
    const float* xg = p.x + (long)b * p.T * p.D + g * DG;
    const int rows = TMF + K - 1;
    for (int idx = tid; idx < ROWS * CH; idx += 256) {
        const int f = idx % ROWS, cc = idx / ROWS;
        const int ts = t0 + f - pad;
        uint4 h = make_uint4(0, 0, 0, 0), l = make_uint4(0, 0, 0, 0);
        if (f < rows && ts >= 0 && ts < p.T) {
            const float4 v0 = *(const float4*)(xg + (long)ts * p.D + cc * 8);
            const float4 v1 = *(const float4*)(xg + (long)ts * p.D + cc * 8 + 4);
            if constexpr (X3) {
                split8(v0, v1, h, l);
            } else {
                h.x = Cvt<T>::pack2(v0.x, v0.y);
                h.y = Cvt<T>::pack2(v0.z, v0.w);
                h.z = Cvt<T>::pack2(v1.x, v1.y);
                h.w = Cvt<T>::pack2(v1.z, v1.w);
            }
        }
        *(uint4*)(win + ((size_t)cc * ROWS + f) * 16) = h;
        if constexpr (X3) *(uint4*)(win + WIN + ((size_t)cc * ROWS + f) * 16) = l;
    }
    P16_WSTORE(0)
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nkc = (int)(Ktot / P16_KC);  // = DG
    const int arow = wave * (TMF / 4) + l15;
    for (int kc = 0; kc < nkc; ++kc) {
        if (kc + 1 < nkc) { P16_WLOAD(kc + 1) }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int idx = kc * 16 + s * 4 + kgrp;  // 16-byte chunk index along k
            const int j = idx / CH, cc = idx - j * CH;
            const size_t aoff = ((size_t)cc * ROWS + arow + j) * 16;
            uint4 fa[NS][MT], fb[NS][NT];
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) {
                const char* wb = wl + (pl * 2 + (kc & 1)) * WBUF;
#pragma unroll
                for (int n = 0; n < NT; ++n) fb[pl][n] = *(const uint4*)(wb + ((n * 16 + l15) * 16 + ((s * 4 + kgrp) ^ l15)) * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[pl][m] = *(const uint4*)(win + pl * WIN + aoff + m * 256);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if constexpr (X3) {
                        acc[m][n] = Mma16x16<T>::run(fa[1][m], fb[0][n], acc[m][n]);
                        acc[m][n] = Mma16x16<T>::run(fa[0][m], fb[1][n], acc[m][n]);
                    }
                    acc[m][n] = Mma16x16<T>::run(fa[0][m], fb[0][n], acc[m][n]);
                }
        }
        if (kc + 1 < nkc) { P16_WSTORE((kc + 1) & 1) }
        __syncthreads();
    }
#undef P16_WLOAD
#undef P16_WSTORE

    // D layout of 16x16: col = lane&15 (channel), row = 4*(lane>>4) + reg (frame)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = g * DG + n * 16 + l15;
        const float bias = p.bias[c];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + wave * (TMF / 4) + m * 16 + 4 * kgrp + r;
                if (t < p.T) {
                    const long o = ((long)b * p.T + t) * p.D + c;
                    const float y = acc[m][n][r] + bias;
                    p.out[o] = p.plain ? y : p.x[o] + gelu_fast(y);
                }
            }
    }
}

template <typename T, int DG, bool X3>
hipError_t pc16_launch(const PosConvParams& p, hipStream_t s) {
    constexpr int NS = X3 ? 2 : 1, TMF = X3 ? 128 : P16_TM;
    const int ROWS = (TMF + p.K - 1 + 15) & ~15;
    const size_t lds = (size_t)NS * (DG / 8) * ROWS * 16 + (size_t)NS * 2 * DG * 256;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = ensure_dynamic_lds<posconv16_kernel<T, DG, X3>>((int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(((p.T + TMF - 1) / TMF) * p.G * p.B));  // 1-D: pc_work maps it XCD-aware
    hipLaunchKernelGGL((posconv16_kernel<T, DG, X3>), grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

template <typename T, bool X3 = false>
hipError_t pc16_dispatch(const PosConvParams& p, int dg, hipStream_t s) {
    switch (dg) {
        case 32: return pc16_launch<T, 32, X3>(p, s);
        case 48: return pc16_launch<T, 48, X3>(p, s);
        case 64: return pc16_launch<T, 64, X3>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// 16-bit operand modes: p.w is the 16-bit [G][Dg][K*Dg] pack (k = tap*Dg + ci), x / out stay fp32.
hipError_t launch_posconv16(int dtype, const PosConvParams& p, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return hipSuccess;
    const int dg = p.D / p.G;
    if (dg * p.G != p.D || (p.K & 1) || ((long)p.K * dg) % P16_KC) return hipErrorInvalidValue;
    if (dtype == BF16) return pc16_dispatch<bf16_tag>(p, dg, s);
    if (dtype == F16) return pc16_dispatch<f16_tag>(p, dg, s);
    if (dtype == 3) return pc16_dispatch<bf16_tag, true>(p, dg, s);  // S3ENC_F32X3: p.w = bf16 hi image followed by the lo image
    return hipErrorInvalidValue;
}

hipError_t launch_posconv(const PosConvParams& p, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return hipSuccess;
    const int dg = p.D / p.G;
    if (dg * p.G != p.D) return hipErrorInvalidValue;
    switch (dg) {
        case 32: return pc_launch<32>(p, s);
        case 48: return pc_launch<48>(p, s);
        case 64: return pc_launch<64>(p, s);
    }
    return hipErrorInvalidValue;  // D/groups must be 32, 48 or 64 (tiny / base / large)
}

}  // namespace s3
