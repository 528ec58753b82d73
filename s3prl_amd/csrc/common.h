// common.h — shared device helpers for the gfx950 kernels of libs3enc.
#pragma once
// cache-policy modifier of every global_load_lds (LDS-DMA) instruction of the 16-bit GEMM kernels: "" (default) = wave scope,
// " sc1" = agent scope (an experiment of round 6's third session, profiles/r06c_concurrent_forwards.md; make EXTRA='-DS3_GLDS_MOD="\" sc1\""')
#ifndef S3_GLDS_MOD
#define S3_GLDS_MOD ""
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

enum DType : int { F32 = 0, BF16 = 1, F16 = 2 };

constexpr int WAVE = 64;
constexpr float LN_EPS = 1e-5f;

// ---- 16-bit storage tags: data live in memory as raw uint16 -----------------------------------------
struct bf16_tag {};
struct f16_tag {};

// fp32 -> bf16, round to nearest even: one v_cvt_pk_bf16_f32 on gfx950 (the host packer h_bf16 rounds identically)
__device__ __forceinline__ u16 f32_to_bf16(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ __forceinline__ float bf16_to_f32(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ u16 f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(u16, h);
}
__device__ __forceinline__ float f16_to_f32(u16 h) { return (float)__builtin_bit_cast(_Float16, h); }

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    typedef float store_t;
    static __device__ __forceinline__ float to(float f) { return f; }
    static __device__ __forceinline__ float from(float f) { return f; }
};
template <> struct Cvt<bf16_tag> {
    typedef u16 store_t;
    static __device__ __forceinline__ u16 to(float f) { return f32_to_bf16(f); }
    // two values -> one packed dword (a in the low half): a single v_cvt_pk_bf16_f32
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
    }
    static __device__ __forceinline__ float from(u16 h) { return bf16_to_f32(h); }
};
template <> struct Cvt<f16_tag> {
    typedef u16 store_t;
    static __device__ __forceinline__ u16 to(float f) { return f32_to_f16(f); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
    }
    static __device__ __forceinline__ float from(u16 h) { return f16_to_f32(h); }
};

// Split-precision operands (gemm_x3.hip, posconv x3): 8 fp32 (two float4) -> bf16x8 hi and bf16x8 lo with
// x = hi + lo + r, |r| <= 2^-17 |x|  (hi = bf16(x) by v_cvt_pk_bf16_f32, lo = bf16(x - hi))
__device__ __forceinline__ void split8(const float4& x0, const float4& x1, uint4& hi, uint4& lo) {
    auto pair = [](float a, float b, unsigned& h, unsigned& l) {
        h = Cvt<bf16_tag>::pack2(a, b);
        const float ha = __uint_as_float(h << 16), hb = __uint_as_float(h & 0xffff0000u);
        l = Cvt<bf16_tag>::pack2(a - ha, b - hb);
    };
    pair(x0.x, x0.y, hi.x, lo.x);
    pair(x0.z, x0.w, hi.y, lo.y);
    pair(x1.x, x1.y, hi.z, lo.z);
    pair(x1.z, x1.w, hi.w, lo.w);
}

// One LayerNorm output element from the row's statistics: ((v - mu) * rs) * g + b with the last product fused — the ONE form both the
// row LayerNorm (norm.hip) and a GEMM epilogue that rebuilds a LayerNorm output from its input row (gemm16.hip, GemmParams::res_ln_*)
// evaluate, so that the two give the same bits.
__device__ __forceinline__ float ln_affine(float v, float mu, float rs, float g, float b) { return __builtin_fmaf((v - mu) * rs, g, b); }

// erf-GELU in fp32: nn.GELU() / F.gelu(x.float()) on the reference path.
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf-GELU with ONE transcendental, for the 16-bit operand modes and the split-precision modes (where libm erff would cost more
// VALU cycles than a K = 768 contraction costs MFMA cycles):
//     gelu(x) = max(x, 0) - |x|/2 * erfc(|x| / sqrt 2),        erfc(z) = 2 ^ (z * Q(z)),  Q of degree 7 on z in [0, 6]
// (Q fitted on [0, 6] to log2(erfc(z)) / z with the error weighted by erfc; beyond it Q keeps falling (Q <= -10, leading
// coefficient negative), so 2^(z Q) stays below 2^-60 and reaches exactly 0; z is clamped at 16 only so that x = +inf gives
// +inf instead of inf * 0.  gelu(-inf) and gelu(NaN) are NaN, as in 0.5 * x * (1 + erf(x / sqrt 2)).)
// erfc is computed directly, so there is no 1 + erf cancellation on the negative side and no reciprocal: 7 FMAs + v_exp_f32.
// Against an fp64 evaluation on N(0, s) inputs, s = 0.3 / 1 / 3: relative Frobenius error 7.0e-8 / 3.5e-8 / 2.2e-8 — the level of
// 0.5 * x * (1 + erff(x / sqrt 2)) itself evaluated in fp32 (4.0e-8 / 3.7e-8 / 3.0e-8); the A&S 7.1.26 form of rounds 1-2 (rcp +
// exp + degree 5) was at 1.3e-7 / 8.7e-8 / 5.2e-8 with 4.5 more issue slots per element.
#define S3_GELU_Q0 -1.627915263e+00f
#define S3_GELU_Q1 -9.183272123e-01f
#define S3_GELU_Q2 -1.488733739e-01f
#define S3_GELU_Q3 2.905271389e-02f
#define S3_GELU_Q4 -1.628119033e-03f
#define S3_GELU_Q5 -9.987915400e-04f
#define S3_GELU_Q6 3.023426980e-04f
#define S3_GELU_Q7 -2.878726809e-05f
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    const float z = fminf(ax * 0.70710678118654752440f, 16.0f);
    float q = fmaf(S3_GELU_Q7, z, S3_GELU_Q6);
    q = fmaf(q, z, S3_GELU_Q5);
    q = fmaf(q, z, S3_GELU_Q4);
    q = fmaf(q, z, S3_GELU_Q3);
    q = fmaf(q, z, S3_GELU_Q2);
    q = fmaf(q, z, S3_GELU_Q1);
    q = fmaf(q, z, S3_GELU_Q0);
    const float e = __builtin_amdgcn_exp2f(z * q);             // erfc(z)
    const float hz = z * 0.70710678118654752440f;               // |x| / 2 inside the fitted range
    return fmaf(-hz, e, (x + ax) * 0.5f);                       // (x + |x|) / 2 = max(x, 0), exact, NaN-propagating
}

// The same erf-GELU on a PAIR of values with packed fp32 VALU (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: one issue slot
// for two lanes' worth of work; the transcendental, |x| and the clamp stay per element): ~12.5 issue slots per element instead
// of 19.  For epilogues and elementwise kernels only — beside MFMAs packed fp32 is an anti-lever (MI355X_MICROARCH.md,
// per-instruction constants).  Bit-identical to gelu_fast per element (same operations, same order).
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f32x2 zu = ax * 0.70710678118654752440f;
    const f32x2 z = {__builtin_fminf(zu.x, 16.0f), __builtin_fminf(zu.y, 16.0f)};
    f32x2 q = __builtin_elementwise_fma(z, (f32x2){S3_GELU_Q7, S3_GELU_Q7}, (f32x2){S3_GELU_Q6, S3_GELU_Q6});
    q = __builtin_elementwise_fma(q, z, (f32x2){S3_GELU_Q5, S3_GELU_Q5});
    q = __builtin_elementwise_fma(q, z, (f32x2){S3_GELU_Q4, S3_GELU_Q4});
    q = __builtin_elementwise_fma(q, z, (f32x2){S3_GELU_Q3, S3_GELU_Q3});
    q = __builtin_elementwise_fma(q, z, (f32x2){S3_GELU_Q2, S3_GELU_Q2});
    q = __builtin_elementwise_fma(q, z, (f32x2){S3_GELU_Q1, S3_GELU_Q1});
    q = __builtin_elementwise_fma(q, z, (f32x2){S3_GELU_Q0, S3_GELU_Q0});
    const f32x2 a = z * q;
    const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2 hz = z * 0.70710678118654752440f;
    return __builtin_elementwise_fma(-hz, e, (x + ax) * 0.5f);
}
__device__ __forceinline__ void gelu_fast4(float4& v) {
    const f32x2 a = gelu_fast2((f32x2){v.x, v.y}), b = gelu_fast2((f32x2){v.z, v.w});
    v = make_float4(a.x, a.y, b.x, b.y);
}

// GELU of the compute mode: libm erff for the fp32 path, the one-transcendental form for the 16-bit operand modes
template <typename T> __device__ __forceinline__ float gelu_mode(float x) { return gelu_fast(x); }
template <> __device__ __forceinline__ float gelu_mode<float>(float x) { return gelu_erf(x); }
// four values at once: FAST = the packed one-transcendental form (16-bit operand modes, split-precision modes), else libm erff
template <bool FAST> __device__ __forceinline__ void gelu4(float& a, float& b, float& c, float& d) {
    if constexpr (FAST) {
        const f32x2 p = gelu_fast2((f32x2){a, b}), q = gelu_fast2((f32x2){c, d});
        a = p.x; b = p.y; c = q.x; d = q.y;
    } else {
        a = gelu_erf(a); b = gelu_erf(b); c = gelu_erf(c); d = gelu_erf(d);
    }
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane of the row; four DPP moves, no LDS
// (quad xor 1, quad xor 2, row_half_mirror, row_mirror: each step adds a disjoint partial)
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

// Wave-wide reductions with the result in every lane: the xor butterfly `v = v (op) __shfl_xor(v, o)`, o = 32, 16, 8, 4, 2, 1 —
// same pairing and same order, and (op) commutative, so bit-identical to it — WITHOUT the LDS: __shfl_xor lowers to
// ds_bpermute_b32 + s_waitcnt lgkmcnt(0), six dependent LDS round trips per reduction (two reductions per LayerNorm row;
// 12 on the critical path of every frame of the layer_norm-mode conv0).  xor_pair<O>(u, a, b) returns {own, partner} of the
// lane pair (l, l ^ O) in an order that is the same for both lanes of the pair:
//   O = 32 / 16: v_permlane32_swap / v_permlane16_swap of a register with its copy leave {lower's, lower's} and {upper's, upper's};
//   O = 8: DPP row_ror:8;  O = 4: row_shl:4 on banks 0 / 2 + row_shr:4 on banks 1 / 3;  O = 2 / 1: quad_perm.
template <int O> __device__ __forceinline__ void xor_pair(unsigned u, unsigned& a, unsigned& b) {
    if constexpr (O == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        a = r[0]; b = r[1];
    } else if constexpr (O == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        a = r[0]; b = r[1];
    } else if constexpr (O == 8) {
        a = u; b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, false);           // row_ror:8
    } else if constexpr (O == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)u, 0x104, 0xf, 0x5, false);                  // row_shl:4 -> lanes of banks 0, 2
        a = u; b = (unsigned)__builtin_amdgcn_update_dpp(t, (int)u, 0x114, 0xf, 0xa, false);           // row_shr:4 -> lanes of banks 1, 3
    } else if constexpr (O == 2) {
        a = u; b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, false);            // quad_perm [2,3,0,1]
    } else {
        a = u; b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, false);            // quad_perm [1,0,3,2]
    }
}
template <int O> __device__ __forceinline__ float xor_sum_f(float v) {
    unsigned a, b;
    xor_pair<O>(__float_as_uint(v), a, b);
    return __uint_as_float(a) + __uint_as_float(b);
}
template <int O> __device__ __forceinline__ float xor_max_f(float v) {
    unsigned a, b;
    xor_pair<O>(__float_as_uint(v), a, b);
    return fmaxf(__uint_as_float(a), __uint_as_float(b));
}
template <int O> __device__ __forceinline__ double xor_sum_d(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    unsigned alo, blo, ahi, bhi;
    xor_pair<O>((unsigned)u, alo, blo);
    xor_pair<O>((unsigned)(u >> 32), ahi, bhi);
    return __longlong_as_double((long long)(((unsigned long long)ahi << 32) | alo)) +
           __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = xor_sum_f<32>(v); v = xor_sum_f<16>(v); v = xor_sum_f<8>(v);
    v = xor_sum_f<4>(v); v = xor_sum_f<2>(v); v = xor_sum_f<1>(v);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v = xor_sum_d<32>(v); v = xor_sum_d<16>(v); v = xor_sum_d<8>(v);
    v = xor_sum_d<4>(v); v = xor_sum_d<2>(v); v = xor_sum_d<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = xor_max_f<32>(v); v = xor_max_f<16>(v); v = xor_max_f<8>(v);
    v = xor_max_f<4>(v); v = xor_max_f<2>(v); v = xor_max_f<1>(v);
    return v;
}

}  // namespace s3
