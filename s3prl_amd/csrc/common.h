// common.h — shared device helpers for the gfx950 kernels of libs3enc.
#pragma once
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

// erf-GELU in fp32: nn.GELU() / F.gelu(x.float()) on the reference path.
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf-GELU with |erf error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26: one v_exp, one v_rcp, a degree-5 Horner) for the
// 16-bit operand modes, where libm erff would cost more VALU cycles than a K = 768 contraction costs MFMA cycles.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
    const float erf_abs = fmaf(-poly, e, 1.0f);
    const float erf = __builtin_copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf);
}

// The same erf-GELU on a PAIR of values with packed fp32 VALU (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: one issue slot
// for two lanes' worth of work; the two transcendentals and the sign transfer stay per element): ~11 issue slots per
// element instead of 19.  For epilogues and elementwise kernels only — beside MFMAs packed fp32 is an anti-lever
// (MI355X_MICROARCH.md, per-instruction constants).  Bit-identical to gelu_fast per element (same operations, same order).
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f32x2 z = ax * 0.70710678118654752440f;
    const f32x2 d = __builtin_elementwise_fma(z, (f32x2){0.3275911f, 0.3275911f}, (f32x2){1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2 poly = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){0.254829592f, 0.254829592f});
    poly = poly * t;
    const f32x2 zz = (z * -1.44269504088896340736f) * z;
    const f32x2 e = {__builtin_amdgcn_exp2f(zz.x), __builtin_amdgcn_exp2f(zz.y)};
    const f32x2 erf_abs = __builtin_elementwise_fma(-poly, e, (f32x2){1.0f, 1.0f});
    const f32x2 erf = {__builtin_copysignf(erf_abs.x, x.x), __builtin_copysignf(erf_abs.y, x.y)};
    return (x * 0.5f) * (erf + 1.0f);
}
__device__ __forceinline__ void gelu_fast4(float4& v) {
    const f32x2 a = gelu_fast2((f32x2){v.x, v.y}), b = gelu_fast2((f32x2){v.z, v.w});
    v = make_float4(a.x, a.y, b.x, b.y);
}

// GELU of the compute mode: libm-exact erf for the fp32 path, the fast erf for the 16-bit operand modes
template <typename T> __device__ __forceinline__ float gelu_mode(float x) { return gelu_fast(x); }
template <> __device__ __forceinline__ float gelu_mode<float>(float x) { return gelu_erf(x); }
// four values at once: FAST = the packed 1.5e-7 erf (16-bit operand modes and the split-precision mode), else libm erff
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace s3
