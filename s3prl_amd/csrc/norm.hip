// norm.hip — row LayerNorm over C (eps 1e-5, biased variance, two-pass in registers), optional erf-GELU.
// HBM-bound: one wave64 per row, float4 loads, statistics by wavefront shuffles, dual store
// (fp32 hidden-state tap / residual stream and/or the 16-bit operand of the next GEMM).
// Serves: LayerNorm(512) before post_extract_proj (hubert_model.py:482-483), encoder.layer_norm
// (wav2vec2_model.py:3049-3050,3069-3070), self_attn_layer_norm / final_layer_norm (:3274-3320) and the
// per-conv Fp32LayerNorm + GELU of layer_norm-mode extractors (:2887-2897).
#include "kernels.h"

namespace s3 {
namespace {

template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, long rows, int C, int act,
                                                        float* out32, void* out16) {
    typedef typename Cvt<T>::store_t store_t;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int nch = C >> 2;
    const float* xr = x + row * C;
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        v[i] = ch < nch ? *(const float4*)(xr + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float invC = 1.f / (float)C;
    const float mu = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) * invC + LN_EPS);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch >= nch) continue;
        const float4 g = *(const float4*)(gamma + 4 * ch);
        const float4 bt = *(const float4*)(beta + 4 * ch);
        float4 y;
        y.x = (v[i].x - mu) * rs * g.x + bt.x;
        y.y = (v[i].y - mu) * rs * g.y + bt.y;
        y.z = (v[i].z - mu) * rs * g.z + bt.z;
        y.w = (v[i].w - mu) * rs * g.w + bt.w;
        if (act) {
            y.x = gelu_mode<T>(y.x);
            y.y = gelu_mode<T>(y.y);
            y.z = gelu_mode<T>(y.z);
            y.w = gelu_mode<T>(y.w);
        }
        if (out32) *(float4*)(out32 + row * C + 4 * ch) = y;
        if (out16) {
            if constexpr (sizeof(store_t) == 4) {
                *(float4*)((float*)out16 + row * C + 4 * ch) = y;
            } else {
                ushort4 h;
                h.x = Cvt<T>::to(y.x);
                h.y = Cvt<T>::to(y.y);
                h.z = Cvt<T>::to(y.z);
                h.w = Cvt<T>::to(y.w);
                *(ushort4*)((u16*)out16 + row * C + 4 * ch) = h;
            }
        }
    }
}

template <typename T>
hipError_t ln_dispatch(const float* x, const float* gamma, const float* beta, long rows, int C, int act, float* out32,
                       void* out16, hipStream_t s) {
    const int per_lane = ((C >> 2) + 63) / 64;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define S3_LN(N) hipLaunchKernelGGL((layernorm_kernel<T, N>), grid, block, 0, s, x, gamma, beta, rows, C, act, out32, out16)
    if (per_lane <= 1) S3_LN(1);
    else if (per_lane == 2) S3_LN(2);
    else if (per_lane == 3) S3_LN(3);
    else if (per_lane == 4) S3_LN(4);
    else S3_LN(8);
#undef S3_LN
    return hipGetLastError();
}

}  // namespace

hipError_t launch_layernorm(int dtype, const float* x, const float* gamma, const float* beta, long rows, int C, int act,
                            float* out32, void* out16, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if ((C & 3) || C > 2048) return hipErrorInvalidValue;
    switch (dtype) {
        case F32: return ln_dispatch<float>(x, gamma, beta, rows, C, act, out32, out16, s);
        case BF16: return ln_dispatch<bf16_tag>(x, gamma, beta, rows, C, act, out32, out16, s);
        case F16: return ln_dispatch<f16_tag>(x, gamma, beta, rows, C, act, out32, out16, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace s3
