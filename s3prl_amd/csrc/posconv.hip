// posconv.hip — the convolutional relative-position embedding:
//   out = x + GELU( SamePad( Conv1d(D, D, k=128, padding=64, groups=16)(x) ) + bias )
// make_conv_pos + SamePad (wav2vec2_model.py:2937-2953,1797-1808; WavLM.py:539-552); the weight-norm
// (w = g * v / ||v||, dim=2) is folded at pack time.
//
// Per group this is a Toeplitz GEMM: out[t][co] = sum_{j<K} sum_{ci<Dg} x[t + j - K/2][ci] * w[co][ci][j].
// One workgroup = (batch b, group g, 64 output frames).  The (64 + K - 1) x Dg input window of the group is staged
// ONCE in LDS (the 128-frame halo is re-used by all 128 taps — the A operand of tap j is just the window shifted by
// j rows), the Dg x Dg weight slice of tap j streams through a double-buffered LDS tile shared by the 4 waves, and each
// wave accumulates 16 frames x Dg channels with v_mfma_f32_16x16x4_f32 (exact fp32).  Residual add, bias and erf-GELU
// are fused in the epilogue; x is read once and the (B,T,D) result written once.
#include "kernels.h"

namespace s3 {
namespace {

constexpr int PC_TM = 64;  // output frames per workgroup

template <int DG>
__global__ __launch_bounds__(256) void posconv_kernel(PosConvParams p) {
    constexpr int NT = DG / 16;       // 16-wide output-channel tiles
    constexpr int NCC = DG / 16;      // 16-deep input-channel chunks
    constexpr int RS = DG + 4;        // LDS row stride of the x window (floats)
    constexpr int WSZ = DG * DG;      // floats per tap
    constexpr int NWV = (WSZ / 4 + 255) / 256;  // float4 per thread per tap
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = p.K;
    const int rows = PC_TM + K - 1;
    float* xs = lds;
    float* wl = lds + ((rows * RS + 3) & ~3);

    const int b = blockIdx.z, g = blockIdx.y;
    const int t0 = blockIdx.x * PC_TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int pad = K / 2;

    const float* xg = p.x + (long)b * p.T * p.D + g * DG;
    for (int idx = tid; idx < rows * (DG / 4); idx += 256) {
        const int rr = idx / (DG / 4), c4 = idx % (DG / 4);
        const int ts = t0 + rr - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ts >= 0 && ts < p.T) v = *(const float4*)(xg + (long)ts * p.D + 4 * c4);
        *(float4*)(xs + rr * RS + 4 * c4) = v;
    }
    const float4* wg = (const float4*)(p.w + (long)g * K * WSZ);
    float4 wreg[NWV];
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
            if (e < WSZ / 4) *(float4*)(wl + buf * WSZ + 4 * e) = wreg[i];
        }
    };
    wload(0);
    wstore(0);
    __syncthreads();

    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* xrow = xs + (wave * 16 + l15) * RS + 4 * kk;
    for (int j = 0; j < K; ++j) {
        if (j + 1 < K) wload(j + 1);
        const float* wb = wl + (j & 1) * WSZ + l15 * 16 + 4 * kk;
        const float* xa = xrow + j * RS;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            const float4 a = *(const float4*)(xa + 16 * cc);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 w = *(const float4*)(wb + (cc * DG + n * 16) * 16);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[n], 0, 0, 0);
            }
        }
        if (j + 1 < K) wstore((j + 1) & 1);
        __syncthreads();
    }

    // D layout of 16x16: col = lane&15 (channel), row = 4*(lane>>4) + reg (frame)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = g * DG + n * 16 + l15;
        const float bias = p.bias[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + wave * 16 + 4 * kk + r;
            if (t < p.T) {
                const long o = ((long)b * p.T + t) * p.D + c;
                p.out[o] = p.x[o] + gelu_erf(acc[n][r] + bias);
            }
        }
    }
}

template <int DG>
hipError_t pc_launch(const PosConvParams& p, hipStream_t s) {
    const int rows = PC_TM + p.K - 1;
    const size_t lds = (size_t)(((rows * (DG + 4) + 3) & ~3) + 2 * DG * DG) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)posconv_kernel<DG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((p.T + PC_TM - 1) / PC_TM, p.G, p.B);
    hipLaunchKernelGGL(posconv_kernel<DG>, grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

}  // namespace

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
