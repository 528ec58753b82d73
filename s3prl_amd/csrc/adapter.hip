// adapter.hip — the memory-bound passes of multires-HuBERT's conv adapters (upstream/multires_hubert/hubert_model.py:970-1266)
// and the emission of a block's state at the finest frame rate (multires_hubert/expert.py:26-27,93-101).
//
// A conv adapter stage is   y = GELU(GroupNorm(1, D)(conv(x)))  ->  (y[:n] + skip[:n]) * sqrt(0.4)  [-> (+ highway) * sqrt(0.4)].
// The convolutions themselves (Conv1d stride s / ConvTranspose1d stride s over all D channels) are GEMMs over a
// zero-bordered frame buffer (engine.hip packs the weights for that); what is left is HBM-bound row work:
//   pad_copy      (B, T, D) fp32 [+ a second term] -> the zero-bordered operand buffer (fp32 and / or 16-bit)
//   group1_stats  per-utterance sum / sum of squares of the conv output (GroupNorm with ONE group: all frames x channels)
//   apply         normalise + affine + GELU + the skip / highway terms + scale, zero the padded frames, write the next
//                 stage's zero-bordered operand or the next block's input
//   emit_up       a block's state -> its slot of the caller's slab, every frame repeated `factor` times, cut to T_out
// Every kernel: one workgroup per output row, float4 per lane, 16-byte accesses only.
#include "kernels.h"

namespace s3 {
namespace {

template <typename T>
__device__ __forceinline__ void store_row4(float* o32, void* o16, long off, const float4& v) {
    if (o32) *(float4*)(o32 + off) = v;
    if constexpr (sizeof(typename Cvt<T>::store_t) == 2) {
        if (o16) {
            uint2 h;
            h.x = Cvt<T>::pack2(v.x, v.y);
            h.y = Cvt<T>::pack2(v.z, v.w);
            *(uint2*)((u16*)o16 + off) = h;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pad_copy_kernel(PadCopyParams p) {
    const int row = blockIdx.x, b = blockIdx.y;
    const int t = row - p.lead;
    const bool data = t >= 0 && t < p.rows && (!p.zero_from || t < p.zero_from[b]);
    const long off = ((long)b * p.total + row) * p.D;
    const float* a = p.a + (long)b * p.a_bs + (long)t * p.D;
    const float* c = p.b ? p.b + (long)b * p.b_bs + (long)t * p.D : nullptr;
    for (int ch = threadIdx.x; ch < (p.D >> 2); ch += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (data) {
            v = *(const float4*)(a + 4 * ch);
            if (c) {
                const float4 w = *(const float4*)(c + 4 * ch);
                v = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
            }
        }
        store_row4<T>(p.out32, p.out16, off + 4 * ch, v);
    }
}

// block (blockIdx.x of GS_BLOCKS, utterance blockIdx.y): partial[b][blk] = {sum, sum of squares} of its slice, in a fixed
// order (no atomics: the statistics are bit-reproducible run to run)
__global__ __launch_bounds__(256) void group1_stats_kernel(const float* x, long bs, long count4, double* partial) {
    const int b = blockIdx.y;
    const float4* xb = (const float4*)(x + (long)b * bs);
    const long per = (count4 + GS_BLOCKS - 1) / GS_BLOCKS;
    const long lo = (long)blockIdx.x * per, hi = lo + per < count4 ? lo + per : count4;
    float s = 0.f, q = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float4 v = xb[i];
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
    __shared__ double sh[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[2 * w] = ds;
        sh[2 * w + 1] = dq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = partial + ((long)b * GS_BLOCKS + blockIdx.x) * 2;
        o[0] = (sh[0] + sh[2]) + (sh[4] + sh[6]);
        o[1] = (sh[1] + sh[3]) + (sh[5] + sh[7]);
    }
}

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void adapter_apply_kernel(AdapterApplyParams p) {
    const int row = blockIdx.x, b = blockIdx.y;
    const int t = row - p.lead;
    const long off = ((long)b * p.total + row) * p.D;
    const bool data = t >= 0 && t < p.rows && (!p.zero_from || t < p.zero_from[b]);
    if (!data) {  // border rows of the operand geometry, rows past the kept length, padded frames
        for (int ch = threadIdx.x; ch < (p.D >> 2); ch += 256)
            store_row4<T>(p.out32, p.out16, off + 4 * ch, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    // GroupNorm(1, D) statistics of utterance b from the GS_BLOCKS partials (Fp32GroupNorm: biased variance, eps 1e-5)
    __shared__ float sh_mu, sh_rs;
    if (threadIdx.x < 64) {
        const double* pp = p.partial + ((long)b * GS_BLOCKS + threadIdx.x) * 2;
        const double s = wave_sum_d(pp[0]), q = wave_sum_d(pp[1]);
        if (threadIdx.x == 0) {
            const double mu = s / p.count;
            double var = q / p.count - mu * mu;
            var = var > 0.0 ? var : 0.0;
            sh_mu = (float)mu;
            sh_rs = (float)(1.0 / sqrt(var + (double)LN_EPS));
        }
    }
    __syncthreads();
    const float mu = sh_mu, rs = sh_rs;
    const float* cv = p.conv + (long)b * p.conv_bs + (long)t * p.D;
    const float* r1 = p.r1 + (long)b * p.r1_bs + (long)(((long)t * p.r1_mul) / p.r1_div) * p.D;
    const float* r2 = p.r2 ? p.r2 + (long)b * p.r2_bs + (long)(((long)t * p.r2_mul) / p.r2_div) * p.D : nullptr;
    for (int ch = threadIdx.x; ch < (p.D >> 2); ch += 256) {
        const float4 v = *(const float4*)(cv + 4 * ch);
        const float4 g = *(const float4*)(p.gamma + 4 * ch), be = *(const float4*)(p.beta + 4 * ch);
        float a0 = fmaf((v.x - mu) * rs, g.x, be.x), a1 = fmaf((v.y - mu) * rs, g.y, be.y);
        float a2 = fmaf((v.z - mu) * rs, g.z, be.z), a3 = fmaf((v.w - mu) * rs, g.w, be.w);
        gelu4<FAST>(a0, a1, a2, a3);
        const float4 x1 = *(const float4*)(r1 + 4 * ch);
        float4 y = make_float4((a0 + x1.x) * p.scale, (a1 + x1.y) * p.scale, (a2 + x1.z) * p.scale, (a3 + x1.w) * p.scale);
        if (r2) {
            const float4 x2 = *(const float4*)(r2 + 4 * ch);
            y = make_float4((y.x + x2.x) * p.scale, (y.y + x2.y) * p.scale, (y.z + x2.z) * p.scale, (y.w + x2.w) * p.scale);
        }
        store_row4<T>(p.out32, p.out16, off + 4 * ch, y);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void emit_up_kernel(const float* x, long x_bs, int factor, int rows_out, int D, float* out32,
                                                      void* out16) {
    const int t = blockIdx.x, b = blockIdx.y;
    const float* src = x + (long)b * x_bs + (long)(t / factor) * D;
    const long off = ((long)b * rows_out + t) * D;
    for (int ch = threadIdx.x; ch < (D >> 2); ch += 256) store_row4<T>(out32, out16, off + 4 * ch, *(const float4*)(src + 4 * ch));
}

// Featurizer term of a block's state at the finest frame rate, without writing the state: acc[b][t] (+)= w * s[b][t / factor]
// (norm: w * layer_norm(s row), no affine, eps 1e-5) — the arithmetic of emit_kernel / LnAcc in norm.hip.  A wave per row.
template <int NCH>
__global__ __launch_bounds__(256) void emit_up_acc_kernel(const float* x, long x_bs, int factor, int rows_out, long rows, int C,
                                                          LnAcc fa) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int nch = C >> 2;
    const long b = row / rows_out;
    const int t = (int)(row - b * rows_out);
    const float* xr = x + b * x_bs + (long)(t / factor) * C;
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        v[i] = ch < nch ? *(const float4*)(xr + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    float a = fa.w, c0 = 0.f;
    if (fa.norm) {
        const float invC = 1.f / (float)C;
        const float mu = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                const float a0 = v[i].x - mu, b0 = v[i].y - mu, c1 = v[i].z - mu, d0 = v[i].w - mu;
                q += (a0 * a0 + b0 * b0) + (c1 * c1 + d0 * d0);
            }
        }
        a = fa.w * rsqrtf(wave_sum(q) * invC + LN_EPS);
        c0 = -mu * a;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch >= nch) continue;
        float4* dst = (float4*)(fa.acc + row * C + 4 * ch);
        float4 tt = fa.init ? make_float4(0.f, 0.f, 0.f, 0.f) : *dst;
        tt.x += fmaf(v[i].x, a, c0);
        tt.y += fmaf(v[i].y, a, c0);
        tt.z += fmaf(v[i].z, a, c0);
        tt.w += fmaf(v[i].w, a, c0);
        *dst = tt;
    }
}

}  // namespace

hipError_t launch_emit_upsampled_acc(const float* x, long x_bs, int factor, int B, int rows_out, int D, const LnAcc& fa,
                                     hipStream_t s) {
    if (B <= 0 || rows_out <= 0 || !fa.acc) return hipSuccess;
    if ((D & 3) || D > 2048 || factor < 1) return hipErrorInvalidValue;
    const long rows = (long)B * rows_out;
    const int per_lane = ((D >> 2) + 63) / 64;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define S3_EU(N) hipLaunchKernelGGL((emit_up_acc_kernel<N>), grid, block, 0, s, x, x_bs, factor, rows_out, rows, D, fa)
    if (per_lane <= 1) S3_EU(1);
    else if (per_lane == 2) S3_EU(2);
    else if (per_lane == 3) S3_EU(3);
    else if (per_lane == 4) S3_EU(4);
    else S3_EU(8);
#undef S3_EU
    return hipGetLastError();
}

hipError_t launch_pad_copy(int dtype, const PadCopyParams& p, hipStream_t s) {
    if (p.B <= 0 || p.total <= 0) return hipSuccess;
    if ((p.D & 3) || (!p.out32 && !p.out16)) return hipErrorInvalidValue;
    dim3 grid(p.total, p.B), block(256);
    switch (dtype) {
        case F32: hipLaunchKernelGGL(pad_copy_kernel<float>, grid, block, 0, s, p); break;
        case BF16: hipLaunchKernelGGL(pad_copy_kernel<bf16_tag>, grid, block, 0, s, p); break;
        case F16: hipLaunchKernelGGL(pad_copy_kernel<f16_tag>, grid, block, 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_group1_stats(const float* x, long bs, long count, int B, double* partial, hipStream_t s) {
    if (B <= 0 || count <= 0) return hipSuccess;
    if ((count & 3) || (bs & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(group1_stats_kernel, dim3(GS_BLOCKS, B), dim3(256), 0, s, x, bs, count >> 2, partial);
    return hipGetLastError();
}

hipError_t launch_adapter_apply(int dtype, const AdapterApplyParams& p, hipStream_t s) {
    if (p.B <= 0 || p.total <= 0) return hipSuccess;
    if ((p.D & 3) || (!p.out32 && !p.out16) || p.r1_div <= 0 || (p.r2 && p.r2_div <= 0) || p.count <= 0) return hipErrorInvalidValue;
    dim3 grid(p.total, p.B), block(256);
    const bool fast = p.fast_gelu != 0 || tuning().gelu32 == 1;
    switch (dtype) {
        case F32:
            if (fast) hipLaunchKernelGGL((adapter_apply_kernel<float, true>), grid, block, 0, s, p);
            else hipLaunchKernelGGL((adapter_apply_kernel<float, false>), grid, block, 0, s, p);
            break;
        case BF16: hipLaunchKernelGGL((adapter_apply_kernel<bf16_tag, true>), grid, block, 0, s, p); break;
        case F16: hipLaunchKernelGGL((adapter_apply_kernel<f16_tag, true>), grid, block, 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_emit_upsampled(int dtype, const float* x, long x_bs, int factor, int B, int rows_out, int D, float* out32,
                                 void* out16, hipStream_t s) {
    if (B <= 0 || rows_out <= 0) return hipSuccess;
    if ((D & 3) || factor < 1 || (!out32 && !out16)) return hipErrorInvalidValue;
    dim3 grid(rows_out, B), block(256);
    switch (dtype) {
        case F32: hipLaunchKernelGGL(emit_up_kernel<float>, grid, block, 0, s, x, x_bs, factor, rows_out, D, out32, out16); break;
        case BF16: hipLaunchKernelGGL(emit_up_kernel<bf16_tag>, grid, block, 0, s, x, x_bs, factor, rows_out, D, out32, out16); break;
        case F16: hipLaunchKernelGGL(emit_up_kernel<f16_tag>, grid, block, 0, s, x, x_bs, factor, rows_out, D, out32, out16); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace s3
