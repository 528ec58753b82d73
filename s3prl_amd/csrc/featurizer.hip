// featurizer.hip — the consumer of the hidden_states: the trainable weighted sum over layers of
// s3prl.nn.Featurizer._weighted_sum (nn/upstream.py:312-328) / upstream.interfaces.Featurizer._weighted_sum
// (upstream/interfaces.py:221-249):
//     out = sum_l softmax(w)_l * [F.layer_norm(h_l, (D,)) if normalize else h_l]
// HBM-bound: every selected layer is read exactly once (L * rows * D * 4 B) and the (rows, D) result written once —
// the reference materialises torch.stack(...) (a second copy of all layers), optionally its layer-normed copy, and the
// weighted product before reducing.  One wavefront per (b, t) row, float4 loads, layer-norm statistics by wavefront
// shuffles.  The backward kernel produces the only gradient the frozen-upstream setting needs, d out / d softmax(w):
//     g_l = sum_{rows, D} grad_out * hn_l       (deterministic two-stage reduction; softmax backward stays in torch)
#include "kernels.h"

namespace s3 {
namespace {

constexpr int WS_MAXV = 8;  // float4 per lane: D <= 64 * 4 * 8 = 2048

struct WsWeights {
    float w[S3_WS_MAX_LAYERS];
};

template <bool NORM>
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* hs, long layer_stride, int L, WsWeights wt, long rows,
                                                           int D, float* out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D / 256;         // full float4 rounds per lane
    const int rem = (D % 256) / 4;  // lanes < rem hold one more float4
    float4 acc[WS_MAXV];
#pragma unroll
    for (int i = 0; i < WS_MAXV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
        const float w = wt.w[l];
        if (w == 0.f) continue;  // unselected layer (wave-uniform)
        const float4* src = (const float4*)(hs + (long)l * layer_stride + row * D);
        float4 v[WS_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WS_MAXV; ++i) {
            const bool on = i < nv || (i == nv && lane < rem);
            v[i] = on ? src[i * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        float mean = 0.f, rstd = 1.f;
        if (NORM) {
            mean = wave_sum(s) / (float)D;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < WS_MAXV; ++i) {
                const bool on = i < nv || (i == nv && lane < rem);
                if (on) {
                    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                    q += a * a + b * b + c * c + d * d;
                }
            }
            rstd = rsqrtf(wave_sum(q) / (float)D + LN_EPS);
        }
        const float a = w * rstd, c0 = -mean * a;
#pragma unroll
        for (int i = 0; i < WS_MAXV; ++i) {
            acc[i].x += fmaf(v[i].x, a, c0);
            acc[i].y += fmaf(v[i].y, a, c0);
            acc[i].z += fmaf(v[i].z, a, c0);
            acc[i].w += fmaf(v[i].w, a, c0);
        }
    }
    float4* dst = (float4*)(out + row * D);
#pragma unroll
    for (int i = 0; i < WS_MAXV; ++i)
        if (i < nv || (i == nv && lane < rem)) dst[i * 64 + lane] = acc[i];
}

// partial[block][l] = sum over the block's rows of <grad_out_row, hn_l_row>
template <bool NORM>
__global__ __launch_bounds__(256) void weighted_sum_bwd_kernel(const float* hs, long layer_stride, int L, long rows, int D,
                                                               const float* gout, int rows_per_block, double* partial) {
    __shared__ double red[4][S3_WS_MAX_LAYERS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D / 256, rem = (D % 256) / 4;
    double accl[S3_WS_MAX_LAYERS];
    for (int l = 0; l < L; ++l) accl[l] = 0.0;
    const long r0 = (long)blockIdx.x * rows_per_block;
    for (long row = r0 + wave; row < r0 + rows_per_block && row < rows; row += 4) {
        float4 g[WS_MAXV];
        const float4* gs = (const float4*)(gout + row * D);
#pragma unroll
        for (int i = 0; i < WS_MAXV; ++i)
            g[i] = (i < nv || (i == nv && lane < rem)) ? gs[i * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < L; ++l) {
            const float4* src = (const float4*)(hs + (long)l * layer_stride + row * D);
            float4 v[WS_MAXV];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < WS_MAXV; ++i) {
                const bool on = i < nv || (i == nv && lane < rem);
                v[i] = on ? src[i * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                s += v[i].x + v[i].y + v[i].z + v[i].w;
            }
            float mean = 0.f, rstd = 1.f;
            if (NORM) {
                mean = wave_sum(s) / (float)D;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < WS_MAXV; ++i) {
                    const bool on = i < nv || (i == nv && lane < rem);
                    if (on) {
                        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                        q += a * a + b * b + c * c + d * d;
                    }
                }
                rstd = rsqrtf(wave_sum(q) / (float)D + LN_EPS);
            }
            float dot = 0.f, gsum = 0.f;
#pragma unroll
            for (int i = 0; i < WS_MAXV; ++i) {
                dot += g[i].x * v[i].x + g[i].y * v[i].y + g[i].z * v[i].z + g[i].w * v[i].w;
                gsum += g[i].x + g[i].y + g[i].z + g[i].w;
            }
            // <g, (v - mean) * rstd> = rstd * (<g, v> - mean * sum g)
            const float part = NORM ? rstd * (dot - mean * gsum) : dot;
            accl[l] += (double)wave_sum(part);
        }
    }
    if (lane == 0)
        for (int l = 0; l < L; ++l) red[wave][l] = accl[l];
    __syncthreads();
    if (threadIdx.x < L) partial[(long)blockIdx.x * L + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void weighted_sum_bwd_final(const double* partial, int nblocks, int L, float* grad_w) {
    const int l = threadIdx.x;
    if (l >= L) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(long)b * L + l];
    grad_w[l] = (float)s;
}

}  // namespace

hipError_t launch_weighted_sum(const float* hs, long layer_stride, int L, const float* w_host, int normalize, long rows, int D,
                               float* out, hipStream_t st) {
    if (L <= 0 || L > S3_WS_MAX_LAYERS || D <= 0 || (D & 3) || D > 64 * 4 * WS_MAXV || (layer_stride & 3)) return hipErrorInvalidValue;
    if (rows <= 0) return hipSuccess;
    WsWeights wt{};
    for (int l = 0; l < L; ++l) wt.w[l] = w_host[l];
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (normalize)
        hipLaunchKernelGGL(weighted_sum_kernel<true>, grid, dim3(256), 0, st, hs, layer_stride, L, wt, rows, D, out);
    else
        hipLaunchKernelGGL(weighted_sum_kernel<false>, grid, dim3(256), 0, st, hs, layer_stride, L, wt, rows, D, out);
    return hipGetLastError();
}

int weighted_sum_bwd_blocks(long rows) {
    long nb = (rows + 63) / 64;
    return (int)(nb > 2048 ? 2048 : nb);
}

hipError_t launch_weighted_sum_bwd(const float* hs, long layer_stride, int L, int normalize, long rows, int D,
                                   const float* grad_out, double* partial, float* grad_w, hipStream_t st) {
    if (L <= 0 || L > S3_WS_MAX_LAYERS || D <= 0 || (D & 3) || D > 64 * 4 * WS_MAXV || (layer_stride & 3)) return hipErrorInvalidValue;
    const int nb = weighted_sum_bwd_blocks(rows);
    const int rpb = (int)((rows + nb - 1) / nb);
    if (normalize)
        hipLaunchKernelGGL(weighted_sum_bwd_kernel<true>, dim3(nb), dim3(256), 0, st, hs, layer_stride, L, rows, D, grad_out, rpb, partial);
    else
        hipLaunchKernelGGL(weighted_sum_bwd_kernel<false>, dim3(nb), dim3(256), 0, st, hs, layer_stride, L, rows, D, grad_out, rpb, partial);
    hipLaunchKernelGGL(weighted_sum_bwd_final, dim3(1), dim3(64), 0, st, partial, nb, L, grad_w);
    return hipGetLastError();
}

}  // namespace s3
