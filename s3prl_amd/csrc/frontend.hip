// frontend.hip — raw 16 kHz PCM -> first conv layer activations.  HBM/VALU-bound.
//
//  * wav_norm_stats : per-utterance mean / rstd for task_cfg.normalize  (F.layer_norm(wav, wav.shape),
//                     hubert/expert.py:57-58) — applied on load by the consumers, never materialised.
//  * gn_stats       : Fp32GroupNorm(C, C) statistics of conv0's output (wav2vec2_model.py:2902,1841-1853)
//                     WITHOUT computing conv0: conv0 is linear in the waveform, so with
//                     S[j] = sum_t x[s*t+j] and R[j][j'] = sum_t x[s*t+j] x[s*t+j'] (k0 + k0^2 numbers per
//                     utterance, accumulated in fp64 over all L0 frames incl. the zero padding):
//                       mean_c = w_c.S / L0,   E[y^2]_c = w_c^T R w_c / L0.
//                     One pass over 4 B/sample of PCM instead of a pass over the 512-channel activation.
//  * conv0          : Conv1d(1, C, k=10, s=5) + {GroupNorm affine | LayerNorm over C} + erf-GELU, one wave per
//                     output frame, lanes across channels (coalesced 1 KiB row stores, LayerNorm statistics by
//                     wavefront shuffles), the PCM window of a 64-frame tile staged once in LDS and broadcast.
//                     Output is channel-last (B, L0, C) in the compute dtype, which makes conv1..6 plain
//                     strided-row GEMMs (gemm.hip).
// Padding (pad_sequence + wav mask, hubert/expert.py:60-66) is never built: reads past lens[b] return 0.
#include "kernels.h"

namespace s3 {

namespace {

constexpr int STAT_CHUNK = 16384;  // samples per block of wav_norm_stats
constexpr int GN_FRAMES = 4096;    // frames per block of gn_stats
constexpr int ROWLEN = 1 + STAT_K0_MAX;

__device__ __forceinline__ float load_wav(const float* w, long len, long i, float mean, float rstd) {
    return i < len ? (w[i] - mean) * rstd : 0.f;
}

__device__ __forceinline__ double block_sum_d(double v, double* red /*[4]*/) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void wav_sum_kernel(WavTable w, double* partial, int chunks) {
    __shared__ double red[4];
    const int b = blockIdx.y, ch = blockIdx.x;
    const float* x = w.ptrs[b];
    const long len = w.lens[b];
    const long beg = (long)ch * STAT_CHUNK;
    long end = beg + STAT_CHUNK;
    end = end < len ? end : len;
    double s = 0, s2 = 0;
    for (long i = beg + threadIdx.x; i < end; i += 256) {
        const double v = x[i];
        s += v;
        s2 += v * v;
    }
    s = block_sum_d(s, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        partial[((long)b * chunks + ch) * 2 + 0] = s;
        partial[((long)b * chunks + ch) * 2 + 1] = s2;
    }
}

__global__ void wav_norm_final_kernel(WavTable w, const double* partial, int chunks, int normalize, double eps, float2* norm) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= w.B) return;
    if (!normalize) {
        norm[b] = make_float2(0.f, 1.f);
        return;
    }
    double s = 0, s2 = 0;
    for (int c = 0; c < chunks; ++c) {  // fixed order: run-to-run deterministic
        s += partial[((long)b * chunks + c) * 2 + 0];
        s2 += partial[((long)b * chunks + c) * 2 + 1];
    }
    const double n = (double)w.lens[b];
    const double mean = s / n;
    double var = s2 / n - mean * mean;  // biased, like F.layer_norm
    var = var > 0 ? var : 0;
    norm[b] = make_float2((float)mean, (float)(1.0 / sqrt(var + eps)));
}

// partial[b][chunk][j][0] = sum_t x[s0 t + j];  [1+jj] = sum_t x[s0 t + j] x[s0 t + jj]
__global__ __launch_bounds__(256) void gn_lag_kernel(WavTable w, const float2* norm, int k0, int s0, long L0, double* partial,
                                                     int chunks) {
    __shared__ double red[4];
    const int ch = blockIdx.x, b = blockIdx.y, j = blockIdx.z;
    const float* x = w.ptrs[b];
    const long len = w.lens[b];
    const float mean = norm[b].x, rstd = norm[b].y;
    const long t_beg = (long)ch * GN_FRAMES;
    long t_end = t_beg + GN_FRAMES;
    t_end = t_end < L0 ? t_end : L0;
    double acc[ROWLEN];
#pragma unroll
    for (int e = 0; e < ROWLEN; ++e) acc[e] = 0;
    for (long t = t_beg + threadIdx.x; t < t_end; t += 256) {
        const long base = t * s0;
        const double xj = load_wav(x, len, base + j, mean, rstd);
        acc[0] += xj;
#pragma unroll
        for (int jj = 0; jj < STAT_K0_MAX; ++jj)
            if (jj < k0) acc[1 + jj] += xj * (double)load_wav(x, len, base + jj, mean, rstd);
    }
    double* dst = partial + (((long)b * chunks + ch) * k0 + j) * ROWLEN;
#pragma unroll
    for (int e = 0; e < ROWLEN; ++e) {
        const double v = block_sum_d(acc[e], red);
        if (threadIdx.x == 0) dst[e] = v;
    }
}

// The same numbers from ONE block per (chunk, utterance) (round 6, second session).  gn_lag_kernel above runs k0 blocks per chunk, each
// re-reading the chunk's PCM at a 20-byte lane stride (11 scattered loads per frame) and reducing its 17 accumulators one
// __syncthreads pair at a time: 95.7 us per forward under rocprofv3 for 20 MB of PCM (profiles/r06_kernel_stats_bf16.md).  Here the window of
// 1024 frames is staged in LDS with coalesced loads and normalised once, a thread keeps S[K0] and the UPPER triangle of R in registers
// (x_j x_jj is an exact fp64 product of two floats, so R[jj][j] is the same sum in the same order as R[j][jj]: mirrored on the way out)
// and all wave sums cross the block in one exchange.  Same frames per thread in the same order, same butterfly, same four-wave sum:
// bit-identical to gn_lag_kernel (tests/test_ops_gpu.py::test_gn_stats_one_block_form_is_bit_identical).
constexpr int GN_SUB = 1024;  // frames staged at a time (a divisor of GN_FRAMES, a multiple of the block)
template <int K0>
__global__ __launch_bounds__(256) void gn_lag_all_kernel(WavTable w, const float2* norm, int s0, long L0, double* partial, int chunks) {
    constexpr int NR = K0 * (K0 + 1) / 2, NACC = K0 + NR;
    __shared__ float win[(GN_SUB - 1) * 8 + K0];  // stride <= 8 (launcher)
    __shared__ double red[4][NACC];
    const int ch = blockIdx.x, b = blockIdx.y;
    const float* x = w.ptrs[b];
    const long len = w.lens[b];
    const float mean = norm[b].x, rstd = norm[b].y;
    const long t_beg = (long)ch * GN_FRAMES;
    long t_end = t_beg + GN_FRAMES;
    t_end = t_end < L0 ? t_end : L0;
    double S[K0], R[NR];
#pragma unroll
    for (int j = 0; j < K0; ++j) S[j] = 0;
#pragma unroll
    for (int e = 0; e < NR; ++e) R[e] = 0;
    for (long u_beg = t_beg; u_beg < t_end; u_beg += GN_SUB) {
        const int nfr = (int)((t_end - u_beg) < GN_SUB ? (t_end - u_beg) : GN_SUB);
        const int nwin = (nfr - 1) * s0 + K0;
        __syncthreads();  // (the previous window has been consumed)
        // eight loads in flight per thread (a plain loop compiles to one load + s_waitcnt vmcnt(0) per element: 20 serial round trips
        // per window)
        const long pos0 = u_beg * s0;
        for (int i0 = threadIdx.x; i0 < nwin; i0 += 8 * 256) {
            float raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                raw[u] = (i < nwin && pos0 + i < len) ? x[pos0 + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < nwin) win[i] = pos0 + i < len ? (raw[u] - mean) * rstd : 0.f;  // (= load_wav)
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nfr; t += 256) {  // (frames t_beg + tid + 256 i, i ascending: gn_lag_kernel's order)
            double xv[K0];
#pragma unroll
            for (int j = 0; j < K0; ++j) xv[j] = (double)win[t * s0 + j];
            int e = 0;
#pragma unroll
            for (int j = 0; j < K0; ++j) {
                S[j] += xv[j];
#pragma unroll
                for (int jj = j; jj < K0; ++jj) R[e++] += xv[j] * xv[jj];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < K0; ++j) {
        const double v = wave_sum_d(S[j]);
        if (lane == 0) red[wave][j] = v;
    }
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        const double v = wave_sum_d(R[e]);
        if (lane == 0) red[wave][K0 + e] = v;
    }
    __syncthreads();
    // thread (j, jj): partial[b][chunk][j][0] = S[j], [1 + jj] = R[min][max]
    for (int o = threadIdx.x; o < K0 * (1 + K0); o += 256) {
        const int j = o / (1 + K0), c = o % (1 + K0);
        int a;
        if (c == 0) a = j;
        else {
            const int jj = c - 1, lo = j < jj ? j : jj, hi = j < jj ? jj : j;
            a = K0 + lo * K0 - lo * (lo - 1) / 2 + (hi - lo);
        }
        partial[(((long)b * chunks + ch) * K0 + j) * ROWLEN + c] = ((red[0][a] + red[1][a]) + red[2][a]) + red[3][a];
    }
}

__global__ __launch_bounds__(256) void gn_final_kernel(const double* partial, int chunks, int k0, long L0, const float* w0,
                                                       const float* gamma, const float* beta, int C, float2* gn) {
    __shared__ double sums[STAT_K0_MAX * ROWLEN];
    const int b = blockIdx.x;  // (blockIdx.y: a slab of 256 channels; every slab rebuilds the 110 sums — cheaper than a second launch)
    for (int e = threadIdx.x; e < k0 * ROWLEN; e += 256) {
        double s = 0;
        for (int c = 0; c < chunks; ++c) s += partial[((long)b * chunks + c) * k0 * ROWLEN + e];
        sums[e] = s;
    }
    __syncthreads();
    const double inv = 1.0 / (double)L0;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += 256 * gridDim.y) {
        double m = 0, e2 = 0;
        double wd[STAT_K0_MAX];  // (the channel's taps once, in registers: the loop below re-read each of them k0 times from memory)
#pragma unroll
        for (int j = 0; j < STAT_K0_MAX; ++j) wd[j] = j < k0 ? (double)w0[c * k0 + j] : 0.0;
#pragma unroll
        for (int j = 0; j < STAT_K0_MAX; ++j) {
            if (j < k0) {
                const double wj = wd[j];
                m += wj * sums[j * ROWLEN];
                double r = 0;
#pragma unroll
                for (int jj = 0; jj < STAT_K0_MAX; ++jj)
                    if (jj < k0) r += wd[jj] * sums[j * ROWLEN + 1 + jj];
                e2 += wj * r;
            }
        }
        m *= inv;
        e2 *= inv;
        double var = e2 - m * m;  // conv bias shifts the mean only, so it cancels in (y - mean)
        var = var > 0 ? var : 0;
        const double scale = (double)gamma[c] / sqrt(var + (double)LN_EPS);
        gn[(long)b * C + c] = make_float2((float)scale, (float)((double)beta[c] - m * scale));
    }
}

// ---- conv0 -------------------------------------------------------------------------------------------------
constexpr int C0_FT = 64;  // frames per block

// CS = waves that share one frame, each owning a contiguous C / CS slice of the channels (GroupNorm mode only: no
// reduction across channels).  CS = 2 for C = 512 halves the per-lane weight registers (149 -> 81 VGPRs, 3 -> 5-6 waves
// per SIMD), which is what lets the GELU arithmetic of one wave overlap the 1 KiB row stores of another.
// FAST: packed fp32 arithmetic for the taps (v_pk_fma_f32 on channel pairs) and the packed one-transcendental GELU (common.h) — the
// 16-bit operand modes and the split-precision modes (this kernel is VALU-bound there: 40 FMAs + ~50 GELU slots per 4
// outputs), and the fp32 mode by default (tuning key gelu32 = 0: scalar FMAs in the same tap order and libm erff).
template <typename T, int NG, int K0, int CS = 1, bool FAST = false>
__global__ __launch_bounds__(256) void conv0_kernel(Conv0Params p) {
    typedef typename Cvt<T>::store_t store_t;
    // 16-bit output with two channel groups per lane: the lane owns 8 CONSECUTIVE channels (groups g = 0, 1 are channels
    // 8*lane + 4*g ..) so that a row is written with one 16-byte store per lane — 8-byte stores are issue-bound at
    // ~2.4 TB/s here, 16-byte ones reach the fp32 variant's 3.6+ TB/s
    constexpr bool WIDE = sizeof(store_t) == 2 && NG == 2 && CS == 1;
    auto chan0 = [&](int lane_, int g, int coff_) { return WIDE ? coff_ + 8 * lane_ + 4 * g : coff_ + 4 * (lane_ + 64 * g); };
    __shared__ float xs[(C0_FT - 1) * 8 + STAT_K0_MAX];  // stride <= 8 supported
    const int b = blockIdx.y;
    const long t0 = (long)blockIdx.x * C0_FT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = p.wav.ptrs[b];
    const long len = p.wav.lens[b];
    const float mean = p.norm[b].x, rstd = p.norm[b].y;
    const int nwin = (C0_FT - 1) * p.s0 + K0;
    for (int i = threadIdx.x; i < nwin; i += 256) xs[i] = load_wav(x, len, t0 * p.s0 + i, mean, rstd);

    // per-lane constants: channels coff + 4*(lane+64g) .. +3
    const int coff = (wave % CS) * (NG * 256);
    float w[NG][4][K0];
    float a0[NG][4], a1[NG][4];  // GN: scale, shift;  LN: gamma, beta
    float cb[NG][4];
    bool act[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c0 = chan0(lane, g, coff);
        act[g] = c0 < p.C;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = act[g] ? c0 + u : 0;
#pragma unroll
            for (int j = 0; j < K0; ++j) w[g][u][j] = act[g] ? p.w0[c * K0 + j] : 0.f;
            cb[g][u] = (act[g] && p.bias) ? p.bias[c] : 0.f;
            if (p.gn) {
                const float2 ss = p.gn[(long)b * p.C + c];
                a0[g][u] = ss.x;
                a1[g][u] = ss.y;
            } else {
                a0[g][u] = p.ln_g[c];
                a1[g][u] = p.ln_b[c];
            }
        }
    }
    __syncthreads();

    const float invC = 1.f / (float)p.C;
    // one output frame: conv taps -> {GroupNorm affine | LayerNorm over C} -> GELU, into v[g][0..3] (this lane's channels)
    auto frame = [&](int f, float (&v)[NG][4]) {
        float xv[K0];
#pragma unroll
        for (int j = 0; j < K0; ++j) xv[j] = xs[f * p.s0 + j];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if constexpr (FAST) {  // same per-channel tap order, two channels per instruction
                f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < K0; ++j) {
                    const f32x2 xx = {xv[j], xv[j]};
                    s01 = __builtin_elementwise_fma((f32x2){w[g][0][j], w[g][1][j]}, xx, s01);
                    s23 = __builtin_elementwise_fma((f32x2){w[g][2][j], w[g][3][j]}, xx, s23);
                }
                v[g][0] = s01.x; v[g][1] = s01.y; v[g][2] = s23.x; v[g][3] = s23.y;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < K0; ++j) s = fmaf(w[g][u][j], xv[j], s);
                    v[g][u] = s;
                }
            }
        }
        if (p.gn) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[g][u] = fmaf(v[g][u], a0[g][u], a1[g][u]);
                gelu4<FAST>(v[g][0], v[g][1], v[g][2], v[g][3]);
            }
        } else {
            // Fp32LayerNorm over the C channels of this frame (wav2vec2_model.py:2887-2897), two-pass
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[g][u] += cb[g][u];
                    s += act[g] ? v[g][u] : 0.f;
                }
            const float mu = wave_sum(s) * invC;
            float q = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = v[g][u] - mu;
                    q += act[g] ? d * d : 0.f;
                }
            const float rs = rsqrtf(wave_sum(q) * invC + LN_EPS);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[g][u] = (v[g][u] - mu) * rs * a0[g][u] + a1[g][u];
                gelu4<FAST>(v[g][0], v[g][1], v[g][2], v[g][3]);
            }
        }
    };
    constexpr int FSTEP = 4 / CS;  // frames between two iterations of one wave
    // (tried for the 16-bit GroupNorm variant and dropped: two frames per iteration with a DPP quad exchange so that every
    //  lane issues ONE 16-byte store per two frames instead of two 8-byte ones — same 0.42-0.48 ms: the kernel is bound by
    //  the two transcendentals of each GELU, not by store issue; packed fp32 taps did not move it either)
    for (int f = wave / CS; f < C0_FT; f += FSTEP) {
        const long t = t0 + f;
        if (t >= p.L0) break;
        float v[NG][4];
        frame(f, v);
        store_t* o = (store_t*)p.out + ((long)b * p.L0 + t) * p.C;
        if constexpr (WIDE) {
            if (act[0])  // C % 8 == 0 on this path (launcher): both groups are in or out together
                *(uint4*)(o + chan0(lane, 0, coff)) =
                    make_uint4(Cvt<T>::pack2(v[0][0], v[0][1]), Cvt<T>::pack2(v[0][2], v[0][3]),
                               Cvt<T>::pack2(v[1][0], v[1][1]), Cvt<T>::pack2(v[1][2], v[1][3]));
            continue;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (!act[g]) continue;
            const int c0 = chan0(lane, g, coff);
            if constexpr (sizeof(store_t) == 4) {
                if (p.nt) __builtin_nontemporal_store((f32x4){v[g][0], v[g][1], v[g][2], v[g][3]}, (f32x4*)(o + c0));
                else *(float4*)(o + c0) = make_float4(v[g][0], v[g][1], v[g][2], v[g][3]);
            } else {
                ushort4 h;
                h.x = Cvt<T>::to(v[g][0]);
                h.y = Cvt<T>::to(v[g][1]);
                h.z = Cvt<T>::to(v[g][2]);
                h.w = Cvt<T>::to(v[g][3]);
                *(ushort4*)(o + c0) = h;
            }
        }
    }
}

template <typename T, bool FAST>
hipError_t conv0_dispatch(const Conv0Params& p, hipStream_t s) {
    dim3 grid((unsigned)((p.L0 + C0_FT - 1) / C0_FT), p.wav.B);
    const int ng = (p.C + 255) / 256;
    if (ng <= 1)
        hipLaunchKernelGGL((conv0_kernel<T, 1, 10, 1, FAST>), grid, dim3(256), 0, s, p);
    // (16-bit GroupNorm extractors also go through the two-waves-per-frame variant with 8-byte stores: the one-wave,
    //  8-channels-per-lane variant with 16-byte row stores needs 134 VGPRs and measured 0.71 ms against 0.42 ms)
    else if (ng == 2 && p.gn)  // GroupNorm extractor (base models): two waves per frame, 256 channels each
        hipLaunchKernelGGL((conv0_kernel<T, 1, 10, 2, FAST>), grid, dim3(256), 0, s, p);
    else if (ng == 2)
        hipLaunchKernelGGL((conv0_kernel<T, 2, 10, 1, FAST>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv0_kernel<T, 4, 10, 1, FAST>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace

size_t stats_partial_elems(int B, long n_max) {
    const size_t c1 = (size_t)((n_max + STAT_CHUNK - 1) / STAT_CHUNK) * 2;
    const size_t c2 = (size_t)((n_max / 1 + GN_FRAMES - 1) / GN_FRAMES) * STAT_K0_MAX * ROWLEN;
    return (size_t)B * (c1 > c2 ? c1 : c2);
}

hipError_t launch_wav_norm_stats(const WavTable& w, int normalize, double* partial, float2* norm, hipStream_t s, float eps) {
    const int chunks = (int)((w.n_max + STAT_CHUNK - 1) / STAT_CHUNK);
    if (normalize) hipLaunchKernelGGL(wav_sum_kernel, dim3(chunks, w.B), dim3(256), 0, s, w, partial, chunks);
    hipLaunchKernelGGL(wav_norm_final_kernel, dim3((w.B + 63) / 64), dim3(64), 0, s, w, partial, chunks, normalize, (double)(eps > 0.f ? eps : LN_EPS), norm);
    return hipGetLastError();
}

hipError_t launch_gn_stats(const WavTable& w, const float2* norm, const float* w0, const float* gamma, const float* beta,
                           int C, int k0, int s0, long L0, double* partial, double* /*sums*/, float2* gn, hipStream_t s) {
    if (k0 > STAT_K0_MAX) return hipErrorInvalidValue;
    const int chunks = (int)((L0 + GN_FRAMES - 1) / GN_FRAMES);
    if (k0 == 10 && s0 >= 1 && s0 <= 8 && tuning().gn_lag_one_block)
        hipLaunchKernelGGL(gn_lag_all_kernel<10>, dim3(chunks, w.B), dim3(256), 0, s, w, norm, s0, L0, partial, chunks);
    else
        hipLaunchKernelGGL(gn_lag_kernel, dim3(chunks, w.B, k0), dim3(256), 0, s, w, norm, k0, s0, L0, partial, chunks);
    hipLaunchKernelGGL(gn_final_kernel, dim3(w.B, (C + 255) / 256), dim3(256), 0, s, partial, chunks, k0, L0, w0, gamma, beta, C, gn);
    return hipGetLastError();
}

hipError_t launch_conv0(int dtype, const Conv0Params& p, hipStream_t s) {
    // the kernel is specialised for the first layer every released checkpoint has: k = 10, stride <= 8, C <= 1024
    if (p.k0 != 10 || p.s0 > 8 || p.s0 < 1 || p.C > 1024 || (p.C & 3)) return hipErrorInvalidValue;
    switch (dtype) {
        case F32: return ((p.fast ? tuning().conv0_fast : tuning().gelu32) == 1) ? conv0_dispatch<float, true>(p, s) : conv0_dispatch<float, false>(p, s);
        // (conv0_fast = 0: the FAST = false instantiation, a diagnostic of the concurrent-forward finding — kernels.h)
        case BF16: return tuning().conv0_fast ? conv0_dispatch<bf16_tag, true>(p, s) : conv0_dispatch<bf16_tag, false>(p, s);
        case F16: return tuning().conv0_fast ? conv0_dispatch<f16_tag, true>(p, s) : conv0_dispatch<f16_tag, false>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace s3
