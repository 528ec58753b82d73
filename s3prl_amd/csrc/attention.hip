// attention.hip — padding-masked multi-head self-attention, head_dim 64, flash-style (scores never leave the CU).
//
// Restates the math F.multi_head_attention_forward executes for the reference call sites
// wav2vec2_model.py:1146-1168 and wavlm/modules.py:556-579:  softmax(q k^T + mask) v  with q pre-scaled by
// head_dim^-0.5 (folded into W_q at pack time), key-padding mask as -inf, fp32 softmax, and for WavLM the additive
// gated relative-position bias  gate[b,h,i] * table[h, (j-i)+(T-1)]  (wavlm/modules.py:448-462,535-551) applied
// in-kernel — the reference's (B*H, T, T) fp32 bias tensor (1.15 GB per layer at B=32, T=749) is never built.
//
// Layout trick (no transposes, no P round-trip through LDS): each wave owns 32 queries and computes the
// TRANSPOSED score tile  S^T = K Q^T  with one 32x32 MFMA chain, so a lane holds 16 keys of ONE query
// (col = lane&31 = query).  Row max / row sum are then lane-local plus one half-wave exchange, and the
// probabilities are already in the B-operand layout of the second MFMA chain  O^T += V^T P^T.
// fp32 path: v_mfma_f32_32x32x2_f32 (exact);  16-bit path: v_mfma_f32_32x32x16_{bf16,f16} with V staged
// transposed in LDS.  Padded queries are computed like the reference does; keys beyond
// valid[b] are skipped tile-wise and masked inside the last tile.
#include "kernels.h"

namespace s3 {
namespace {

constexpr int HD = 64;        // head dim
constexpr int QT = 128;       // queries per workgroup (4 waves x 32)
constexpr int KT = 32;        // keys per tile
constexpr int KS32 = HD + 4;  // fp32 LDS row stride (floats): 272 B rows -> conflict-free ds_read_b128

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

struct BiasCtx {
    const float* table;  // this head's [2T-1] row or null
    float gate;
    int qpos;  // query index + (T-1) offset folded:  idx = key - q + (T-1)
};

__global__ __launch_bounds__(256) void attn_f32_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * KS32];
    __shared__ __attribute__((aligned(16))) float Vs[KT * KS32];
    const int b = blockIdx.z, head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = p.H * HD;
    const long ld = 3L * D;
    const float* base = (const float*)p.qkv + (long)b * p.T * ld + head * HD;
    const int q_g = blockIdx.x * QT + wave * 32 + l31;
    const int q_c = q_g < p.T ? q_g : p.T - 1;

    // Q fragment: B operand, lane (q, half) holds Q[q][half*32 + s], s = 0..31
    float qf[32];
    {
        const float* qp = base + (long)q_c * ld + half * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = *(const float4*)(qp + 4 * i);
            qf[4 * i] = t.x;
            qf[4 * i + 1] = t.y;
            qf[4 * i + 2] = t.z;
            qf[4 * i + 3] = t.w;
        }
    }
    const float* btab = p.bias_table ? p.bias_table + (long)head * (2 * p.T - 1) : nullptr;
    const float gate = (btab && p.gate) ? p.gate[((long)b * p.H + head) * p.T + q_c] : 1.f;

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int valid = p.valid[b];
    const int ntiles = (valid + KT - 1) / KT;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            int kr = kt * KT + row;
            kr = kr < p.T ? kr : p.T - 1;
            const float* src = base + (long)kr * ld + c4 * 4;
            *(float4*)(Ks + row * KS32 + c4 * 4) = *(const float4*)(src + D);
            *(float4*)(Vs + row * KS32 + c4 * 4) = *(const float4*)(src + 2 * D);
        }
        __syncthreads();

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = Ks + l31 * KS32 + half * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 kf = *(const float4*)(kp + 4 * i);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * i], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * i + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * i + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * i + 3], s, 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * KT + crow(r, half);
            float v = s[r];
            if (btab && key < p.T) v += gate * btab[key - q_c + p.T - 1];
            v = key < valid ? v : -INFINITY;
            s[r] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            ps += s[r];
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vp = Vs + crow(r, half) * KS32 + l31;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], s[r], o1, 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (q_g < p.T) {
        float* op = (float*)p.out + ((long)b * p.T + q_g) * D + head * HD + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(float4*)(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *(float4*)(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

// ---- 16-bit operands ------------------------------------------------------------------------------------------
constexpr int KS16 = HD + 8;   // u16 per K row: 144 B rows -> conflict-free ds_read_b128
constexpr int VS16 = KT + 4;   // u16 per V^T row: 72 B rows -> conflict-free ds_read_b64

template <typename T> struct Mma16;
template <> struct Mma16<bf16_tag> {
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<f16_tag> {
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

template <typename T>
__global__ __launch_bounds__(256) void attn_h16_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) u16 Ks[KT * KS16];
    __shared__ __attribute__((aligned(16))) u16 Vt[HD * VS16];
    const int b = blockIdx.z, head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = p.H * HD;
    const long ld = 3L * D;
    const u16* base = (const u16*)p.qkv + (long)b * p.T * ld + head * HD;
    const int q_g = blockIdx.x * QT + wave * 32 + l31;
    const int q_c = q_g < p.T ? q_g : p.T - 1;

    // Q fragment (B operand): step st covers dims st*16 .. +15, this half-wave holds 8 of them
    uint4 qf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const uint4*)(base + (long)q_c * ld + st * 16 + 8 * half);

    const float* btab = p.bias_table ? p.bias_table + (long)head * (2 * p.T - 1) : nullptr;
    const float gate = (btab && p.gate) ? p.gate[((long)b * p.H + head) * p.T + q_c] : 1.f;

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int valid = p.valid[b];
    const int ntiles = (valid + KT - 1) / KT;
    const int srow = tid >> 3, sc8 = tid & 7;  // staging: key row, 8-element column group
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        {
            int kr = kt * KT + srow;
            kr = kr < p.T ? kr : p.T - 1;
            const u16* src = base + (long)kr * ld + sc8 * 8;
            *(uint4*)(Ks + srow * KS16 + sc8 * 8) = *(const uint4*)(src + D);
            const uint4 vv = *(const uint4*)(src + 2 * D);
            const unsigned w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Vt[(sc8 * 8 + 2 * i) * VS16 + srow] = (u16)(w[i] & 0xffffu);
                Vt[(sc8 * 8 + 2 * i + 1) * VS16 + srow] = (u16)(w[i] >> 16);
            }
        }
        __syncthreads();

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const uint4 kf = *(const uint4*)(Ks + l31 * KS16 + st * 16 + 8 * half);
            s = Mma16<T>::run(kf, qf[st], s);
        }
        // VALU, not the matrix pipe, bounds this kernel at the 16-bit MFMA rate, so the softmax is kept lean:
        // bias / mask passes only where they apply, and the running max is only raised (and O, l rescaled) when some
        // query's tile max exceeds it by more than 8 — softmax is invariant to the reference point, e^8 fits every
        // operand type, and after the first tile the rescale of the 32 O registers is almost never needed.
        if (btab) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * KT + crow(r, half);
                if (key < p.T) s[r] += gate * btab[key - q_c + p.T - 1];
            }
        }
        if (kt * KT + KT > valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = kt * KT + crow(r, half) < valid ? s[r] : -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__any(mx > m_run + 8.f)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * 1.44269504088896340736f);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= alpha;
                o1[r] *= alpha;
            }
        }
        const float mneg = -m_run * 1.44269504088896340736f;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], 1.44269504088896340736f, mneg));
            ps += s[r];
        }
        l_run += ps;
        // P^T as B operand: step u uses regs 8u..8u+7  <->  keys 16u + {0,1,2,3,8,9,10,11} + 4*half
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint4 pf;
            pf.x = Cvt<T>::pack2(s[8 * u + 0], s[8 * u + 1]);
            pf.y = Cvt<T>::pack2(s[8 * u + 2], s[8 * u + 3]);
            pf.z = Cvt<T>::pack2(s[8 * u + 4], s[8 * u + 5]);
            pf.w = Cvt<T>::pack2(s[8 * u + 6], s[8 * u + 7]);
            const u16* v0 = Vt + l31 * VS16 + 16 * u + 4 * half;
            const u16* v1 = v0 + 32 * VS16;
            const uint2 a00 = *(const uint2*)(v0), a01 = *(const uint2*)(v0 + 8);
            const uint2 a10 = *(const uint2*)(v1), a11 = *(const uint2*)(v1 + 8);
            o0 = Mma16<T>::run(make_uint4(a00.x, a00.y, a01.x, a01.y), pf, o0);
            o1 = Mma16<T>::run(make_uint4(a10.x, a10.y, a11.x, a11.y), pf, o1);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (q_g < p.T) {
        u16* op = (u16*)p.out + ((long)b * p.T + q_g) * D + head * HD + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ushort4 h0, h1;
            h0.x = Cvt<T>::to(o0[4 * g] * inv);
            h0.y = Cvt<T>::to(o0[4 * g + 1] * inv);
            h0.z = Cvt<T>::to(o0[4 * g + 2] * inv);
            h0.w = Cvt<T>::to(o0[4 * g + 3] * inv);
            h1.x = Cvt<T>::to(o1[4 * g] * inv);
            h1.y = Cvt<T>::to(o1[4 * g + 1] * inv);
            h1.z = Cvt<T>::to(o1[4 * g + 2] * inv);
            h1.w = Cvt<T>::to(o1[4 * g + 3] * inv);
            *(ushort4*)(op + 8 * g) = h0;
            *(ushort4*)(op + 32 + 8 * g) = h1;
        }
    }
}

// WavLM gate from the layer input split into heads (wavlm/modules.py:535-549):
//   g = sigmoid( sum4( grep_linear(x_head) ) ) -> (a, b);  gate = a * (b * grep_a[h] - 1) + 2
__global__ __launch_bounds__(256) void wavlm_gate_kernel(const float* x, const float* gw, const float* gb, const float* ga,
                                                         int B, int T, int H, float* gate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*T*H
    const long total = (long)B * T * H;
    if (idx >= total) return;
    const int h = (int)(idx % H);
    const long bt = idx / H;
    const float* xr = x + bt * (long)H * HD + h * HD;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = gb[o];
    for (int k = 0; k < HD; ++k) {
        const float xv = xr[k];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = fmaf(gw[o * HD + k], xv, acc[o]);
    }
    const float sa = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    const float sb = (acc[4] + acc[5]) + (acc[6] + acc[7]);
    const float a = 1.f / (1.f + __expf(-sa));
    const float bb = 1.f / (1.f + __expf(-sb));
    const int b = (int)(bt / T), t = (int)(bt % T);
    gate[((long)b * H + h) * T + t] = a * (bb * ga[h] - 1.f) + 2.f;
}

}  // namespace

hipError_t launch_attention(int dtype, const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return hipSuccess;
    dim3 grid((p.T + QT - 1) / QT, p.H, p.B), block(256);
    switch (dtype) {
        case F32: hipLaunchKernelGGL(attn_f32_kernel, grid, block, 0, s, p); break;
        case BF16: hipLaunchKernelGGL(attn_h16_kernel<bf16_tag>, grid, block, 0, s, p); break;
        case F16: hipLaunchKernelGGL(attn_h16_kernel<f16_tag>, grid, block, 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_wavlm_gate(const float* x, const float* grep_w, const float* grep_b, const float* grep_a, int B, int T,
                             int H, float* gate, hipStream_t s) {
    const long total = (long)B * T * H;
    hipLaunchKernelGGL(wavlm_gate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, grep_w, grep_b, grep_a, B,
                       T, H, gate);
    return hipGetLastError();
}

}  // namespace s3
