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
#include <algorithm>
#include <type_traits>

#include "kernels.h"

// Timing probes (results are garbage by design): compiled in only by tools/micro/attn_lab.hip
//   1 = stage only the first K/V tile (no global loads / LDS stores afterwards), 2 = no softmax VALU (p = score),
//   4 = no P.V MFMAs, 8 = no Q.K MFMAs, 16 = no per-tile barrier
#ifdef S3_ATTN_PROBE
#define S3_PROBE(p_, bit_) ((p_).probe & (bit_))
#else
#define S3_PROBE(p_, bit_) 0
#endif

namespace s3 {
namespace {

constexpr int HD = 64;        // head dim
constexpr int QT = 128;       // queries per workgroup (4 waves x 32)
constexpr int KT = 32;        // keys per tile
constexpr int KS32 = HD + 4;  // fp32 LDS row stride (floats): 272 B rows -> conflict-free ds_read_b128
constexpr int BIAS_PAD = 64;  // WavLM bias window: entries past T + QT - 1 so that the keys of a partial last tile stay in range

// XCD-aware work map.  Workgroup w of a 1-D grid runs on XCD w % 8 (each XCD has a private 4 MiB L2).  The query blocks of
// one (batch, head) unit all read that unit's K and V: with the plain (q-block, head, batch) grid they land on `nqb`
// DIFFERENT XCDs and every L2 fetches its own copy (round 2: 210 MB fetched per launch for 74 MB of q|k|v, L2 hit 0.42 — the
// 16-bit kernel ran at the HBM rate, not at any on-chip limit).  Here XCD x owns the units u = x (mod 8) and runs a unit's
// query blocks back to back, so K / V come from HBM once.  The launcher pads the grid to 8 * ceil(units / 8) * nqb.
struct AttnWork {
    int b, head, qb;
    bool live;
};
__device__ __forceinline__ AttnWork attn_work(const AttnParams& p) {
    const int nqb = (p.T + 127) / 128;
    const int wg = blockIdx.x, xcd = wg & 7, local = wg >> 3;
    const int unit = xcd + 8 * (local / nqb);
    AttnWork w;
    w.qb = local % nqb;
    w.live = unit < p.B * p.H;
    w.b = unit / p.H;
    w.head = unit % p.H;
    return w;
}

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Exchange between the two half-waves (lane l <-> l ^ 32) WITHOUT the LDS: v_permlane32_swap swaps the upper half of one
// register with the lower half of another, so swap(x, copy of x) leaves {x_lo, x_lo} and {x_hi, x_hi}.  __shfl_xor(.., 32)
// lowers to ds_bpermute_b32 + s_waitcnt lgkmcnt(0), i.e. an LDS round trip that also drains the prefetched K / V fragment
// reads — once per 32-key half on the softmax's critical path.
__device__ __forceinline__ float xhalf_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct BiasCtx {
    const float* table;  // this head's [2T-1] row or null
    float gate;
    int qpos;  // query index + (T-1) offset folded:  idx = key - q + (T-1)
};

__global__ __launch_bounds__(256, 3) void attn_f32_kernel(AttnParams p) {
    // K and V tiles double-buffered in LDS; the next tile's global loads are in flight (registers) while the current
    // tile is multiplied: one barrier per 32 keys
    __shared__ __attribute__((aligned(16))) float Ks[2 * KT * KS32];
    __shared__ __attribute__((aligned(16))) float Vs[2 * KT * KS32];
    const AttnWork wk = attn_work(p);
    if (!wk.live) return;
    const int b = wk.b, head = wk.head;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = p.H * HD;
    const long ld = 3L * D;
    const float* base = (const float*)p.qkv + (long)b * p.T * ld + head * HD;
    const int q_g = wk.qb * QT + wave * 32 + l31;
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
    // WavLM: the window of this head's relative-position table that the workgroup's 128 queries can meet (T + 127
    // entries: key - query in [-(q0 + 127), T - 1 - q0]) is gathered once into LDS from the (2R+1)-entry global table,
    // clamped to |key - query| <= R = max_distance where the bucket saturates (wavlm/modules.py:436-444); a per-element
    // global gather would bound the kernel
    extern __shared__ float bias_s[];
    const float* btab = nullptr;
    if (p.bias_table) {
        const int R = p.table_R;
        const float* src = p.bias_table + (long)head * (2 * R + 1) + R;
        const int rel0 = -(wk.qb * QT + QT - 1);  // smallest key - query this workgroup can meet
        for (int i = threadIdx.x; i < p.T + QT - 1 + BIAS_PAD; i += 256) bias_s[i] = src[min(max(rel0 + i, -R), R)];
        btab = bias_s;  // made visible by the first __syncthreads() of the key loop
    }
    const float gate = (btab && p.gate) ? p.gate[((long)b * p.H + head) * p.T + q_c] : 1.f;
    const int bias_off = wk.qb * QT + QT - 1;  // window index = (key - query) + bias_off

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int valid = p.valid[b];
    const int ntiles = (valid + KT - 1) / KT;
    f32x4 kreg[2], vreg[2];  // native vectors (HIP float4 arrays held across the loop end up in scratch)
    const int srow = tid >> 4, sc4 = tid & 15;  // staging: rows srow and srow+16, float4 column sc4
#define A32_LOAD(kt_)                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                       \
        int kr_ = (kt_) * KT + srow + 16 * i_;                                               \
        kr_ = kr_ < p.T ? kr_ : p.T - 1;                                                     \
        const float* src_ = base + (long)kr_ * ld + sc4 * 4;                                 \
        kreg[i_] = *(const f32x4*)(src_ + D);                                                \
        vreg[i_] = *(const f32x4*)(src_ + 2 * D);                                            \
    }
#define A32_STORE(buf_)                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                       \
        *(f32x4*)(Ks + (buf_) * KT * KS32 + (srow + 16 * i_) * KS32 + sc4 * 4) = kreg[i_];   \
        *(f32x4*)(Vs + (buf_) * KT * KS32 + (srow + 16 * i_) * KS32 + sc4 * 4) = vreg[i_];   \
    }
    A32_LOAD(0)
    A32_STORE(0)
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) { A32_LOAD(kt + 1) }
        const float* Kb = Ks + (kt & 1) * KT * KS32;
        const float* Vb = Vs + (kt & 1) * KT * KS32;

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = Kb + l31 * KS32 + half * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 kf = *(const float4*)(kp + 4 * i);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * i], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * i + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * i + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * i + 3], s, 0, 0, 0);
        }
        // bias / mask passes only where they apply; the running max is only raised (and O, l rescaled) when some
        // query's tile max exceeds it by more than 8 — softmax is invariant to the reference point (exact in real
        // arithmetic, rounding-level in fp32) and the 32 O registers are then almost never rescaled
        if (btab) {
            // branch-free: the window is padded by BIAS_PAD entries, so keys of the last (partial) tile index valid LDS;
            // their scores are masked below.  16 reads at compile-time offsets from one base instead of 16 predicated
            // read-wait-add sequences
            {
                const float* bb = btab + (kt * KT + 4 * half - q_c + bias_off);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = fmaf(gate, bb[(r & 3) + 8 * (r >> 2)], s[r]);
            }
        }
        if (kt * KT + KT > valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = kt * KT + crow(r, half) < valid ? s[r] : -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = xhalf_max(mx);
        if (__any(mx > m_run + 8.f)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= alpha;
                o1[r] *= alpha;
            }
        }
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_run);
            ps += s[r];
        }
        l_run += ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vp = Vb + crow(r, half) * KS32 + l31;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], s[r], o1, 0, 0, 0);
        }
        if (kt + 1 < ntiles) { A32_STORE((kt + 1) & 1) }
        __syncthreads();
    }
#undef A32_LOAD
#undef A32_STORE
    const float l_tot = xhalf_sum(l_run);
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

// ---- round 6: the persistent form of the fp32 kernel (the headline mode's attention) ---------------------------------------------
// Same arithmetic in the same order as attn_f32_kernel (bit-identical), the life cycle of attn_h16p_kernel below: 3 resident
// workgroups per CU walk the (batch, head, query block) items of their XCD; the next item's first K / V tile rides in the staging
// registers under the current item's last tile and its Q fragment is fetched while the current item's result is normalised and
// stored.  The fp32 kernel moves twice the bytes of the 16-bit one (147 MB of q|k|v in, 49 MB out per HuBERT-base launch): with a
// one-shot grid every workgroup of a round loads, multiplies and stores in phase and ~17 % of the launch has no MFMA to issue
// (mfma_busy 0.83, profiles/r05_pmc_fp32.md).
__global__ __launch_bounds__(256, 3) void attn_f32p_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[2 * KT * KS32];
    __shared__ __attribute__((aligned(16))) float Vs[2 * KT * KS32];
    extern __shared__ float bias_s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = p.H * HD, ldi = 3 * D;
    const int nqb = (p.T + QT - 1) / QT;
    const int xcd = blockIdx.x & 7, wpx = gridDim.x >> 3;  // (the launcher's grid is a multiple of 8)
    const int units = p.B * p.H;
    const int n_items = (units > xcd ? (units - xcd + 7) / 8 : 0) * nqb;
    int it = blockIdx.x >> 3;
    if (it >= n_items) return;

    struct Item {
        const float* base;  // this (batch, head)'s q row 0
        int b, head, qb;
    };
    auto item_of = [&](int i) {
        Item w;
        const int unit = xcd + 8 * (i / nqb);
        w.qb = i % nqb;
        w.b = unit / p.H;
        w.head = unit % p.H;
        w.base = (const float*)p.qkv + (long)w.b * p.T * ldi + w.head * HD;
        return w;
    };
    // rows 0 .. T - 1 of the item's (batch, head) behind a buffer descriptor: one constant 32-bit byte offset per lane and load, the
    // tile index in the scalar offset; rows past the last frame read as ZERO (their scores are masked: P = 0 meets V = 0)
    auto rsrc_of = [&](const Item& w) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)w.base, 0, (p.T * ldi - w.head * HD) * 4, 0x00020000);
    };
    const int srow = tid >> 4, sc4 = tid & 15;  // staging: rows srow and srow + 16, float4 column sc4
    const unsigned tstride_b = (unsigned)(KT * ldi * 4);
    const unsigned so0 = (unsigned)((srow * ldi + sc4 * 4) * 4), so1 = so0 + (unsigned)(16 * ldi * 4);
    f32x4 kreg[2], vreg[2];
    auto load_tile = [&](const __amdgpu_buffer_rsrc_t& rs, int kt) {
        const unsigned so = (unsigned)kt * tstride_b;
        kreg[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, so0 + (unsigned)(D * 4), so, 0));
        vreg[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, so0 + (unsigned)(2 * D * 4), so, 0));
        kreg[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, so1 + (unsigned)(D * 4), so, 0));
        vreg[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, so1 + (unsigned)(2 * D * 4), so, 0));
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(f32x4*)(Ks + buf * KT * KS32 + (srow + 16 * i) * KS32 + sc4 * 4) = kreg[i];
            *(f32x4*)(Vs + buf * KT * KS32 + (srow + 16 * i) * KS32 + sc4 * 4) = vreg[i];
        }
    };

    Item cur = item_of(it);
    float qf[32];
    int q_g, q_c, valid, ntiles, bias_off;
    float gate = 1.f;
    const float* btab = nullptr;
    auto enter_scalars = [&](const Item& w) {
        q_g = w.qb * QT + wave * 32 + l31;
        q_c = q_g < p.T ? q_g : p.T - 1;
        valid = p.valid[w.b];
        ntiles = (valid + KT - 1) / KT;
        bias_off = w.qb * QT + QT - 1;  // window index = (key - query) + bias_off
    };
    auto load_q = [&](const __amdgpu_buffer_rsrc_t& rs) {  // B operand: lane (q, half) holds Q[q][half * 32 + s], s = 0 .. 31
        const unsigned qo = (unsigned)((q_c * ldi + half * 32) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, qo + 16 * i, 0, 0));
            qf[4 * i] = t[0];
            qf[4 * i + 1] = t[1];
            qf[4 * i + 2] = t[2];
            qf[4 * i + 3] = t[3];
        }
    };
    auto enter_bias = [&](const Item& w) {  // WavLM: the item's window of its head's table (see attn_f32_kernel) + the query's gate
        if (p.bias_table) {
            const int R = p.table_R;
            const float* src = p.bias_table + (long)w.head * (2 * R + 1) + R;
            const int rel0 = -(w.qb * QT + QT - 1);
            for (int i = threadIdx.x; i < p.T + QT - 1 + BIAS_PAD; i += 256) bias_s[i] = src[min(max(rel0 + i, -R), R)];
            btab = bias_s;
            gate = p.gate ? p.gate[((long)w.b * p.H + w.head) * p.T + q_c] : 1.f;
        }
    };
    enter_scalars(cur);
    __amdgpu_buffer_rsrc_t rs_cur = rsrc_of(cur);
    load_q(rs_cur);
    enter_bias(cur);
    load_tile(rs_cur, 0);
    store_tile(0);
    __syncthreads();
    int buf = 0;  // the LDS buffer the next tile to multiply sits in (runs on across items)

    f32x16 o0, o1;
    while (true) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        const int nxt = it + wpx;
        const bool has_next = nxt < n_items;
        Item nx = cur;
        if (ntiles == 0 && has_next) {  // (an utterance without a single valid frame: nothing to multiply, the hand-over still happens)
            nx = item_of(nxt);
            rs_cur = rsrc_of(nx);
            load_tile(rs_cur, 0);
        }
        for (int kt = 0; kt < ntiles; ++kt) {
            if (kt + 1 < ntiles) {
                load_tile(rs_cur, kt + 1);
            } else if (has_next) {  // the next item's first tile rides in the staging registers under this item's last tile
                nx = item_of(nxt);
                rs_cur = rsrc_of(nx);
                load_tile(rs_cur, 0);
            }
            const float* Kb = Ks + buf * KT * KS32;
            const float* Vb = Vs + buf * KT * KS32;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const float* kp = Kb + l31 * KS32 + half * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 kf = *(const float4*)(kp + 4 * i);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * i], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * i + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * i + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * i + 3], s, 0, 0, 0);
            }
            if (btab) {
                const float* bb = btab + (kt * KT + 4 * half - q_c + bias_off);  // branch-free, see attn_f32_kernel
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = fmaf(gate, bb[(r & 3) + 8 * (r >> 2)], s[r]);
            }
            if (kt * KT + KT > valid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = kt * KT + crow(r, half) < valid ? s[r] : -INFINITY;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = xhalf_max(mx);
            if (__any(mx > m_run + 8.f)) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __expf(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __expf(s[r] - m_run);
                ps += s[r];
            }
            l_run += ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vp = Vb + crow(r, half) * KS32 + l31;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], s[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], s[r], o1, 0, 0, 0);
            }
            if (kt + 1 < ntiles) {
                store_tile(buf ^ 1);
                __syncthreads();
                buf ^= 1;
            }
        }
        // ---- seam: this item's result goes out while the next item's operands come in ----
        const float inv = 1.f / xhalf_sum(l_run);
        const bool q_ok = q_g < p.T;
        const long obase = (long)cur.b * p.T * D + cur.head * HD;  // (wave-uniform)
        const unsigned orow = (unsigned)(q_g * D + 4 * half);
        if (has_next) {
            store_tile(buf ^ 1);  // (frees the staging registers)
            enter_scalars(nx);    // (q_g / q_c / valid / ntiles now belong to the next item)
            load_q(rs_cur);       // lands under the normalisation and the stores below
        }
        if (q_ok) {
            float* op = (float*)p.out + obase + orow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *(float4*)(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                *(float4*)(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            }
        }
        if (!has_next) break;
        __syncthreads();  // every wave is past its last tile: the other K / V buffer (and the bias window) may be re-used
        buf ^= 1;
        it = nxt;
        cur = nx;
        if (p.bias_table) {
            enter_bias(cur);
            __syncthreads();
        }
    }
}

// ---- 16-bit operands ------------------------------------------------------------------------------------------
// 64 keys per tile, K and V^T double-buffered in LDS, the next tile's global loads in flight (registers) while the
// current one is multiplied: one barrier per 64 keys.  V was transposed on the way into LDS two keys at a time (32-bit
// writes of a key pair per dim, 16 VALU + 8 LDS writes per thread and tile) so that the A operand of O^T += V^T P^T is read
// as 8-byte vectors — still the persistent kernel's and the fp32x3 kernel's form; the one-shot kernel: switch 256 below.
#ifndef S3_ATTN_EXP
#define S3_ATTN_EXP 420  // lab switches (tools/micro/build.sh): 1 = next-tile loads pinned, 4 = the bias kernel with the scalar reference at
                       // three waves per SIMD, 8 = s_setprio(1) around a tile's work, 16 = the staged tile written to LDS mid-tile,
                       // 32 = the reference block from the matrix pipe (RefM<2>), 128 = 16-byte result stores (v_permlane32_swap
                       // pairs), 256 = V row-major in LDS + ds_read_b64_tr_b16 (one-shot kernel).  Default 4 | 32 | 128 | 256 since
                       // the second session of round 6 (profiles/r06b_attn_lab_variants.md); 0 rebuilds the first session's kernels
#endif
constexpr int KT16 = 64;         // keys per tile
constexpr int KS16 = HD + 8;     // u16 per K row: 144 B rows -> conflict-free ds_read_b128
constexpr int VS16 = KT16 + 4;   // u16 per V^T row: 136 B rows -> conflict-free ds_read_b64
constexpr int KBUF16 = KT16 * KS16, VBUF16 = HD * VS16;
// Lab switch 256 (round 6, second session): V stays ROW-MAJOR in LDS ([key][dim], two 16-byte writes per thread and tile like K — no
// transposition on the VALU) and the A operand of O^T += V^T P^T is fetched with ds_read_b64_tr_b16: within a group of 16 lanes, lane
// 4 j + r supplies the address of dims 4 r .. 4 r + 3 of key j and lane i receives dim i of keys 0..3.  160-byte rows: the four 32-byte
// row pieces a lane group touches fall into four different bank octets.
constexpr int VSR16 = HD + 16, VBUFR16 = KT16 * VSR16;
constexpr bool ATTN_VTR = (S3_ATTN_EXP & 256) != 0;
constexpr int VBUFX16 = ATTN_VTR ? VBUFR16 : VBUF16;
__device__ __forceinline__ uint2 lds_tr16(const u16* p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}

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

// The softmax reference -m as the initial value of a score block's accumulator.  Vector form: 16 registers that only change when
// the reference moves (the MFMA takes them as its C operand directly).  Scalar form (round 6): ONE register, the 16-register block is
// filled from it per 32-key half (16 v_mov) — the bias-free kernel drops from 168 registers + 5 spilled to 144 and no spill, and its
// compiler-made schedule gets 7-10 % faster (attn_lab: 62.4 -> 58.0 us HuBERT-base, 81.0 -> 73.1 HuBERT-large, 150 -> 139 WavLM-large
// without bias; profiles/r06_attn_lab.md).  With the vector form the bias kernel ran at two waves per SIMD (the scalar form at three:
// 154 -> 168 us, the extra moves only cost).  Same values either way: bit-identical.  Since the second session of round 6 every 16-bit
// kernel takes the matrix-pipe form below (RefM<2>), the bias kernel at three waves per SIMD with it (162 registers, no spill).
template <int KIND, typename T> struct RefM;
template <typename T> struct RefM<0, T> {
    f32x16 v;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
    }
    __device__ __forceinline__ f32x16 block() const { return v; }
    __device__ __forceinline__ float lower(float d) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] -= d;
        return d;
    }
};
template <typename T> struct RefM<1, T> {
    float s;
    __device__ __forceinline__ void zero() { s = 0.f; }
    __device__ __forceinline__ f32x16 block() const {
        f32x16 x;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = s;
        return x;
    }
    __device__ __forceinline__ float lower(float d) {
        s -= d;
        return d;
    }
};
// Matrix-pipe form (round 6, second session): the block of -m comes out of ONE extra MFMA instead of 16 v_mov per 32-key half —
// the kernel's compute phase is bound by VALU issue (profiles/r06_attn_lab.md) while the matrix pipe idles two thirds of it.  The
// reference is kept as a two-term 16-bit number m = hi + lo (exact in the fp32 accumulator): the lower half-wave's A fragment
// carries 1, 1 in k slots 0 and 1 (the upper half-wave's eight slots are zero), every lane's B fragment -hi, -lo of its query.
// Softmax is invariant to the reference point, so rounding m to 16 + 16 bits costs nothing as long as every score, the rescale
// factor and the running sum see the SAME m: lower() returns the step the reference really took.
template <typename T> struct RefM<2, T> {
    float m;        // the reference (= hi + lo exactly)
    unsigned nb;    // B fragment word 0: (-hi, -lo)
    unsigned a1;    // A fragment word 0: (1, 1) on lanes 0-31, 0 on lanes 32-63
    __device__ __forceinline__ void zero() {
        m = 0.f;
        nb = 0u;
        a1 = (threadIdx.x & 32) ? 0u : Cvt<T>::pack2(1.f, 1.f);
    }
    __device__ __forceinline__ f32x16 block() const {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        return Mma16<T>::run(make_uint4(a1, 0u, 0u, 0u), make_uint4(nb, 0u, 0u, 0u), z);
    }
    __device__ __forceinline__ float lower(float d) {
        const float want = m + d;
        const float hi = Cvt<T>::from(Cvt<T>::to(want));
        const float lo = Cvt<T>::from(Cvt<T>::to(want - hi));
        const float m_new = hi + lo;
        const float step = m_new - m;
        m = m_new;
        nb = Cvt<T>::pack2(-hi, -lo);
        return step;
    }
};
constexpr int attn_ref_kind(bool bias) { return (!bias || (S3_ATTN_EXP & 4)) ? ((S3_ATTN_EXP & 32) ? 2 : 1) : 0; }

// maximum of the 16 scores a lane holds.  v_med3_f32(a, b, +inf) = max(a, b) without the canonicalising v_max that fmaxf puts on
// every MFMA output; the compiler folds the chain into v_max3_f32 pairs itself.  (An inline-asm v_max3_f32 tree was tried and is
// WRONG: the hazard recogniser does not count an asm statement as a reader of MFMA results, the required wait states are not
// inserted and the tree reads accumulators that are still being written — f16 rows came out inf in tools/micro/attn_lab.)
__device__ __forceinline__ float row_max16(const f32x16& sc) {
    float mx = __builtin_amdgcn_fmed3f(sc[0], sc[1], INFINITY);
#pragma unroll
    for (int r = 2; r < 16; ++r) mx = __builtin_amdgcn_fmed3f(mx, sc[r], INFINITY);
    return mx;
}

// the normalised 32 x 64 result of a wave as 16-bit rows.  A lane holds, of its query's row, the dims 8g + 4 half + {0..3} of both
// 32-dim blocks: 8-byte pieces, eight stores.  Lab switch 128: v_permlane32_swap hands the lower half-wave the upper one's piece
// of group g and the upper one the lower's piece of group g + 1, so that each lane owns 16 contiguous bytes: four stores.
template <typename T>
__device__ __forceinline__ void store_o16(u16* row, int half, const f32x16& o0, const f32x16& o1, float inv) {
#if S3_ATTN_EXP & 128
    u16* op = row + 8 * half;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const f32x16& o = blk ? o1 : o0;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            unsigned ax = Cvt<T>::pack2(o[4 * g] * inv, o[4 * g + 1] * inv), ay = Cvt<T>::pack2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            unsigned bx = Cvt<T>::pack2(o[4 * g + 4] * inv, o[4 * g + 5] * inv), by = Cvt<T>::pack2(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
            const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            *(uint4*)(op + 32 * blk + 8 * g) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
    }
#else
    u16* op = row + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *(uint2*)(op + 8 * g) = make_uint2(Cvt<T>::pack2(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                                           Cvt<T>::pack2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
        *(uint2*)(op + 32 + 8 * g) = make_uint2(Cvt<T>::pack2(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                                                Cvt<T>::pack2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
    }
#endif
}
constexpr int attn_h16_waves(bool bias) { return (bias && !(S3_ATTN_EXP & 4)) ? 2 : 3; }

// BIAS: the WavLM relative-position bias path compiled in (with a 16-register reference block its 16 table reads in flight need
// > 168 registers: two waves per SIMD; with the matrix-pipe reference 162 = three); the bias-free variant: 144 registers.
// Operand contract of the 16-bit kernels: q arrives pre-scaled by head_dim^-0.5 * log2(e) (folded into W_q / b_q at pack
// time), so the scores are base-2 logarithms and the softmax is exp2 without a per-score multiply.
template <typename T, bool BIAS>
__global__ __launch_bounds__(256, attn_h16_waves(BIAS)) void attn_h16_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) u16 Ks[2 * KBUF16];
    __shared__ __attribute__((aligned(16))) u16 Vt[2 * VBUFX16];
    const AttnWork wk = attn_work(p);
    if (!wk.live) return;
    const int b = wk.b, head = wk.head;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = p.H * HD;
    const long ld = 3L * D;
    const u16* base = (const u16*)p.qkv + (long)b * p.T * ld + head * HD;
    const int q_g = wk.qb * QT + wave * 32 + l31;
    const int q_c = q_g < p.T ? q_g : p.T - 1;

    // Q fragment (B operand): step st covers dims st*16 .. +15, this half-wave holds 8 of them
    uint4 qf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const uint4*)(base + (long)q_c * ld + st * 16 + 8 * half);

    // WavLM: the window of this head's relative-position table that the workgroup's 128 queries can meet (T + 127
    // entries: key - query in [-(q0 + 127), T - 1 - q0]) is gathered once into LDS from the (2R+1)-entry global table,
    // clamped to |key - query| <= R = max_distance where the bucket saturates (wavlm/modules.py:436-444); a per-element
    // global gather would bound the kernel
    extern __shared__ float bias_s[];
    const float* btab = nullptr;
    if (BIAS && p.bias_table) {
        const int R = p.table_R;
        const float* src = p.bias_table + (long)head * (2 * R + 1) + R;
        const int rel0 = -(wk.qb * QT + QT - 1);  // smallest key - query this workgroup can meet
        for (int i = threadIdx.x; i < p.T + QT - 1 + BIAS_PAD; i += 256) bias_s[i] = src[min(max(rel0 + i, -R), R)];
        btab = bias_s;  // made visible by the first __syncthreads() of the key loop
    }
    const float gate = (btab && p.gate) ? p.gate[((long)b * p.H + head) * p.T + q_c] : 1.f;
    const int bias_off = wk.qb * QT + QT - 1;  // window index = (key - query) + bias_off

    // Softmax bookkeeping (round 3: the kernel was VALU-issue-bound — ~300 VALU + 32 exp per 64 keys beside 16 MFMAs, a
    // third of them address arithmetic for the staging loads):
    //   * the scores arrive in the LOG2 domain (log2(e) is folded into W_q / b_q with head_dim^-0.5 at pack time) and
    //     ALREADY relative to the running reference m: the accumulator of the S^T = K Q^T chain starts at -m (`negm`, 16
    //     registers that only change when the reference moves), so p = exp2(score) with no per-score multiply-add;
    //   * (the row sum stays on the VALU: carrying it on the matrix pipe — a third accumulator block with an all-ones A
    //     fragment — costs 20 registers and, with them, the third wave per SIMD);
    //   * the reference only moves when some query's scores exceed it by more than 2^8 (deferred max); the first 32 keys
    //     set it exactly (a reference of 0 could underflow every p of a row);
    //   * staging pointers advance by a constant per tile (the clamp to the last frame is only computed for the last tile)
    //     and the last (partial) tile is peeled, so the steady-state body has no mask / half-count branches;
    //   * one 32-key score tile is live at a time: with both halves' chains in flight (and their K / V fragments
    //     pre-loaded) the kernel needs 216 registers = two waves per SIMD, and was slower than this form at three.
    f32x16 o0, o1;
    RefM<attn_ref_kind(BIAS), T> negm;
    negm.zero();
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float l_run = 0.f;
    bool first = true;
    const float gate2 = gate * 1.44269504088896340736f;  // the bias joins log2-domain scores

    const int valid = p.valid[b];
    const int ntiles = (valid + KT16 - 1) / KT16;
    // staging roles: K — rows (tid>>3) and +32, 16-byte chunk tid&7;  V — key pair tid&31, dim group tid>>5
    const int krow = tid >> 3, kc8 = tid & 7;
    const int vj = tid & 31, vdg = tid >> 5;
    u32x4 kreg[2], vreg[2];
    const u16 *kp0, *kp1, *vp0, *vp1;
    auto set_ptrs = [&](int kt) {  // rows past the last frame re-read it (their scores are masked)
        auto clampk = [&](int kr) { return kr < p.T ? kr : p.T - 1; };
        const int k0_ = kt * KT16;
        kp0 = base + (long)clampk(k0_ + krow) * ld + D + kc8 * 8;
        kp1 = base + (long)clampk(k0_ + krow + 32) * ld + D + kc8 * 8;
        if (ATTN_VTR) {
            vp0 = base + (long)clampk(k0_ + krow) * ld + 2 * D + kc8 * 8;
            vp1 = base + (long)clampk(k0_ + krow + 32) * ld + 2 * D + kc8 * 8;
        } else {
            vp0 = base + (long)clampk(k0_ + 2 * vj) * ld + 2 * D + vdg * 8;
            vp1 = base + (long)clampk(k0_ + 2 * vj + 1) * ld + 2 * D + vdg * 8;
        }
    };
    auto load_tile = [&]() {
        kreg[0] = *(const u32x4*)kp0;
        kreg[1] = *(const u32x4*)kp1;
        vreg[0] = *(const u32x4*)vp0;
        vreg[1] = *(const u32x4*)vp1;
    };
    u16* const ks_st = Ks + krow * KS16 + kc8 * 8;
    u16* const vt_st = ATTN_VTR ? Vt + krow * VSR16 + kc8 * 8 : Vt + (vdg * 8) * VS16 + 2 * vj;
    auto store_tile = [&](int buf) {
        u16* ks_ = ks_st + buf * KBUF16;
        u16* vt_ = vt_st + buf * VBUFX16;
        *(u32x4*)ks_ = kreg[0];
        *(u32x4*)(ks_ + 32 * KS16) = kreg[1];
        if (ATTN_VTR) {
            *(u32x4*)vt_ = vreg[0];
            *(u32x4*)(vt_ + 32 * VSR16) = vreg[1];
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned a_ = vreg[0][i], b_ = vreg[1][i];
            *(unsigned*)(vt_ + (2 * i) * VS16) = (a_ & 0xffffu) | (b_ << 16);
            *(unsigned*)(vt_ + (2 * i + 1) * VS16) = (a_ >> 16) | (b_ & 0xffff0000u);
        }
    };
    set_ptrs(0);
    load_tile();
    store_tile(0);
    __syncthreads();
    const long tstride = (long)KT16 * ld;  // elements between the same row of consecutive tiles
    const u16* const kf_rd = Ks + l31 * KS16 + 8 * half;
    const u16* const vf_rd = ATTN_VTR ? Vt + (4 * half + ((lane & 15) >> 2)) * VSR16 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)
                                      : Vt + l31 * VS16 + 4 * half;

    // one 64-key tile.  FULL: every key is valid (all tiles but the last)
    auto tile = [&](auto full_c, int kt) {
        constexpr bool FULL = decltype(full_c)::value;
        const u16* ks = kf_rd + (kt & 1) * KBUF16;
        const u16* vt = vf_rd + (kt & 1) * VBUFX16;
        const bool two = FULL || kt * KT16 + 32 < valid;  // wave-uniform: the second half has at least one valid key
        auto scores = [&](int h) {
            f32x16 sc = negm.block();
            if (S3_PROBE(p, 8)) return sc;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const uint4 kf = *(const uint4*)(ks + h * 32 * KS16 + st * 16);
                sc = Mma16<T>::run(kf, qf[st], sc);
            }
            return sc;
        };
        // softmax of one 32-key half (`sc`: its scores minus the reference) and O^T += V^T P^T, la += 1 P^T
        auto half_step = [&](f32x16& sc, int h) {
            const int k0_ = kt * KT16 + h * 32;
            if (BIAS && btab) {
                const float* bb = btab + (k0_ + 4 * half - q_c + bias_off);  // branch-free, see attn_f32_kernel
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = fmaf(gate2, bb[(r & 3) + 8 * (r >> 2)], sc[r]);
            }
            if (!FULL && k0_ + 32 > valid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = k0_ + crow(r, half) < valid ? sc[r] : -INFINITY;
            }
            if (S3_PROBE(p, 2)) goto pv;
            {
            // max of the 16 scores as v_med3_f32(a, b, +inf): fmaxf on MFMA outputs costs a canonicalising v_max per input
            const float mx = xhalf_max(row_max16(sc));
            if (__builtin_expect(first || __any(mx > 8.f), 0)) {
                // move the reference: exactly onto the maximum for the first keys of a row, up by the excess afterwards
                const float delta = negm.lower(first ? mx : fmaxf(mx, 0.f));  // (the step the reference really takes)
                // first: O = l = 0 and the scale must be exactly 1 — exp2(-mx) overflows to +inf when every log2-domain score of
                // the half is below -128 (q . b_k is softmax-invariant, so nothing bounds it) and 0 * inf would poison the row
                const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                    sc[r] -= delta;
                }
                l_run *= alpha;
                first = false;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]);
            {
                float ps0 = sc[0] + sc[1], ps1 = sc[2] + sc[3];
#pragma unroll
                for (int r = 4; r < 16; r += 4) {
                    ps0 += sc[r] + sc[r + 1];
                    ps1 += sc[r + 2] + sc[r + 3];
                }
                l_run += ps0 + ps1;
            }
            }
        pv:
            if (S3_PROBE(p, 4)) {
                l_run += sc[0] + sc[5] + sc[11];
                return;
            }
            // P^T as B operand: step u uses regs 8u..8u+7  <->  keys 32h + 16u + {0,1,2,3,8,9,10,11} + 4*half
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint4 pf;
                pf.x = Cvt<T>::pack2(sc[8 * u + 0], sc[8 * u + 1]);
                pf.y = Cvt<T>::pack2(sc[8 * u + 2], sc[8 * u + 3]);
                pf.z = Cvt<T>::pack2(sc[8 * u + 4], sc[8 * u + 5]);
                pf.w = Cvt<T>::pack2(sc[8 * u + 6], sc[8 * u + 7]);
                uint2 a00, a01, a10, a11;
                if (ATTN_VTR) {
                    const u16* v0 = vt + (32 * h + 16 * u) * VSR16;
                    a00 = lds_tr16(v0), a01 = lds_tr16(v0 + 8 * VSR16);
                    a10 = lds_tr16(v0 + 32), a11 = lds_tr16(v0 + 8 * VSR16 + 32);
                } else {
                    const u16* v0 = vt + 32 * h + 16 * u;
                    a00 = *(const uint2*)(v0), a01 = *(const uint2*)(v0 + 8);
                    a10 = *(const uint2*)(v0 + 32 * VS16), a11 = *(const uint2*)(v0 + 32 * VS16 + 8);
                }
                o0 = Mma16<T>::run(make_uint4(a00.x, a00.y, a01.x, a01.y), pf, o0);
                o1 = Mma16<T>::run(make_uint4(a10.x, a10.y, a11.x, a11.y), pf, o1);
            }
        };
        f32x16 sc = scores(0);
        half_step(sc, 0);
#if S3_ATTN_EXP & 16
        if (FULL && !S3_PROBE(p, 1)) store_tile((kt + 1) & 1);  // (lab: the staged tile goes to its LDS buffer between the halves)
#endif
        if (two) {
            sc = scores(1);
            half_step(sc, 1);
        }
    };

    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + 1 < ntiles;
        if (more) {
            if (kt + 2 == ntiles) {
                set_ptrs(kt + 1);  // the last tile: rows may run past the last frame
            } else {
                kp0 += tstride;
                kp1 += tstride;
                vp0 += tstride;
                vp1 += tstride;
            }
            if (!S3_PROBE(p, 1)) load_tile();
#if S3_ATTN_EXP & 1
            __builtin_amdgcn_sched_barrier(0);  // (lab: the next tile's loads stay in front of this tile's work)
#endif
#if S3_ATTN_EXP & 8
            __builtin_amdgcn_s_setprio(1);      // (lab: the multiplying wave ahead of its SIMD partners' loads and stores)
#endif
            tile(std::true_type{}, kt);
#if S3_ATTN_EXP & 8
            __builtin_amdgcn_s_setprio(0);
#endif
#if !(S3_ATTN_EXP & 16)
            if (!S3_PROBE(p, 1)) store_tile((kt + 1) & 1);
#endif
        } else {
            tile(std::false_type{}, kt);
        }
        if (!S3_PROBE(p, 16)) __syncthreads();
    }
    const float inv = 1.f / xhalf_sum(l_run);
    if (p.out_f32) {  // (workgroup-uniform)
        if (q_g < p.T) {
            float* op = (float*)p.out + ((long)b * p.T + q_g) * D + head * HD + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *(float4*)(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                *(float4*)(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            }
        }
        return;
    }
    if (q_g < p.T) store_o16<T>((u16*)p.out + ((long)b * p.T + q_g) * D + head * HD, half, o0, o1, inv);
}

// ---- round 6: the persistent form of the 16-bit kernel ----------------------------------------------------------------------
// Same tile body as attn_h16_kernel (bit-identical results: the per-accumulator order of every MFMA and every softmax update is
// unchanged), different life cycle.  The one-shot grid is phase-locked: all 768 resident workgroups load their Q and first K / V
// tile in the same microseconds (the memory system saturated, the matrix pipe idle), multiply in the same microseconds (the other
// way round) and store together; a second round repeats it — an EMPTY key loop was 21.6 of 46.5 us (profiles/r05_attn_lab.md), of
// which ~15 us is simply the 74 MB of Q / first tiles / outputs moving while nothing multiplies.  Here 3 (BIAS: 2) workgroups per
// CU stay resident and walk the (batch, head, query block) items of their XCD; the NEXT item's first K / V tile is loaded into the
// staging registers in front of the current item's LAST tile, its Q fragments right behind that tile, and both land under the
// current item's last MFMAs, its normalisation and its output stores: after the first item a workgroup never waits for a cold
// prologue again, and the output stores of item i drain under the key loop of item i + 1.
// XCD-aware item order (as attn_work): XCD x owns the units u = x (mod 8); its workgroups take that XCD's items (unit-major, the
// query blocks of a unit adjacent) round-robin, so the co-resident workgroups of an XCD work on the same few units and K / V
// come from HBM once.
template <typename T, bool BIAS>
__global__ __launch_bounds__(256, attn_h16_waves(BIAS)) void attn_h16p_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) u16 Ks[2 * KBUF16];
    __shared__ __attribute__((aligned(16))) u16 Vt[2 * VBUF16];
    extern __shared__ float bias_s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = p.H * HD;
    const long ld = 3L * D;
    const int nqb = (p.T + QT - 1) / QT;
    const int xcd = blockIdx.x & 7, wpx = gridDim.x >> 3;  // (the launcher's grid is a multiple of 8)
    const int units = p.B * p.H;
    const int n_items = (units > xcd ? (units - xcd + 7) / 8 : 0) * nqb;
    int it = blockIdx.x >> 3;
    if (it >= n_items) return;

    // staging roles: K — rows (tid>>3) and +32, 16-byte chunk tid&7;  V — key pair tid&31, dim group tid>>5.
    // Loads go through a buffer descriptor of the item's (batch, head) slab — base and extent in SGPRs — with ONE constant 32-bit byte
    // offset per lane and load; the tile index rides in the scalar offset.  (The one-shot kernel's four 64-bit row pointers, turned
    // into per-tile pointer chains and hoisted out of the item loop together with every clamped row x stride product, cost this
    // kernel 99 spilled registers.)  Rows past the last frame are out of the descriptor's range and read as ZERO — their scores are
    // masked to -inf, so P = 0 meets V = 0; the one-shot kernel re-reads the last frame for the same purpose.
    const int krow = tid >> 3, kc8 = tid & 7;
    const int vj = tid & 31, vdg = tid >> 5;
    u32x4 kreg[2], vreg[2];
    const int ldi = 3 * D;
    const unsigned tstride_b = (unsigned)(KT16 * ldi * 2);                       // bytes between consecutive tiles
    const unsigned ko0 = (unsigned)((krow * ldi + D + kc8 * 8) * 2), ko1 = ko0 + (unsigned)(32 * ldi * 2);
    const unsigned vo0 = (unsigned)((2 * vj * ldi + 2 * D + vdg * 8) * 2), vo1 = vo0 + (unsigned)(ldi * 2);
    auto load_tile = [&](const __amdgpu_buffer_rsrc_t& rs, int kt) {
        const unsigned so = (unsigned)kt * tstride_b;
        kreg[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, ko0, so, 0);
        kreg[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, ko1, so, 0);
        vreg[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo0, so, 0);
        vreg[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo1, so, 0);
    };
    u16* const ks_st = Ks + krow * KS16 + kc8 * 8;
    u16* const vt_st = Vt + (vdg * 8) * VS16 + 2 * vj;
    auto store_tile = [&](int buf) {
        u16* ks_ = ks_st + buf * KBUF16;
        u16* vt_ = vt_st + buf * VBUF16;
        *(u32x4*)ks_ = kreg[0];
        *(u32x4*)(ks_ + 32 * KS16) = kreg[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned a_ = vreg[0][i], b_ = vreg[1][i];
            *(unsigned*)(vt_ + (2 * i) * VS16) = (a_ & 0xffffu) | (b_ << 16);
            *(unsigned*)(vt_ + (2 * i + 1) * VS16) = (a_ >> 16) | (b_ & 0xffff0000u);
        }
    };
    const u16* const kf_rd = Ks + l31 * KS16 + 8 * half;
    const u16* const vf_rd = Vt + l31 * VS16 + 4 * half;

    // ---- the current item ----
    struct Item {
        const u16* base;  // this (batch, head)'s q row 0
        int b, head, qb;
    };
    auto rsrc_of = [&](const Item& w) {  // rows 0 .. T - 1 of the item's (batch, head): [q | k | v] at a stride of 3D elements
        return __builtin_amdgcn_make_buffer_rsrc((void*)w.base, 0, (p.T * ldi - w.head * HD) * 2, 0x00020000);
    };
    auto item_of = [&](int i) {
        Item w;
        const int unit = xcd + 8 * (i / nqb);
        w.qb = i % nqb;
        w.b = unit / p.H;
        w.head = unit % p.H;
        w.base = (const u16*)p.qkv + (long)w.b * p.T * ld + w.head * HD;
        return w;
    };
    Item cur = item_of(it);
    uint4 qf[4];
    int q_g, q_c, valid, ntiles, bias_off;
    float gate2 = 1.44269504088896340736f;
    const float* btab = nullptr;
    auto enter_scalars = [&](const Item& w) {  // everything of an item but its Q fragments and its first tile
        q_g = w.qb * QT + wave * 32 + l31;
        q_c = q_g < p.T ? q_g : p.T - 1;
        valid = p.valid[w.b];
        ntiles = (valid + KT16 - 1) / KT16;
        bias_off = w.qb * QT + QT - 1;  // window index = (key - query) + bias_off
    };
    auto load_q = [&](const __amdgpu_buffer_rsrc_t& rs) {
        const unsigned qo = (unsigned)((q_c * ldi + 8 * half) * 2);
#pragma unroll
        for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, qo + st * 32, 0, 0));
    };
    auto enter_bias = [&](const Item& w) {  // WavLM: the item's window of its head's table (see attn_h16_kernel) + the query's gate
        if (BIAS && p.bias_table) {
            const int R = p.table_R;
            const float* src = p.bias_table + (long)w.head * (2 * R + 1) + R;
            const int rel0 = -(w.qb * QT + QT - 1);
            for (int i = threadIdx.x; i < p.T + QT - 1 + BIAS_PAD; i += 256) bias_s[i] = src[min(max(rel0 + i, -R), R)];
            btab = bias_s;
            gate2 = (p.gate ? p.gate[((long)w.b * p.H + w.head) * p.T + q_c] : 1.f) * 1.44269504088896340736f;
        }
    };
    enter_scalars(cur);
    __amdgpu_buffer_rsrc_t rs_cur = rsrc_of(cur);
    load_q(rs_cur);
    enter_bias(cur);
    load_tile(rs_cur, 0);
    store_tile(0);
    __syncthreads();
    int buf = 0;  // the LDS buffer the next tile to multiply sits in (runs on across items)

    f32x16 o0, o1;
    RefM<attn_ref_kind(BIAS), T> negm;
    float l_run;
    bool first;
    while (true) {
        negm.zero();
#pragma unroll
        for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
        l_run = 0.f;
        first = true;

        // one 64-key tile.  FULL: every key is valid (all tiles but the last)
        auto tile = [&](auto full_c, int kt) {
            constexpr bool FULL = decltype(full_c)::value;
            const u16* ks = kf_rd + buf * KBUF16;
            const u16* vt = vf_rd + buf * VBUF16;
            const bool two = FULL || kt * KT16 + 32 < valid;  // wave-uniform: the second half has at least one valid key
            auto scores = [&](int h) {
                f32x16 sc = negm.block();
                if (S3_PROBE(p, 8)) return sc;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const uint4 kf = *(const uint4*)(ks + h * 32 * KS16 + st * 16);
                    sc = Mma16<T>::run(kf, qf[st], sc);
                }
                return sc;
            };
            auto half_step = [&](f32x16& sc, int h) {
                const int k0_ = kt * KT16 + h * 32;
                if (BIAS && btab) {
                    const float* bb = btab + (k0_ + 4 * half - q_c + bias_off);  // branch-free, see attn_f32_kernel
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = fmaf(gate2, bb[(r & 3) + 8 * (r >> 2)], sc[r]);
                }
                if (!FULL && k0_ + 32 > valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = k0_ + crow(r, half) < valid ? sc[r] : -INFINITY;
                }
                if (S3_PROBE(p, 2)) goto pv;
                {
                const float mx = xhalf_max(row_max16(sc));
                if (__builtin_expect(first || __any(mx > 8.f), 0)) {
                    const float delta = negm.lower(first ? mx : fmaxf(mx, 0.f));  // (the step the reference really takes)
                    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);  // (first: see attn_h16_kernel)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o0[r] *= alpha;
                        o1[r] *= alpha;
                        sc[r] -= delta;
                    }
                    l_run *= alpha;
                    first = false;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]);
                {
                    float ps0 = sc[0] + sc[1], ps1 = sc[2] + sc[3];
#pragma unroll
                    for (int r = 4; r < 16; r += 4) {
                        ps0 += sc[r] + sc[r + 1];
                        ps1 += sc[r + 2] + sc[r + 3];
                    }
                    l_run += ps0 + ps1;
                }
                }
            pv:
                if (S3_PROBE(p, 4)) {
                    l_run += sc[0] + sc[5] + sc[11];
                    return;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint4 pf;
                    pf.x = Cvt<T>::pack2(sc[8 * u + 0], sc[8 * u + 1]);
                    pf.y = Cvt<T>::pack2(sc[8 * u + 2], sc[8 * u + 3]);
                    pf.z = Cvt<T>::pack2(sc[8 * u + 4], sc[8 * u + 5]);
                    pf.w = Cvt<T>::pack2(sc[8 * u + 6], sc[8 * u + 7]);
                    const u16* v0 = vt + 32 * h + 16 * u;
                    const uint2 a00 = *(const uint2*)(v0), a01 = *(const uint2*)(v0 + 8);
                    const uint2 a10 = *(const uint2*)(v0 + 32 * VS16), a11 = *(const uint2*)(v0 + 32 * VS16 + 8);
                    o0 = Mma16<T>::run(make_uint4(a00.x, a00.y, a01.x, a01.y), pf, o0);
                    o1 = Mma16<T>::run(make_uint4(a10.x, a10.y, a11.x, a11.y), pf, o1);
                }
            };
            f32x16 sc = scores(0);
            half_step(sc, 0);
            if (two) {
                sc = scores(1);
                half_step(sc, 1);
            }
        };

        const int nxt = it + wpx;
        const bool has_next = nxt < n_items;
        Item nx = cur;
        if (ntiles == 0 && has_next) {  // (an utterance without a single valid frame: nothing to multiply, the hand-over still happens)
            nx = item_of(nxt);
            rs_cur = rsrc_of(nx);
            load_tile(rs_cur, 0);
        }
        for (int kt = 0; kt < ntiles; ++kt) {
            if (kt + 1 < ntiles) {
                if (!S3_PROBE(p, 1)) load_tile(rs_cur, kt + 1);
                tile(std::true_type{}, kt);
                if (!S3_PROBE(p, 1)) store_tile(buf ^ 1);
            } else {
                if (has_next) {  // the next item's first tile rides in the staging registers under this item's last tile
                    nx = item_of(nxt);
                    rs_cur = rsrc_of(nx);
                    load_tile(rs_cur, 0);
                }
                tile(std::false_type{}, kt);
            }
            if (kt + 1 < ntiles) {
                if (!S3_PROBE(p, 16)) __syncthreads();
                buf ^= 1;
            }
        }
        // ---- seam: this item's result goes out while the next item's operands come in ----
        const float inv = 1.f / xhalf_sum(l_run);
        const bool q_ok = q_g < p.T;
        const long obase = (long)cur.b * p.T * D + cur.head * HD;  // (wave-uniform)
        const unsigned orow = (unsigned)(q_g * D);  // (q_g is the NEXT item's after enter_scalars below)
        if (has_next) {
            store_tile(buf ^ 1);  // (frees the staging registers)
            enter_scalars(nx);    // (q_g / q_c / valid / ntiles now belong to the next item)
            load_q(rs_cur);       // lands under the normalisation and the stores below
        }
        if (p.out_f32) {  // (workgroup-uniform)
            if (q_ok) {
                float* op = (float*)p.out + obase + orow + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *(float4*)(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                    *(float4*)(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
                }
            }
        } else if (q_ok) {
            store_o16<T>((u16*)p.out + obase + orow, half, o0, o1, inv);
        }
        if (!has_next) break;
        __syncthreads();  // every wave is past its last tile: the other K / V buffer (and the bias window) may be re-used
        buf ^= 1;
        it = nxt;
        cur = nx;
        if (BIAS && p.bias_table) {
            enter_bias(cur);
            __syncthreads();
        }
    }
}

// WavLM gate from the layer input split into heads (wavlm/modules.py:535-549):
//   g = sigmoid( sum4( grep_linear(x_head) ) ) -> (a, b);  gate = a * (b * grep_a[h] - 1) + 2
// ---- fp32x3 mode (S3ENC_F32X3): fp32 q|k|v in, fp32 out, both matrix products on split bf16 operands ------------------
// S^T = K Q^T and O^T += V^T P^T are rebuilt from three bf16 MFMAs each on x = hi + lo operands (see gemm_x3.hip):
// K, V are split while they are staged into LDS (hi and lo tiles), Q once into registers, the probabilities P on the
// score registers.  Same structure as attn_h16_kernel (64 staged keys consumed as two 32-key halves, double-buffered
// LDS, register prefetch, deferred max); the softmax itself is fp32 as everywhere.  1/5 of the exact kernel's matrix time.
__global__ __launch_bounds__(256, 3) void attn_x3_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char dyn_x3[];
    u16* Ks = (u16*)dyn_x3;                       // [plane][buffer][KBUF16]
    u16* Vt = Ks + 4 * KBUF16;                    // [plane][buffer][VBUF16]
    float* bias_s = (float*)(Vt + 4 * VBUF16);    // WavLM: the workgroup's (T+127)-entry table window
    const AttnWork wk = attn_work(p);
    if (!wk.live) return;
    const int b = wk.b, head = wk.head;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int D = p.H * HD;
    const long ld = 3L * D;
    const float* base = (const float*)p.qkv + (long)b * p.T * ld + head * HD;
    const int q_g = wk.qb * QT + wave * 32 + l31;
    const int q_c = q_g < p.T ? q_g : p.T - 1;

    // Q fragments (B operand), split once: step st covers dims st*16 .. +15, this half-wave holds 8 of them
    uint4 qh[4], ql[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const float* qp = base + (long)q_c * ld + st * 16 + 8 * half;
        split8(*(const float4*)qp, *(const float4*)(qp + 4), qh[st], ql[st]);
    }
    const float* btab = nullptr;
    if (p.bias_table) {
        const int R = p.table_R;
        const float* src = p.bias_table + (long)head * (2 * R + 1) + R;
        const int rel0 = -(wk.qb * QT + QT - 1);  // smallest key - query this workgroup can meet
        for (int i = threadIdx.x; i < p.T + QT - 1 + BIAS_PAD; i += 256) bias_s[i] = src[min(max(rel0 + i, -R), R)];
        btab = bias_s;  // made visible by the first __syncthreads()
    }
    const float gate = (btab && p.gate) ? p.gate[((long)b * p.H + head) * p.T + q_c] : 1.f;
    const int bias_off = wk.qb * QT + QT - 1;  // window index = (key - query) + bias_off

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int valid = p.valid[b];
    const int ntiles = (valid + KT16 - 1) / KT16;
    const int krow = tid >> 3, kc8 = tid & 7;   // K: rows krow and krow+32, 8-dim chunk kc8
    const int vj = tid & 31, vdg = tid >> 5;    // V: key pair vj, dim group vdg
    f32x4 kreg[2][2], vreg[2][2];               // fp32 in flight; split when written to LDS
    auto clampk = [&](int kr) { return kr < p.T ? kr : p.T - 1; };
    auto f4 = [](const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); };
#define AX3_LOAD(kt_)                                                                                  \
    {                                                                                                  \
        const int k0_ = (kt_) * KT16;                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                             \
            const float* kp_ = base + (long)clampk(k0_ + krow + 32 * i_) * ld + D + kc8 * 8;           \
            kreg[i_][0] = *(const f32x4*)kp_;                                                          \
            kreg[i_][1] = *(const f32x4*)(kp_ + 4);                                                    \
            const float* vp_ = base + (long)clampk(k0_ + 2 * vj + i_) * ld + 2 * D + vdg * 8;          \
            vreg[i_][0] = *(const f32x4*)vp_;                                                          \
            vreg[i_][1] = *(const f32x4*)(vp_ + 4);                                                    \
        }                                                                                              \
    }
#define AX3_STORE(buf_)                                                                                \
    {                                                                                                  \
        uint4 kh_[2], kl_[2], vh_[2], vl_[2];                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                             \
            split8(f4(kreg[i_][0]), f4(kreg[i_][1]), kh_[i_], kl_[i_]);                                \
            split8(f4(vreg[i_][0]), f4(vreg[i_][1]), vh_[i_], vl_[i_]);                                \
            *(uint4*)(Ks + (0 * 2 + (buf_)) * KBUF16 + (krow + 32 * i_) * KS16 + kc8 * 8) = kh_[i_];   \
            *(uint4*)(Ks + (1 * 2 + (buf_)) * KBUF16 + (krow + 32 * i_) * KS16 + kc8 * 8) = kl_[i_];   \
        }                                                                                              \
        _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_) {                                          \
            u16* vt_ = Vt + (pl_ * 2 + (buf_)) * VBUF16;                                               \
            const unsigned a_[4] = {pl_ ? vl_[0].x : vh_[0].x, pl_ ? vl_[0].y : vh_[0].y, pl_ ? vl_[0].z : vh_[0].z, pl_ ? vl_[0].w : vh_[0].w}; \
            const unsigned b_[4] = {pl_ ? vl_[1].x : vh_[1].x, pl_ ? vl_[1].y : vh_[1].y, pl_ ? vl_[1].z : vh_[1].z, pl_ ? vl_[1].w : vh_[1].w}; \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                         \
                *(unsigned*)(vt_ + (vdg * 8 + 2 * i_) * VS16 + 2 * vj) = (a_[i_] & 0xffffu) | (b_[i_] << 16);          \
                *(unsigned*)(vt_ + (vdg * 8 + 2 * i_ + 1) * VS16 + 2 * vj) = (a_[i_] >> 16) | (b_[i_] & 0xffff0000u);  \
            }                                                                                          \
        }                                                                                              \
    }
    AX3_LOAD(0)
    AX3_STORE(0)
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) AX3_LOAD(kt + 1)
        const u16* ksh = Ks + (0 * 2 + (kt & 1)) * KBUF16;
        const u16* ksl = Ks + (1 * 2 + (kt & 1)) * KBUF16;
        const u16* vth = Vt + (0 * 2 + (kt & 1)) * VBUF16;
        const u16* vtl = Vt + (1 * 2 + (kt & 1)) * VBUF16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k0 = kt * KT16 + h * 32;
            if (k0 >= valid) break;  // wave-uniform: the whole half is masked
            f32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int ko = (h * 32 + l31) * KS16 + st * 16 + 8 * half;
                const uint4 kh = *(const uint4*)(ksh + ko), kl = *(const uint4*)(ksl + ko);
                sc = Mma16<bf16_tag>::run(kl, qh[st], sc);
                sc = Mma16<bf16_tag>::run(kh, ql[st], sc);
                sc = Mma16<bf16_tag>::run(kh, qh[st], sc);
            }
            if (btab) {
                {
                    const float* bb = btab + (k0 + 4 * half - q_c + bias_off);  // branch-free, see attn_f32_kernel
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = fmaf(gate, bb[(r & 3) + 8 * (r >> 2)], sc[r]);
                }
            }
            if (k0 + 32 > valid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = k0 + crow(r, half) < valid ? sc[r] : -INFINITY;
            }
            float mx = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            mx = xhalf_max(mx);
            if (__any(mx > m_run + 8.f)) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __expf(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = __expf(sc[r] - m_run);
                ps += sc[r];
            }
            l_run += ps;
            // P^T as B operand, split: step u uses regs 8u..8u+7  <->  keys 32h + 16u + {0,1,2,3,8,9,10,11} + 4*half
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint4 ph, pl;
                split8(make_float4(sc[8 * u], sc[8 * u + 1], sc[8 * u + 2], sc[8 * u + 3]),
                       make_float4(sc[8 * u + 4], sc[8 * u + 5], sc[8 * u + 6], sc[8 * u + 7]), ph, pl);
                const int vo = l31 * VS16 + 32 * h + 16 * u + 4 * half;
                auto frag = [&](const u16* vt, int off) {
                    const uint2 a0 = *(const uint2*)(vt + off), a1 = *(const uint2*)(vt + off + 8);
                    return make_uint4(a0.x, a0.y, a1.x, a1.y);
                };
                const uint4 vh0 = frag(vth, vo), vl0 = frag(vtl, vo);
                const uint4 vh1 = frag(vth, vo + 32 * VS16), vl1 = frag(vtl, vo + 32 * VS16);
                o0 = Mma16<bf16_tag>::run(vl0, ph, o0);
                o0 = Mma16<bf16_tag>::run(vh0, pl, o0);
                o0 = Mma16<bf16_tag>::run(vh0, ph, o0);
                o1 = Mma16<bf16_tag>::run(vl1, ph, o1);
                o1 = Mma16<bf16_tag>::run(vh1, pl, o1);
                o1 = Mma16<bf16_tag>::run(vh1, ph, o1);
            }
        }
        if (kt + 1 < ntiles) AX3_STORE((kt + 1) & 1)
        __syncthreads();
    }
#undef AX3_LOAD
#undef AX3_STORE
    const float l_tot = xhalf_sum(l_run);
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

// One wavefront per (b, t) row: the D floats are read once as coalesced float4s (16 lanes per head), the four summed
// grep_linear outputs of each gate half are ONE dot product with the summed weight rows (sum_o (W_o x + b_o) =
// (sum_o W_o) x + sum_o b_o), reduced over the head's 16 lanes with four shuffles.
__global__ __launch_bounds__(256) void wavlm_gate_kernel(const float* x, const float* gw, const float* gb, const float* ga,
                                                         int B, int T, int H, float* gate) {
    const int lane = threadIdx.x & 63;
    const long bt = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bt >= (long)B * T) return;
    const int k4 = (lane & 15) * 4;  // this lane's 4 dims of its head
    float wa[4], wb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wa[i] = (gw[0 * HD + k4 + i] + gw[1 * HD + k4 + i]) + (gw[2 * HD + k4 + i] + gw[3 * HD + k4 + i]);
        wb[i] = (gw[4 * HD + k4 + i] + gw[5 * HD + k4 + i]) + (gw[6 * HD + k4 + i] + gw[7 * HD + k4 + i]);
    }
    const float ba = (gb[0] + gb[1]) + (gb[2] + gb[3]), bb0 = (gb[4] + gb[5]) + (gb[6] + gb[7]);
    const float4* xr = (const float4*)(x + bt * (long)H * HD);
    const int b = (int)(bt / T), t = (int)(bt % T);
    for (int h0 = 0; h0 < H; h0 += 4) {  // 4 heads per round of 64 lanes
        const int h = h0 + (lane >> 4);
        float sa = 0.f, sb = 0.f;
        if (h < H) {
            const float4 v = xr[h0 * 16 + lane];
            sa = v.x * wa[0] + v.y * wa[1] + v.z * wa[2] + v.w * wa[3];
            sb = v.x * wb[0] + v.y * wb[1] + v.z * wb[2] + v.w * wb[3];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            sa += __shfl_xor(sa, o, 64);
            sb += __shfl_xor(sb, o, 64);
        }
        if (h < H && (lane & 15) == 0) {
            const float a = 1.f / (1.f + __expf(-(sa + ba)));
            const float bb = 1.f / (1.f + __expf(-(sb + bb0)));
            gate[((long)b * H + h) * T + t] = a * (bb * ga[h] - 1.f) + 2.f;
        }
    }
}

}  // namespace


hipError_t launch_attention(int dtype, const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return hipSuccess;
    static_assert(QT == 128, "attn_work assumes 128 queries per workgroup");
    const int units8 = (p.B * p.H + 7) / 8;
    dim3 grid((unsigned)(8 * units8 * ((p.T + QT - 1) / QT))), block(256);  // XCD-aware 1-D work map (attn_work)
    const size_t dyn = p.bias_table ? (size_t)(p.T + QT - 1 + BIAS_PAD) * sizeof(float) : 0;  // the workgroup's table window
    if (dyn > 24 * 1024) return hipErrorInvalidValue;  // T <= 6017 frames (120 s): the window must fit beside K/V (engine checks)
    switch (dtype) {
        case F32:
            if (tuning().attn_persist) {
                const long items8 = (long)units8 * ((p.T + QT - 1) / QT);  // items of the fullest XCD
                const long wpx = std::min<long>(items8, 32L * 3);          // 32 CUs per XCD, 3 resident workgroups per CU
                hipLaunchKernelGGL(attn_f32p_kernel, dim3((unsigned)(8 * wpx)), block, dyn, s, p);
            } else {
                hipLaunchKernelGGL(attn_f32_kernel, grid, block, dyn, s, p);
            }
            break;
        case BF16:
        case F16: {
            const bool bf = dtype == BF16, bias = p.bias_table != nullptr;
            if (tuning().attn_persist) {
                // persistent workgroups: 3 (BIAS: 2) per CU — what __launch_bounds__ guarantees registers for — but never more
                // than the XCD's share of items needs (a grid of idle workgroups is not free)
                const int per_cu = attn_h16_waves(bias);
                const long items8 = (long)units8 * ((p.T + QT - 1) / QT);  // items of the fullest XCD
                const long wpx = std::min<long>(items8, 32L * per_cu);     // 32 CUs per XCD
                dim3 pgrid((unsigned)(8 * wpx));
                if (bf) {
                    if (bias) hipLaunchKernelGGL((attn_h16p_kernel<bf16_tag, true>), pgrid, block, dyn, s, p);
                    else hipLaunchKernelGGL((attn_h16p_kernel<bf16_tag, false>), pgrid, block, dyn + tuning().attn_lds_pad, s, p);
                } else {
                    if (bias) hipLaunchKernelGGL((attn_h16p_kernel<f16_tag, true>), pgrid, block, dyn, s, p);
                    else hipLaunchKernelGGL((attn_h16p_kernel<f16_tag, false>), pgrid, block, dyn + tuning().attn_lds_pad, s, p);
                }
                break;
            }
            if (bf) {
                if (bias) hipLaunchKernelGGL((attn_h16_kernel<bf16_tag, true>), grid, block, dyn, s, p);
                else hipLaunchKernelGGL((attn_h16_kernel<bf16_tag, false>), grid, block, dyn + tuning().attn_lds_pad, s, p);
            } else {
                if (bias) hipLaunchKernelGGL((attn_h16_kernel<f16_tag, true>), grid, block, dyn, s, p);
                else hipLaunchKernelGGL((attn_h16_kernel<f16_tag, false>), grid, block, dyn + tuning().attn_lds_pad, s, p);
            }
            break;
        }
        case 3: {  // S3ENC_F32X3: fp32 q|k|v and output, split-precision products; all of its LDS is dynamic
            const size_t lds = (size_t)(4 * KBUF16 + 4 * VBUF16) * sizeof(u16) + dyn;
            hipError_t e = ensure_dynamic_lds<attn_x3_kernel>((int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(attn_x3_kernel, grid, block, lds, s, p);
            break;
        }
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_wavlm_gate(const float* x, const float* grep_w, const float* grep_b, const float* grep_a, int B, int T,
                             int H, float* gate, hipStream_t s) {
    const long rows = (long)B * T;
    hipLaunchKernelGGL(wavlm_gate_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, grep_w, grep_b, grep_a, B, T, H,
                       gate);
    return hipGetLastError();
}

}  // namespace s3
