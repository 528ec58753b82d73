// engine_internal.h — what the translation units of the engine share: error reporting, device buffers, the weight containers,
// the handle (struct s3enc_encoder), per-kernel profiling, the workspace allocator and the multires-HuBERT plan.
//   engine.hip    s3enc_create (weight packing), the single-resolution forward schedule, the handle's C ABI
//   multires.hip  the multires-HuBERT U-net behind post_extract_proj
//   ops.hip       single-kernel entry points (s3enc_op_*), the weighted sum, fbank, tuning keys
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/s3enc.h"
#include "kernels.h"

namespace s3e {
using namespace s3;


extern thread_local std::string g_err;  // s3enc_last_error()

inline int fail(const std::string& msg) {
    g_err = msg;
    return 1;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            char _b[512];                                                                              \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return fail(_b);                                                                           \
        }                                                                                              \
    } while (0)

// ---- host-side dtype conversion -------------------------------------------------------------------------
inline uint16_t h_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline uint16_t h_f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}
inline float h_from16(uint16_t v, int dtype) {
    if (dtype == BF16) {
        uint32_t u = ((uint32_t)v) << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    _Float16 h;
    memcpy(&h, &v, 2);
    return (float)h;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
        o.p = nullptr;
        o.bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);  // implicit device sync: nothing in flight still uses it
            p = nullptr;
            bytes = 0;
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    // Growth on the forward path, ordered on `st` instead of synchronising the device: the old block is released with
    // hipFreeAsync (it may still be read by launches already enqueued on `st`) and the new one comes from the
    // stream-ordered allocator, 25 % larger than asked so a serving loop with drifting batch shapes settles after a few
    // growths.  Falls back to the synchronising path if the runtime has no stream-ordered pool.
    hipError_t ensure_on_stream(size_t n, hipStream_t st) {
        if (n <= bytes) return hipSuccess;
        const size_t want = n + n / 4;
        void* np = nullptr;
        hipError_t e = hipMallocAsync(&np, want, st);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return ensure(want);
        }
        if (p) {
            e = hipFreeAsync(p, st);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(st);
                (void)hipFree(p);
            }
        }
        p = np;
        bytes = want;
        return hipSuccess;
    }
};

// current-device RAII: the entry points never leave the caller's (torch's) current device changed
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

inline hipError_t upload_f32(DevBuf& d, const std::vector<float>& v) {
    hipError_t e = d.ensure(v.size() * 4 + 16);
    if (e != hipSuccess) return e;
    return hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice);
}
// Host-side packers run over independent ranges on up to 16 threads (a large model's 315 M weights: 4.6 s of s3enc_create in the
// fp16x2 mode on one thread).  Thread creation can fail (pids cgroup, ulimit -u, bad_alloc) and nothing may throw across the C
// boundary: the ranges whose thread did not start run on the calling thread, the started ones are always joined.
template <typename F>
inline void parallel_ranges(long n, long min_per_thread, F&& fn) {  // fn(begin, end)
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if ((long)nt > n / std::max<long>(1, min_per_thread)) nt = (unsigned)std::max<long>(1, n / std::max<long>(1, min_per_thread));
    if (nt <= 1) {
        fn(0L, n);
        return;
    }
    std::vector<std::thread> th;
    unsigned started = 0;
    try {
        th.reserve(nt);
        for (; started < nt; ++started) th.emplace_back([&fn, n, nt, started]() { fn(n * started / nt, n * (started + 1) / nt); });
    } catch (...) {
    }
    for (unsigned t = started; t < nt; ++t) fn(n * t / nt, n * (t + 1) / nt);
    for (auto& x : th) x.join();
}

inline hipError_t upload_cvt(DevBuf& d, const std::vector<float>& v, int dtype) {
    if (dtype == F32) return upload_f32(d, v);
    std::vector<uint16_t> h(v.size());
    const float* src = v.data();
    uint16_t* dst = h.data();
    if (dtype == BF16)
        parallel_ranges((long)v.size(), 1 << 20, [=](long a, long b) { for (long i = a; i < b; ++i) dst[i] = h_bf16(src[i]); });
    else
        parallel_ranges((long)v.size(), 1 << 20, [=](long a, long b) { for (long i = a; i < b; ++i) dst[i] = h_f16(src[i]); });
    hipError_t e = d.ensure(h.size() * 2 + 16);
    if (e != hipSuccess) return e;
    return hipMemcpy(d.p, h.data(), h.size() * 2, hipMemcpyHostToDevice);
}

// Positional-conv weight (folded, [D][Dg][K] like nn.Conv1d.weight) -> the layout of the kernel of `dtype`:
//   fp32   [G][K][Dg/16][Dg(co)][16]         (posconv_kernel: tap-major 16-deep input-channel chunks; inside a row of 16 the
//                                             four 16-byte slots are stored at slot ^ pc_w_swizzle(co): the kernel's
//                                             ds_read_b128 of a (16 co) x (16 ci) fragment block is then bank-conflict free)
//   16-bit [G][Dg(co)][k = tap*Dg + ci]       (posconv16_kernel: the W operand of the implicit GEMM)
inline void pack_posconv(const std::vector<float>& w, int D, int G, int K, int dtype, std::vector<float>& out) {
    const int Dg = D / G;
    out.assign((size_t)G * K * Dg * Dg, 0.f);
    for (int gi = 0; gi < G; ++gi)
        for (int n = 0; n < Dg; ++n)
            for (int ci = 0; ci < Dg; ++ci)
                for (int k = 0; k < K; ++k) {
                    const float x = w[((long)(gi * Dg + n) * Dg + ci) * K + k];
                    if (dtype == F32)
                        out[((((long)gi * K + k) * (Dg / 16) + ci / 16) * Dg + n) * 16 + ((((ci % 16) >> 2) ^ pc_w_swizzle(n)) << 2) + (ci & 3)] = x;
                    else
                        out[(((long)gi * Dg + n) * K + k) * Dg + ci] = x;
                }
}

// S3ENC_F16X2: a GEMM weight (N, K) as fp16 rows [hi(K) | lo(K)], w = hi + lo + O(2^-22 |w|)
inline hipError_t upload_f16_hi_lo(DevBuf& d, const std::vector<float>& v, long N, long K) {
    std::vector<uint16_t> h((size_t)N * 2 * K);
    const float* src = v.data();
    uint16_t* dst = h.data();
    parallel_ranges(N, std::max<long>(1, (1 << 19) / std::max<long>(1, K)), [=](long n0, long n1) {
        for (long n = n0; n < n1; ++n)
            for (long k = 0; k < K; ++k) {
                const float w = src[(size_t)n * K + k];
                const uint16_t hi = h_f16(w);
                dst[(size_t)n * 2 * K + k] = hi;
                dst[(size_t)n * 2 * K + K + k] = h_f16(w - h_from16(hi, F16));
            }
    });
    hipError_t e = d.ensure(h.size() * 2 + 16);
    if (e != hipSuccess) return e;
    return hipMemcpy(d.p, h.data(), h.size() * 2, hipMemcpyHostToDevice);
}
// a GEMM operand weight (N, K) in the handle's operand format
inline hipError_t upload_gemm_w(DevBuf& d, const std::vector<float>& v, long N, long K, int dtype, bool x2) {
    return x2 ? upload_f16_hi_lo(d, v, N, K) : upload_cvt(d, v, dtype);
}

// S3ENC_F16X2, round 5: the MX-fp4 image of a weight's lo term (lo = w - fp16(w)) for gemm16.hip's MXW K step.  Per row and
// 32-k block: an E8M0 scale 2^e, e = ceil(log2(max|lo| / 6)) (e2m1's largest magnitude is 6), and 32 e2m1 values lo / 2^e rounded
// to nearest (ties to the even code), element i of the block in nibble i of its 16 bytes.  data: (N, K/32, 16) bytes, scales:
// (N, K/32) bytes (bias 127).
inline unsigned mx_e2m1(float x) {  // |x| <= 6 expected (larger saturates); returns the 4-bit code (sign in bit 3)
    const unsigned sign = x < 0.f ? 8u : 0u;
    const float a = x < 0.f ? -x : x;
    // codes 0..7 = 0, 0.5, 1, 1.5, 2, 3, 4, 6; midpoints 0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5 go to the even code
    unsigned c;
    if (a < 0.25f) c = 0; else if (a == 0.25f) c = 0;
    else if (a < 0.75f) c = 1; else if (a == 0.75f) c = 2;
    else if (a < 1.25f) c = 2; else if (a == 1.25f) c = 2;
    else if (a < 1.75f) c = 3; else if (a == 1.75f) c = 4;
    else if (a < 2.5f) c = 4; else if (a == 2.5f) c = 4;
    else if (a < 3.5f) c = 5; else if (a == 3.5f) c = 6;
    else if (a < 5.f) c = 6; else if (a == 5.f) c = 6;
    else c = 7;
    return sign | c;
}
inline void pack_mx4_lo_rows(const float* w, long K, long n0, long n1, uint8_t* data, uint8_t* scales) {
    const long kb = K / 32;
    for (long n = n0; n < n1; ++n)
        for (long b = 0; b < kb; ++b) {
            float lo[32], amax = 0.f;
            for (int i = 0; i < 32; ++i) {
                const float x = w[(size_t)n * K + b * 32 + i];
                lo[i] = x - h_from16(h_f16(x), F16);
                const float a = lo[i] < 0.f ? -lo[i] : lo[i];
                amax = a > amax ? a : amax;
            }
            int ex = -127;
            if (!(amax <= 3.0e38f)) {
                // a weight outside the fp16 range (hi = inf, lo = -inf) or NaN: the product is non-finite whatever this image says
                // (the forward's status word reports it); E8M0 0xFF is the format's NaN
                ex = 128;
            } else if (amax > 0.f) {
                ex = (int)ceilf(log2f(amax / 6.f));
                while (ex < 127 && ldexpf(6.f, ex) < amax) ++ex;  // (log2f rounding: the scaled maximum must not exceed 6)
                ex = ex < -127 ? -127 : (ex > 127 ? 127 : ex);
            }
            const float inv = ex > 127 ? 0.f : ldexpf(1.f, -ex);
            uint8_t* d = data + ((size_t)n * kb + b) * 16;
            for (int i = 0; i < 32; ++i) d[i >> 1] |= (uint8_t)(mx_e2m1(lo[i] * inv) << ((i & 1) * 4));
            scales[(size_t)n * kb + b] = (uint8_t)(ex + 127);
        }
}
// (rows are independent: a large model's 277 M weights take 6.6 s on one host thread, under a second on eight)
inline void pack_mx4_lo(const std::vector<float>& w, long N, long K, std::vector<uint8_t>& data, std::vector<uint8_t>& scales) {
    const long kb = K / 32;
    data.assign((size_t)N * kb * 16, 0);
    scales.assign((size_t)N * kb, 0);
    const float* src = w.data();
    uint8_t *dd = data.data(), *ds = scales.data();
    parallel_ranges(N, 64, [=](long n0, long n1) { pack_mx4_lo_rows(src, K, n0, n1, dd, ds); });
}
struct MxImage {
    DevBuf data, scales;
    long N = 0, K = 0;
    int kind = 0;  // which GEMM of the path the weight belongs to: a bit of the tuning key gemm16_mx (1 conv, 2 q|k|v, 4 fc1, 8 fc2)
};
inline hipError_t upload_mx4_lo(MxImage& m, const std::vector<float>& v, long N, long K) {
    std::vector<uint8_t> d, s;
    pack_mx4_lo(v, N, K, d, s);
    hipError_t e = m.data.ensure(d.size() + 256);
    if (e != hipSuccess) return e;
    e = m.scales.ensure(s.size() + 256);
    if (e != hipSuccess) return e;
    m.N = N;
    m.K = K;
    e = hipMemcpy(m.data.p, d.data(), d.size(), hipMemcpyHostToDevice);
    return e != hipSuccess ? e : hipMemcpy(m.scales.p, s.data(), s.size(), hipMemcpyHostToDevice);
}

// S3ENC_F32X3: upload the pair-packed bf16 hi / lo image of an (N, K) fp32 weight (K % 32 == 0, else left empty: that
// GEMM then runs on the exact kernel)
inline hipError_t upload_x3(DevBuf& d, const std::vector<float>& v, long N, long K) {
    if (K % 32 || (long)v.size() < N * K) return hipSuccess;
    std::vector<uint16_t> pk;
    pack_x3(v.data(), N, K, pk);
    hipError_t e = d.ensure(pk.size() * 2 + 16);
    if (e != hipSuccess) return e;
    return hipMemcpy(d.p, pk.data(), pk.size() * 2, hipMemcpyHostToDevice);
}

// S3ENC_F32X3 positional conv: the 16-bit layout [G][Dg][K*Dg] as a bf16 hi image followed by the lo image
inline hipError_t upload_posconv_x3(DevBuf& d, const std::vector<float>& w, int D, int G, int K) {
    std::vector<float> lay;
    pack_posconv(w, D, G, K, BF16, lay);
    std::vector<uint16_t> img(lay.size() * 2);
    for (size_t i = 0; i < lay.size(); ++i) {
        const uint16_t h = h_bf16(lay[i]);
        img[i] = h;
        img[lay.size() + i] = h_bf16(lay[i] - h_from16(h, BF16));
    }
    hipError_t e = d.ensure(img.size() * 2 + 16);
    if (e != hipSuccess) return e;
    return hipMemcpy(d.p, img.data(), img.size() * 2, hipMemcpyHostToDevice);
}

struct LayerW {
    DevBuf wqkv, bqkv, wo, bo, ln1g, ln1b, w1, b1, w2, b2, ln2g, ln2b;
    DevBuf wqkv3, wo3, w13, w23;  // S3ENC_F32X3: pair-packed bf16 hi / lo images of the four weight matrices
    DevBuf grep_w, grep_b, grep_a;
};
struct ConvW {
    DevBuf w, bias, lng, lnb;  // w: conv0 fp32 [C][k]; conv>=1 compute dtype [C][k*Cin]
    DevBuf w3;                 // S3ENC_F32X3: pair-packed image of w (conv >= 1)
    bool has_bias = false;
};

// multires-HuBERT (multires_hubert/hubert_model.py:337-530): one TransformerEncoder of the U-net, and a conv adapter
struct BlockW {
    std::vector<LayerW> layers;
    DevBuf eln_g, eln_b;  // the block's own encoder.layer_norm
};
struct AdapterConvW {
    DevBuf w, w3;  // the convolution as a GEMM operand (N, K) in the compute dtype (+ the S3ENC_F32X3 image)
    DevBuf g, b;   // Fp32GroupNorm(1, D) affine
};
struct AdapterW {
    AdapterConvW up, down;  // ConvTranspose1d(stride = up_rate) / Conv1d(stride = down_rate); plain variants hold one
    int kind = 0;           // 0 ConvAdapter (both), 1 ConvDownsampler, 2 ConvUpsampler
    int up_rate = 1, down_rate = 1;
};

struct ProfRec {
    int kind;
    hipEvent_t a, b;
};

}  // namespace s3e

using namespace s3e;  // (internal header: only the engine's own translation units include it)

struct s3enc_encoder {
    // (the names below are s3e:: / s3:: types)
    s3enc_config cfg;
    int device = 0;
    int dtype = F32;
    bool x3 = false;  // S3ENC_F32X3
    bool x2 = false;  // S3ENC_F16X2: fp16 data flow, GEMM weights as [hi | lo] fp16 halves (GemmParams.wsplit)
    // S3ENC_F16X2 on a GroupNorm extractor (base models: conv1..6 are GELU-only, nothing renormalises between them): from
    // this conv layer on, the stack runs on fp32 ACTIVATIONS through the three-term GEMM (gemm_x3.hip) instead of rounding
    // every layer's output to fp16 — six stacked roundings are the largest single term of the mode's error on
    // released-checkpoint statistics (tools/fp16_error_budget.py, profiles/r04_fp16_error_budget.md); 0 = off
    int x2_conv_f32_from = 0;
    // S3ENC_F16X2: post_extract_proj reads the fp32 LayerNorm(C) output through the three-term GEMM (0.4 % of the path's FLOPs;
    // the rounding of its operand is the third-largest term of the mode's error budget)
    bool x2_proj_f32 = false;
    // S3ENC_F16X2: the attention output stays fp32 and out_proj reads it through the three-term GEMM (4 % of the path's FLOPs at
    // 1.45x their two-term cost): the rounding of out_proj's operand is the largest non-conv term of the mode's error budget on
    // released-checkpoint statistics (profiles/r04_fp16_error_budget.md)
    bool x2_attn_f32 = false;
    int es = 4;  // element size of the compute dtype
    std::vector<ConvW> conv;
    DevBuf gn_g, gn_b;
    DevBuf fln_g, fln_b, proj_w, proj_b, pos_w, pos_b, eln_g, eln_b;
    DevBuf proj_w3, pos_w3;  // S3ENC_F32X3
    std::vector<LayerW> layers;
    DevBuf rel_table;  // WavLM: [H][2R+1], entry (h, rel + R), R = max_distance (the bucket saturates there)
    int rel_R = 0;
    // data2vec positional-conv stack (cfg.pos_conv_depth > 1): per block the packed conv weight + bias; pos_k = the kernel
    // width as packed (zero taps appended so that the 16-bit implicit GEMM's k axis is a multiple of 128), pos_pad = the
    // real kernel's K / 2; ones / zeros = the affine of LayerNorm(elementwise_affine=False)
    std::vector<DevBuf> pos_ws, pos_bs;
    int pos_k = 0, pos_pad = 0;
    DevBuf ones, zeros;
    DevBuf head_w1, head_b1, head_w2, head_b2, head_w13, head_w23;  // DistilHuBERT prediction heads (+ S3ENC_F32X3 images)
    DevBuf wsum_part;  // persistent partials of s3enc_weighted_sum_backward
    std::vector<BlockW> mr_blocks;      // S3ENC_MULTIRES: encoders..., middle_encoder, decoders... (execution order)
    std::vector<AdapterW> mr_adapters;  // downsample_modules[0..R-2], then upsample_modules[0..R-2]
    DevBuf ws_mr;                       // activation workspace of the U-net behind post_extract_proj

    // S3ENC_F16X2: MX-fp4 images of the weights' lo terms, keyed by the device pointer of the [hi | lo] rows they belong to
    std::map<const void*, std::unique_ptr<MxImage>> mx_images;

    Tuning tun;               // s3enc_set_handle_tuning: this handle's own kernel-variant selection ...
    bool has_tuning = false;  // ... in force (for the calling thread) while its forward enqueues kernels

    DevBuf ws;      // activation workspace
    DevBuf small;   // tables, stats
    void* pinned = nullptr;  // host staging ring
    static constexpr int RING = 4;
    size_t slot_bytes = 0;
    hipEvent_t slot_ev[RING] = {};
    int slot_next = 0;

    std::vector<hipEvent_t> layer_events;  // caller-owned, recorded when hidden_states[l] is final

    // s3enc_forward_status (ABI 6): the device word the row kernels OR into (kernels.h, t_status) is cleared at the start of
    // every forward and copied at its end into the next slot of a small pinned ring, with an event behind the copy — so a
    // status read folds the slots whose event has fired and never touches the device; a host running more than STATUS_RING
    // forwards ahead of the GPU waits for the oldest one when it comes round to its slot
    static constexpr int STATUS_RING = 8;
    DevBuf status_dev;
    int* status_host = nullptr;  // STATUS_RING words
    hipEvent_t status_ev[STATUS_RING] = {};
    bool status_busy[STATUS_RING] = {};
    int status_next = 0;
    int status_sticky = 0;  // bits of finished forwards not yet handed to the caller
    // fold the finished slots into status_sticky; wait: block for all of them.  Returns the number still running.
    int status_collect(bool wait) {
        int running = 0;
        for (int k = 0; k < STATUS_RING; ++k) {
            const int i = (status_next + k) % STATUS_RING;  // oldest first
            if (!status_busy[i]) continue;
            const hipError_t q = wait ? hipEventSynchronize(status_ev[i]) : hipEventQuery(status_ev[i]);
            if (q == hipErrorNotReady) {
                (void)hipGetLastError();
                ++running;
                continue;
            }
            status_busy[i] = false;
            if (q == hipSuccess) status_sticky |= ((volatile int*)status_host)[i];
        }
        return running;
    }

    // profiling
    int prof = 0;  // 0 off, 1 every kernel, 2 only the GEMM launches (the dominant kernel: cheap enough for a timed region)
    std::vector<std::string> kinds;
    std::vector<double> kflops, kbytes;
    std::vector<long> klaunches;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;  // timing events are recycled across profile_reset, never created per forward twice

    // debug taps of the last forward
    struct Tap {
        const void* p;
        long elems;
        int dtype;
    };
    std::map<std::string, Tap> taps;

    ~s3enc_encoder() {
        for (auto& r : recs) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        for (auto ev : ev_pool) (void)hipEventDestroy(ev);
        for (int i = 0; i < RING; ++i)
            if (slot_ev[i]) (void)hipEventDestroy(slot_ev[i]);
        if (pinned) (void)hipHostFree(pinned);
        if (status_host) (void)hipHostFree(status_host);
        for (int i = 0; i < STATUS_RING; ++i)
            if (status_ev[i]) (void)hipEventDestroy(status_ev[i]);
    }

    int kind_id(const char* name) {
        for (size_t i = 0; i < kinds.size(); ++i)
            if (kinds[i] == name) return (int)i;
        kinds.push_back(name);
        kflops.push_back(0);
        kbytes.push_back(0);
        klaunches.push_back(0);
        return (int)kinds.size() - 1;
    }
    bool take_event(hipEvent_t* ev) {
        if (!ev_pool.empty()) {
            *ev = ev_pool.back();
            ev_pool.pop_back();
            return true;
        }
        return hipEventCreate(ev) == hipSuccess;
    }
};

namespace s3e {

struct Prof {
    s3enc_encoder* e;
    hipStream_t st;
    int idx = -1;
    Prof(s3enc_encoder* enc, hipStream_t s, const char* kind, double flops, double bytes) : e(enc), st(s) {
        if (!e || !e->prof) return;
        if (e->prof == 2 && strncmp(kind, "gemm", 4) != 0) return;
        const int k = e->kind_id(kind);
        e->kflops[k] += flops;
        e->kbytes[k] += bytes;
        e->klaunches[k] += 1;
        ProfRec r;
        r.kind = k;
        if (!e->take_event(&r.a)) return;
        if (!e->take_event(&r.b)) {
            e->ev_pool.push_back(r.a);
            return;
        }
        (void)hipEventRecord(r.a, st);
        e->recs.push_back(r);
        idx = (int)e->recs.size() - 1;
    }
    ~Prof() {
        if (idx >= 0) (void)hipEventRecord(e->recs[idx].b, st);
    }
};

// S3ENC_F16X2 hybrids: would the three-term GEMM take an (N, K) product whose A rows are `lda` fp32 elements apart?  The REAL
// predicate (gemm_x3_eligible) on a representative call — 256-byte aligned operands as the workspace allocator hands them out, a
// dense fp32 output — instead of a hand-copied subset of it: s3enc_create decides the hybrids with this, so the forward's
// "not a shape of the three-term GEMM" failures cannot be reached through a drift between two predicates (round-4 ADVICE).
inline bool x3_shape_ok(long N, long K, long lda) {
    GemmParams g{};
    g.A = (const void*)(uintptr_t)256;
    g.W_x3 = (const void*)(uintptr_t)256;
    g.out32 = (float*)(uintptr_t)256;
    g.lda = lda;
    g.a_bs = 4 * lda;  // any multiple of lda: the extractor's batch stride is L * C, the encoder's GEMMs are one batch
    g.M = 256;
    g.N = (int)N;
    g.K = (int)K;
    g.batches = 1;
    g.ldo = N;
    g.o_bs = 256 * N;
    return gemm_x3_eligible(g);
}

// S3ENC_F16X2: every GEMM of the handle runs on its [hi | lo] weight rows — with the lo term as an MX-fp4 image where one was
// packed for exactly this weight (round 5: gemm16.hip MXW; launch_gemm falls back to the two-term loop for other shapes)
inline GemmParams wsplit_of(const s3enc_encoder* e, GemmParams g) {
    g.wsplit = e->x2 ? 1 : 0;
    if (e->x2) {
        auto it = e->mx_images.find(g.W);
        if (it != e->mx_images.end() && it->second->N == g.N && it->second->K == g.K && (tuning().gemm16_mx & it->second->kind)) {
            g.W4 = it->second->data.p;
            g.W4s = it->second->scales.p;
            g.mxw = 1;
        }
    }
    return g;
}

long conv_len(const s3enc_config& c, long n, int upto /*exclusive*/);  // frames after the first `upto` conv layers
int valid_frames(const s3enc_config& c, long length, long n_max);      // un-masked frames under the family's mask rule

// ---- multires-HuBERT frame geometry (mirrors EncoderConfig.multires_plan in s3prl_amd/config.py) ----------------
struct MrBlockPlan {
    int layers;
    long T;       // frames the block runs on
    int factor;   // repeat_interleave factor of its states (multires_hubert/expert.py:41-47)
    int adapter;  // index into mr_adapters of the conv adapter applied before the block, -1 for the first
    long T_in;    // frames entering that adapter
    long T_sum;   // decoders: min(T, residual frames) of align_size_sum (hubert_model.py:777-783)
};
struct MrPlan {
    std::vector<MrBlockPlan> blocks;
    long T_out = 0;  // common length every (repeated) state is cut to (expert.py:93-101)
};
void mr_plan(const s3enc_config& c, long T0, MrPlan& plan);
long output_frames(const s3enc_config& c, long n_samples);  // frames of the states a forward writes

// bump allocator over the workspace
struct Bump {
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base((char*)b) {}
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    }
};

struct FwdOpts {
    int selection = S3ENC_SEL_HIDDEN;
    int out_dtype = F32;
    bool featurize = false;
    int feat_norm = 0;
    const float* w = nullptr;  // host, one per state
};
int num_states(const s3enc_config& c, int selection);
// multires.hip: the U-net behind post_extract_proj (xproj: (B, T0, D) fp32, padded frames zero)
int multires_tail(s3enc_handle e, hipStream_t st, int B, const MrPlan& plan, const std::vector<const int*>& d_valid, float* xproj,
                  void* out, long layer_stride, const FwdOpts& fo);

}  // namespace s3e
