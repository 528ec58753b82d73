// ops.hip — the library-level entry points that are not tied to a handle: tuning keys, the single-kernel entry points the
// op-level tests and micro-benchmarks call (s3enc_op_*), the Featurizer's weighted sum and the fbank upstream.
#include "engine_internal.h"

namespace s3 {
Tuning g_tuning;
thread_local const Tuning* t_tuning = nullptr;
thread_local int* t_status = nullptr;

int tuning_set(Tuning& t, const char* key, int value, const char** err) {
    struct Key {
        const char* name;
        int Tuning::*field;
        int lo, hi;
    };
    static const Key keys[] = {
        {"gemm_variant", &Tuning::gemm_variant, 0, 63},      {"gemm_lds_pad", &Tuning::gemm_lds_pad, 0, 120 * 1024},
        {"gemm32_big", &Tuning::gemm32_big, 0, 5},           {"gemm_x3_tile", &Tuning::gemm_x3_tile, 0, 5},
        {"gemm16_big", &Tuning::gemm16_big, 0, 10},          {"gemm16_rows", &Tuning::gemm16_rows, 0, 1},
        {"attn_lds_pad", &Tuning::attn_lds_pad, 0, 48 * 1024}, {"conv0_nt", &Tuning::conv0_nt, 0, 1},
        {"ws_inplace", &Tuning::ws_inplace, 0, 1},           {"gemm16_pp", &Tuning::gemm16_pp, 0, 3},
        {"gemm16_mx", &Tuning::gemm16_mx, 0, 31},            {"reserve_cus", &Tuning::reserve_cus, 0, 128},
        {"x3_pack_cache", &Tuning::x3_pack_cache, 0, 1},     {"gelu32", &Tuning::gelu32, 0, 1},
        {"comm_self_p2p", &Tuning::comm_self_p2p, 0, 1},     {"attn_persist", &Tuning::attn_persist, 0, 1},
        {"fp16x2_conv1_f32", &Tuning::fp16x2_conv1_f32, 0, 1}, {"gn_lag_one_block", &Tuning::gn_lag_one_block, 0, 1},
        {"ln_rows", &Tuning::ln_rows, 1, 2},                 {"ln1_fold", &Tuning::ln1_fold, 0, 1},
        {"ln_preload", &Tuning::ln_preload, 0, 1},           {"forward_chain", &Tuning::forward_chain, 0, 1},
        {"conv0_fast", &Tuning::conv0_fast, 0, 1},
    };
    static thread_local char msg[160];
    if (!key) {
        *err = "s3enc_set_tuning: null key";
        return 1;
    }
    for (const Key& k : keys)
        if (!strcmp(key, k.name)) {
            if (value < k.lo || value > k.hi) {
                snprintf(msg, sizeof(msg), "%s must be %d..%d", k.name, k.lo, k.hi);
                *err = msg;
                return 1;
            }
            t.*(k.field) = value;
            return 0;
        }
    snprintf(msg, sizeof(msg), "s3enc_set_tuning: unknown key '%s'", key);
    *err = msg;
    return 1;
}
}  // namespace s3

extern "C" {

int s3enc_set_tuning(const char* key, int32_t value) {
    const char* err = nullptr;
    if (tuning_set(g_tuning, key, value, &err)) return fail(err);
    return 0;
}
int s3enc_set_handle_tuning(s3enc_handle h, const char* key, int32_t value) {
    if (!h) return fail("null handle");
    if (!h->has_tuning) {
        h->tun = g_tuning;  // start from the process defaults as they are now
        h->has_tuning = true;
    }
    const char* err = nullptr;
    if (tuning_set(h->tun, key, value, &err)) return fail(err);
    return 0;
}

// ---- single-kernel entry points -------------------------------------------------------------------------------
int s3enc_op_gemm(int32_t dtype, const void* A, int64_t lda, int64_t a_batch_stride, const void* W, const float* bias, int32_t M,
                  int32_t N, int32_t K, int32_t batches, int32_t act, const float* residual, const int32_t* row_limit,
                  float* out32, void* out16, int64_t ldo, int64_t o_batch_stride, void* stream) {
    GemmParams g{};
    g.A = A;
    g.lda = lda;
    g.a_bs = a_batch_stride;
    g.W = W;
    g.bias = bias;
    g.M = M;
    g.N = N;
    g.K = K;
    g.batches = batches;
    g.act = act;
    g.residual = residual;
    g.row_limit = row_limit;
    g.out32 = out32;
    g.out16 = out16;
    g.ldo = ldo;
    g.o_bs = o_batch_stride;
    if (dtype == 4) {  // S3ENC_F16X2: fp16 A, W given as (N, 2K) fp16 rows [hi(K) | lo(K)] — the contraction runs over both
        g.wsplit = 1;
        HIP_TRY(launch_gemm(F16, g, (hipStream_t)stream));
        return 0;
    }
    if (dtype == 5) {  // S3ENC_F16X2 with the lo term on the MX pipe (test / lab entry): fp16 A, W given as FP32 (N, K) on the device;
                       // the [hi | lo] rows and the MX-fp4 image are packed here exactly as s3enc_create packs them
        std::vector<float> hw((size_t)N * K);
        HIP_TRY(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
        DevBuf w2;
        MxImage img;
        HIP_TRY(upload_gemm_w(w2, hw, N, K, F16, true));
        HIP_TRY(upload_mx4_lo(img, hw, N, K));
        g.W = w2.p;
        g.wsplit = 1;
        g.W4 = img.data.p;
        g.W4s = img.scales.p;
        g.mxw = 1;
        Tuning forced = tuning();
        forced.gemm16_mx = 31;  // every shape the kernel can take (the engine's default also weighs the tile rounds)
        TuningScope force(&forced);
        if (!gemm16_mx_eligible(F16, g)) return fail("s3enc_op_gemm: shape / alignment not eligible for the MX second-term kernel (K % 128, N >= 128, 16-byte rows)");
        HIP_TRY(launch_gemm(F16, g, (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // the packed images are freed on return
        return 0;
    }
    if (dtype == 3) {  // S3ENC_F32X3: fp32 operands; W is split into its pair-packed bf16 hi / lo image here
        if (K % 32) return fail("s3enc_op_gemm: S3ENC_F32X3 needs K % 32 == 0");
        // tuning key "x3_pack_cache" (micro-benchmarks only): keep the packed image of the last (W, N, K) and skip the
        // host round trip + synchronisation when the same weight pointer comes back
        static DevBuf cached;
        static const void* cached_w = nullptr;
        static long cached_n = 0, cached_k = 0;
        if (tuning().x3_pack_cache && cached_w == W && cached_n == N && cached_k == K) {
            g.W_x3 = cached.p;
            if (!gemm_x3_eligible(g)) return fail("s3enc_op_gemm: shape / alignment not eligible for the S3ENC_F32X3 kernel");
            HIP_TRY(launch_gemm(F32, g, (hipStream_t)stream));
            return 0;
        }
        std::vector<float> hw((size_t)N * K);
        HIP_TRY(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
        DevBuf w3;
        HIP_TRY(upload_x3(w3, hw, N, K));
        g.W_x3 = w3.p;
        if (!gemm_x3_eligible(g)) return fail("s3enc_op_gemm: shape / alignment not eligible for the S3ENC_F32X3 kernel");
        HIP_TRY(launch_gemm(F32, g, (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // w3 is freed on return (or kept as the cache)
        if (tuning().x3_pack_cache) {
            HIP_TRY(hipDeviceSynchronize());
            std::swap(cached.p, w3.p);
            std::swap(cached.bytes, w3.bytes);
            cached_w = W;
            cached_n = N;
            cached_k = K;
        }
        return 0;
    }
    HIP_TRY(launch_gemm(dtype, g, (hipStream_t)stream));
    return 0;
}

// Measurement hook: `workgroups` workgroups of `threads` threads that do nothing but hold their CU slots for `milliseconds` on `stream`
// — a stand-in for the channel kernels of a collective running beside the encoder (bench.py --steal-cus; there is one GPU per box on
// this pool, so the real RCCL exchange cannot be timed against the compute it overlaps)
namespace {
__global__ void occupy_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
int s3enc_debug_occupy_cus(int32_t workgroups, int32_t threads, double milliseconds, void* stream) {
    // (the hold is bounded: a typo in `milliseconds` must not be able to hang the GPU for an arbitrary time)
    if (workgroups <= 0 || threads <= 0 || threads > 1024 || !(milliseconds >= 0) || milliseconds > 10000.0)
        return fail("s3enc_debug_occupy_cus: bad argument (workgroups > 0, 0 < threads <= 1024, 0 <= milliseconds <= 10000)");
    int dev = 0, khz = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;  // 100 MHz
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(threads), 0, (hipStream_t)stream, (long long)(milliseconds * khz));
    HIP_TRY(hipGetLastError());
    return 0;
}

// Measurement hook: the shader clock a timed region really ran at.  s_memtime ticks once per shader cycle, s_memrealtime at the
// constant reference rate (hipDeviceAttributeWallClockRate, 100 MHz): two samples on the SAME stream around a region give its
// average clock as d(shader) / d(reference) x rate — no concurrent probe, nothing perturbed.  Every XCD has its own counters, so the
// sample is taken by a workgroup that finds itself on XCD 0 (64 workgroups go round-robin over the 8 XCDs; the first claims).
namespace {
__global__ void clock_sample_kernel(unsigned long long* out, unsigned long long ref_khz) {
    if (threadIdx.x) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID, bits 3:0
    if (xcc != 0) return;
    if (atomicCAS(&out[2], 0ull, ref_khz) != 0ull) return;
    out[0] = __builtin_amdgcn_s_memtime();      // shader cycles
    out[1] = __builtin_amdgcn_s_memrealtime();  // reference ticks
    __threadfence_system();
}
}  // namespace
int s3enc_debug_clock_sample(uint64_t* out3_device, void* stream) {
    if (!out3_device) return fail("s3enc_debug_clock_sample: null argument");
    int dev = 0, khz = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;  // 100 MHz
    HIP_TRY(hipMemsetAsync(out3_device, 0, 3 * sizeof(uint64_t), (hipStream_t)stream));
    hipLaunchKernelGGL(clock_sample_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out3_device,
                       (unsigned long long)khz);
    HIP_TRY(hipGetLastError());
    return 0;
}

int s3enc_op_layernorm(int32_t dtype, const float* x, const float* gamma, const float* beta, int32_t rows, int32_t C, int32_t act,
                       float* out32, void* out16, void* stream) {
    HIP_TRY(launch_layernorm(dtype, x, gamma, beta, rows, C, act, out32, out16, (hipStream_t)stream));
    return 0;
}

int s3enc_op_attention(int32_t dtype, const void* qkv, void* out, const int32_t* valid, int32_t B, int32_t T, int32_t H,
                       const float* bias_table, int32_t table_R, const float* gate, void* stream) {
    if (bias_table && (table_R < 0 || T > 6000)) return fail("s3enc_op_attention: bad table_R / T > 6000 with a bias table");
    AttnParams a{};
    a.qkv = qkv;
    a.out = out;
    a.valid = valid;
    a.B = B;
    a.T = T;
    a.H = H;
    a.bias_table = bias_table;
    a.table_R = table_R;
    a.gate = gate;
    HIP_TRY(launch_attention(dtype, a, (hipStream_t)stream));
    return 0;
}

int s3enc_op_conv0(int32_t dtype, const float* const* wavs, const int64_t* lengths, int32_t B, int64_t n_max, int32_t normalize,
                   const float* w0, const float* bias, const float* gn_gamma, const float* gn_beta, const float* ln_gamma,
                   const float* ln_beta, int32_t C, int32_t stride, void* out, void* stream) {
    if (!wavs || !lengths || !w0 || !out) return fail("s3enc_op_conv0: null argument");
    if (B <= 0 || C <= 0 || (C % 32) || C > 1024 || stride <= 0) return fail("s3enc_op_conv0: bad shape");
    if (dtype < 0 || dtype > 2) return fail("s3enc_op_conv0: dtype must be S3ENC_F32 / BF16 / F16");
    if ((gn_gamma != nullptr) == (ln_gamma != nullptr)) return fail("s3enc_op_conv0: pass exactly one of gn_gamma / ln_gamma");
    const int k0 = 10;
    long nm = 0;
    for (int b = 0; b < B; ++b) nm = lengths[b] > nm ? lengths[b] : nm;
    if (n_max > 0 && n_max < nm) return fail("s3enc_op_conv0: n_max is smaller than the longest utterance");
    if (n_max > 0) nm = n_max;
    if (nm < k0) return fail("s3enc_op_conv0: input shorter than the kernel");
    const long L0 = (nm - k0) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    // scratch: table (ptrs, lens), per-utterance norm, per-(b, c) GroupNorm affine, reduction partials
    const size_t tbl = (size_t)B * 16, part = stats_partial_elems(B, nm);
    DevBuf buf;
    Bump sz(nullptr);
    sz.take(tbl);
    sz.take((size_t)B * sizeof(float2));
    sz.take((size_t)B * C * sizeof(float2));
    sz.take(part * 8);
    HIP_TRY(buf.ensure(sz.off + 256));
    Bump bb(buf.p);
    char* d_tbl = (char*)bb.take(tbl);
    float2* d_norm = (float2*)bb.take((size_t)B * sizeof(float2));
    float2* d_gn = (float2*)bb.take((size_t)B * C * sizeof(float2));
    double* d_part = (double*)bb.take(part * 8);
    std::vector<char> host(tbl);
    memcpy(host.data(), wavs, (size_t)B * 8);
    for (int b = 0; b < B; ++b) ((long*)(host.data() + (size_t)B * 8))[b] = (long)lengths[b];
    HIP_TRY(hipMemcpy(d_tbl, host.data(), tbl, hipMemcpyHostToDevice));
    WavTable wt{(const float* const*)d_tbl, (const long*)(d_tbl + (size_t)B * 8), B, nm};
    HIP_TRY(launch_wav_norm_stats(wt, normalize, d_part, d_norm, st));
    if (gn_gamma) HIP_TRY(launch_gn_stats(wt, d_norm, w0, gn_gamma, gn_beta, C, k0, stride, L0, d_part, nullptr, d_gn, st));
    Conv0Params p{};
    p.wav = wt;
    p.norm = d_norm;
    p.w0 = w0;
    p.bias = bias;
    p.gn = gn_gamma ? d_gn : nullptr;
    p.ln_g = ln_gamma;
    p.ln_b = ln_beta;
    p.C = C;
    p.k0 = k0;
    p.s0 = stride;
    p.L0 = L0;
    p.out = out;
    HIP_TRY(launch_conv0(dtype, p, st));
    HIP_TRY(hipStreamSynchronize(st));  // the scratch is freed on return
    return 0;
}

int s3enc_op_wavlm_gate(const float* x, const float* grep_w, const float* grep_b, const float* grep_a, int32_t B, int32_t T,
                        int32_t H, float* gate, void* stream) {
    if (!x || !grep_w || !grep_b || !grep_a || !gate) return fail("s3enc_op_wavlm_gate: null argument");
    if (B <= 0 || T <= 0 || H <= 0) return fail("s3enc_op_wavlm_gate: bad shape");
    HIP_TRY(launch_wavlm_gate(x, grep_w, grep_b, grep_a, B, T, H, gate, (hipStream_t)stream));
    return 0;
}

int s3enc_op_posconv(int32_t dtype, const float* x, const float* w_host, const float* bias, int32_t B, int32_t T, int32_t D,
                     int32_t G, int32_t K, float* out, void* stream) {
    if (!x || !w_host || !bias || !out) return fail("s3enc_op_posconv: null argument");
    if (G <= 0 || D % G) return fail("s3enc_op_posconv: D must be a multiple of groups");
    const int Dg = D / G;
    std::vector<float> w(w_host, w_host + (size_t)D * Dg * K), packed;
    DevBuf dw;
    if (dtype == 3) {
        HIP_TRY(upload_posconv_x3(dw, w, D, G, K));
    } else {
        pack_posconv(w, D, G, K, dtype, packed);
        HIP_TRY(upload_cvt(dw, packed, dtype));
    }
    PosConvParams p{};
    p.x = x;
    p.w = dw.p;
    p.bias = bias;
    p.out = out;
    p.B = B;
    p.T = T;
    p.D = D;
    p.G = G;
    p.K = K;
    HIP_TRY(dtype == F32 ? launch_posconv(p, (hipStream_t)stream) : launch_posconv16(dtype, p, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // the packed weights are freed on return
    return 0;
}

int s3enc_weighted_sum(const float* hs, int64_t layer_stride, int32_t L, const float* w, int32_t normalize, int64_t rows, int32_t D,
                       float* out, void* stream) {
    if (!hs || !w || !out) return fail("s3enc_weighted_sum: null argument");
    if (L <= 0 || L > S3_WS_MAX_LAYERS) return fail("s3enc_weighted_sum: 1..32 layers");
    if (D <= 0 || (D & 3) || D > 2048) return fail("s3enc_weighted_sum: D must be a multiple of 4, <= 2048");
    HIP_TRY(launch_weighted_sum(hs, layer_stride, L, w, normalize, rows, D, out, (hipStream_t)stream));
    return 0;
}

int64_t s3enc_weighted_sum_backward_scratch(int64_t rows, int32_t L) {
    return rows > 0 && L > 0 ? (int64_t)weighted_sum_bwd_blocks(rows) * L : 0;
}

int s3enc_weighted_sum_backward(const float* hs, int64_t layer_stride, int32_t L, int32_t normalize, int64_t rows, int32_t D,
                                const float* grad_out, float* grad_w, double* scratch, void* stream) {
    if (!hs || !grad_out || !grad_w || !scratch) return fail("s3enc_weighted_sum_backward: null argument");
    if (L <= 0 || L > S3_WS_MAX_LAYERS) return fail("s3enc_weighted_sum_backward: 1..32 layers");
    if (D <= 0 || (D & 3) || D > 2048) return fail("s3enc_weighted_sum_backward: D must be a multiple of 4, <= 2048");
    if (rows <= 0) return fail("s3enc_weighted_sum_backward: no rows");
    HIP_TRY(launch_weighted_sum_bwd(hs, layer_stride, L, normalize, rows, D, grad_out, scratch, grad_w, (hipStream_t)stream));
    return 0;
}

static FbankParams fbank_params(const s3enc_fbank_config* c) {
    FbankParams f;
    f.sample_rate = c->sample_rate;
    f.num_mel_bins = c->num_mel_bins;
    f.frame_length_ms = c->frame_length_ms;
    f.frame_shift_ms = c->frame_shift_ms;
    f.preemph = c->preemphasis;
    f.delta_order = c->delta_order;
    f.delta_win = c->delta_win_length;
    f.use_cmvn = c->use_cmvn;
    f.cmvn_eps = c->cmvn_eps;
    return f;
}

int s3enc_fbank_num_frames(const s3enc_fbank_config* cfg, int64_t n_samples, int32_t* frames) {
    if (!cfg || !frames) return fail("s3enc_fbank_num_frames: null argument");
    *frames = (int32_t)fbank_num_frames(n_samples, fbank_params(cfg));
    return 0;
}

int s3enc_fbank_forward(const s3enc_fbank_config* cfg, const float* const* wavs, const int64_t* lengths, int32_t B, float* out,
                        int64_t T_max, int32_t device, void* stream) {
    if (!cfg || !wavs || !lengths || !out) return fail("s3enc_fbank_forward: null argument");
    if (B <= 0) return fail("s3enc_fbank_forward: empty batch");
    const FbankParams f = fbank_params(cfg);
    if (f.delta_order < 0 || f.delta_order > 2 || f.delta_win < 3 || !(f.delta_win & 1)) return fail("s3enc_fbank_forward: unsupported delta configuration");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("s3enc_fbank_forward: no HIP device (there is no CPU fallback)");
    HIP_TRY(hipSetDevice(device));
    const int F = f.num_mel_bins * (f.delta_order + 1);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * T_max * F * sizeof(float), st));
    for (int b = 0; b < B; ++b) {
        const long T = fbank_num_frames(lengths[b], f);
        if (T <= 0) return fail("s3enc_fbank_forward: an utterance is shorter than one analysis window");
        if (T > T_max) return fail("s3enc_fbank_forward: T_max is smaller than an utterance's frame count");
        HIP_TRY(launch_fbank(f, wavs[b], lengths[b], out + (size_t)b * T_max * F, F, st));
    }
    return 0;
}

}  // extern "C"
