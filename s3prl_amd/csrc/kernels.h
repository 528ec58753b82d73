// kernels.h — host-side launch interface of the HIP kernels (internal to libs3enc).
#pragma once
#include <cstring>
#include <vector>

#include "common.h"

namespace s3 {
int device_cus();  // gemmt.hip: CUs of the device the process first launched on (256 on an MI355X)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device) instead of once per launch:
// the call costs several microseconds of host time, and a forward is ~290-560 launches.
template <auto Kern>
inline hipError_t ensure_dynamic_lds(int bytes) {
    static int granted[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && granted[dev] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) granted[dev] = bytes;
    return e;
}

// ---- kernel-variant selection (A/B measurement knobs; results never depend on them) -------------------------
// One struct instead of scattered globals: `g_tuning` holds the process defaults (s3enc_set_tuning); a handle may carry its
// own copy (s3enc_set_handle_tuning), which the engine makes current for the calling thread while that handle's forward
// enqueues its kernels (TuningScope) — so two handles, or two threads, never see each other's settings.
struct Tuning {
    int gemm_variant = 3;  // gemm.hip: bit 0 = 64-byte K stages, bit 1 = LDS-DMA staging, bit 2 = no XCD-aware tile order
    int gemm_lds_pad = 0;  // gemm.hip occupancy probe: extra (unused) dynamic LDS per workgroup
    int gemm32_big = 1;    // gemmt.hip, fp32: 0 off, 1 = tile height by shape, 2..5 = force 256 / 192 / 128 / 64 rows
    int gemm_x3_tile = 1;  // gemmt.hip, S3ENC_F32X3: 0 off, 1 = only the shapes with few tiles, 2..5 = force a height
    int gemm16_big = 3;    // gemm16.hip: 0 off, 3 = choose by shape, 1 / 2 / 4 / 5 / 6 / 7 = force one configuration
    int gemm16_pp = 0;     // gemm16.hip, persistent loop: 0 = every wave issues its LDS-DMA pieces behind fragment steps 0 / 1;
                           // non-zero = waves 4-7 (the SIMD partners of waves 0-3) behind steps 1 / 2 instead (PP 3; measured
                           // within +-1-3 % of 0: profiles/r05_gemm16_loop_probe.md — opt-in)
    int gemm16_mx = 14;    // S3ENC_F16X2, bit mask: which GEMMs take their second weight term as an MX-fp4 image on the scaled-MFMA pipe
                           // where the shape allows (gemm16.hip MXW) — 1 conv1, 2 q|k|v, 4 fc1, 8 fc2; 0 = two fp16 terms everywhere;
                           // 16 = (read at s3enc_create) also the weights whose 256-row tiling is cheaper at the reference batch (measurements).  Default 14: conv1 stays on two fp16 terms —
                           // its A operand (conv0's GroupNorm'd output with loud channels) is the one block-scaled fp4 images suit least
                           // (wav2vec2_base_pl 6.0e-4 with 14, 7.9e-4 with 15: profiles/r05_mx_second_term.md)
    int reserve_cus = 0;   // gemm16.hip, persistent loop: CUs left out of the one-workgroup-per-CU grid.  Measured (profiles/r05_cu_contention.md):
                           // 8-32 foreign workgroups holding CU slots cost the 16-bit step +2 % whatever their number, while a grid of
                           // 240 costs +20 % (the tile rounds are tuned to 256 CUs: q|k|v's 756 tiles become four rounds) — keep 0
    int gemm16_rows = 1;   // gemm16.hip: 1 = GELU epilogues with a 16-bit output take the row-per-lane (no LDS) form, 0 = never
    int attn_lds_pad = 0;  // 16-bit attention occupancy probe
    int attn_persist = 0;  // attention.hip: 1 = persistent workgroups that fetch the next (batch, head, query block) item's Q and first
                           // K / V tile under the current item's last tile (round 6; bit-identical).  MEASURED AND LEFT OFF: the 16-bit
                           // form is 20-30 % slower than the one-shot grid (lab 58 -> 78 us, forward 48.7 -> 53.3 us per launch), the
                           // fp32 form -7 % in the lab and +0.7 % in the forward (profiles/r06_attn_lab.md, r06_attention_persist.md);
                           // against the one-shot kernel of the second session: 43.6 vs 38.5 us HuBERT-base, 54.8 vs 50.4 HuBERT-large,
                           // 110 vs 107 / 130 vs 121 WavLM-large (profiles/r06b_attn_lab_variants.md)
    int conv0_nt = 1;      // frontend.hip, fp32 output: 1 = non-temporal row stores (the 2 GB activation streams past the caches:
                           // 0.578 -> 0.436 ms on HuBERT-base 32 x 10 s, round 4), 0 = plain stores
    int conv0_fast = 1;    // frontend.hip, 16-bit outputs (S3ENC_BF16 / F16 / F16X2) and the split-precision modes' fp32 output: 1 = packed fp32 taps + the packed one-transcendental GELU,
                           // 0 = scalar taps + libm erff (the FAST = false instantiation: ~2x the kernel's time, results a few 16-bit ulps
                           // apart).  Round 6, fourth session: the rows that differ when >= 4 handles' forwards overlap ORIGINATE in
                           // conv0_kernel<16-bit, FAST> — single frames, lanes 48-63, the low halves of the packed outputs — and with 0
                           // (and forward_chain = 0) 32 of 32 overlapping runs kept their bits; neither half of FAST alone, nor wait states,
                           // nor the transcendental unit explains it (profiles/r06d_concurrent_forwards_exclusions.md).  A diagnostic
                           // switch, not a fix: the forward chain stays the default protection
    int ws_inplace = 1;    // engine.hip, post-LN layers: 1 = LayerNorm 1 and fc2 work in place on ONE fp32 buffer (49 MB less
                           // working set per layer: fc2 -3.6 %, LayerNorm -4 % in the bf16 forward, round 4), 0 = two buffers
    int gelu32 = 1;        // S3ENC_F32: 1 = the one-transcendental GELU of every mode (common.h gelu_fast; fp32 rounding level), 0 = libm erff
    int x3_pack_cache = 0; // s3enc_op_gemm(S3ENC_F32X3): keep the packed image of the last weight (micro-benchmarks)
    int fp16x2_conv1_f32 = 0;  // engine.hip, S3ENC_F16X2 (read at s3enc_create): 1 = conv0 writes fp32 and conv1 runs on the three-term GEMM too
    int ln_rows = 1;       // norm.hip: rows per wave of the row LayerNorm.  2 = both rows' loads in flight before the first reduction (launches of
                           // >= 8192 rows; bit-identical by test).  MEASURED AND LEFT OFF (round 6, second session, same lease, HuBERT-base
                           // 32 x 10 s): LN1 + LN2 0.536 -> 0.551 ms per bf16 forward, 0.513 -> 0.529 fp32 — the kernel is at the memory
                           // side's rate (5.3-6.1 TB/s), not short of loads in flight
    int ln_preload = 1;    // norm.hip: 1 = a wave fetches its gamma / beta chunks together with its row instead of behind the two reductions
    int ln1_fold = 1;      // engine.hip, post-LN layers in the 16-bit modes: 1 = LayerNorm 1 writes its 16-bit output and the rows' statistics only, fc2's
                           // epilogue rebuilds the fp32 rows it adds (bit-identical to 0 = LayerNorm 1 writes them)
    int gn_lag_one_block = 1;  // frontend.hip: 1 = GroupNorm lag sums from ONE workgroup per (4096-frame chunk, utterance) over an LDS-staged
                           // window (bit-identical to the k0-workgroups form, 95.7 -> see profiles/r06b_gn_stats.md), 0 = the earlier kernel
    int forward_chain = 1; // engine.hip: 1 = a forward of a 16-bit / split-precision handle waits (hipStreamWaitEvent, no host wait) for the previous
                           // such forward of ANY handle on the same device, whatever stream that one ran on.  With one handle on one stream the
                           // wait is already implied by stream order.  Why: forwards of SEVERAL handles running at once on >= 4 streams are NOT
                           // bit-stable in the modes whose GEMMs stage through global_load_lds (bf16 / fp16 at four streams, fp16x2 / fp32x3 at
                           // eight; never exact fp32, which is left free to overlap): rare rows come out a few 16-bit ulps off
                           // (tools/two_stream_probe.py, profiles/r06c_concurrent_forwards.md; no kernel or pair of kernels reproduces it in
                           // isolation, two streams never do).  Price: sub-batches that under-fill the chip no longer overlap (four
                           // 8-utterance forwards: 8.5 -> 12.1 ms; the same 32 utterances as ONE batch: 6.8 ms).  0 = forwards may overlap
    int comm_self_p2p = 0; // comm.hip, S3ENC_EXCHANGE_DIRECT: 1 = a rank's OWN block also travels as an ncclSend-to-self / ncclRecv-from-self
                           // pair inside the state's group instead of a device copy — on a one-GPU box this is the only way the
                           // all-pairs code (symbols, counts, datatype, group bracketing, stream order behind the layer events)
                           // executes at all: RCCL refuses two ranks on one device.  A test hook; same bytes in the same places
};
extern Tuning g_tuning;
extern thread_local const Tuning* t_tuning;  // the current handle's override, or null
inline const Tuning& tuning() { return t_tuning ? *t_tuning : g_tuning; }
struct TuningScope {
    const Tuning* prev;
    explicit TuningScope(const Tuning* t) : prev(t_tuning) { if (t) t_tuning = t; }
    ~TuningScope() { t_tuning = prev; }
};
int tuning_set(Tuning& t, const char* key, int value, const char** err);  // 0 ok; ops.hip

// ---- forward status word (s3enc_forward_status, ABI 6) ---------------------------------------------------------
// A device int32 owned by the handle; the row LayerNorm kernels OR bit 0 into it when a row's mean or variance is not
// finite (every hidden state passes through one of them: a post-LN state IS a LayerNorm output, a pre-LN residual stream
// is the next LayerNorm's input) — the symptom of a 16-bit overflow upstream of the row, or of non-finite PCM.  Current for
// the calling thread while a handle's forward enqueues its kernels, like the tuning scope above; null outside.
extern thread_local int* t_status;
struct StatusScope {
    int* prev;
    explicit StatusScope(int* p) : prev(t_status) { t_status = p; }
    ~StatusScope() { t_status = prev; }
};

// ---- gemm.hip -----------------------------------------------------------------------------------------
struct GemmParams {
    const void* A;  // (batches, M, K) rows at A + b*a_bs + m*lda (elements of the compute dtype)
    long lda, a_bs;
    const void* W;  // (N, K) row-major, compute dtype
    const void* W_x3 = nullptr;  // fp32 mode only: the same matrix pair-packed as bf16 hi / lo (gemm_x3.hip), or null
    const float* bias;
    int M, N, K, batches;
    int act;                // 0 none, 1 erf-GELU
    const float* residual;  // fp32, indexed like out32; added after the activation
    const int* row_limit;   // per batch: rows >= row_limit[b] are written as 0
    // round 6 (second session): the residual is NOT read as stored — `residual` holds the INPUT rows t of a LayerNorm and the epilogue
    // adds LayerNorm(t) = ln_affine(t, mu, rs, gamma, beta) rebuilt from the row's statistics (res_ln_stats[m] = (mu, rs), written by
    // launch_layernorm(..., stats_out)): the post-LN layers' LayerNorm 1 no longer writes its fp32 output (49 MB per layer at the
    // reference batch), fc2 re-reads the row it normalised.  gemm16_big residual epilogue only (gemm16_res_ln_ok)
    const float2* res_ln_stats = nullptr;
    const float* res_ln_g = nullptr;
    const float* res_ln_b = nullptr;
    float* out32;
    void* out16;
    long ldo, o_bs;
    int variant = -1;  // tuning knob: -1 = the current tuning().gemm_variant
    // 16-bit modes: W row stride in elements (0 = K, or 2K with wsplit) and the two-term weight split of S3ENC_F16X2:
    // every W row is [hi(K) | lo(K)] with w = hi + lo (two fp16), and the contraction runs over 2K with A read twice:
    //   sum_k a[k] * hi[k] + sum_k a[k] * lo[k]   — the weights' rounding error disappears at twice the matrix cost
    long ldw = 0;
    int wsplit = 0;
    // S3ENC_F16X2, round 5: the lo term as an MX-fp4 image — W4: (N, K/32, 16 bytes) e2m1 nibbles, element e of a block in nibble e;
    // W4s: (N, K/32) E8M0 scale bytes; mxw: run the contraction over the hi half of the [hi | lo] rows + this image (gemm16.hip, MXW)
    // instead of over both halves.  Set by the engine when the image exists; launch_gemm falls back to `wsplit` when the shape is
    // not the MX kernel's (gemm16_mx_eligible).
    const void* W4 = nullptr;
    const void* W4s = nullptr;
    int mxw = 0;
};
bool gemm16_mx_eligible(int dtype, const GemmParams& p);
bool gemm16_mx_weight_rule(long N, long K);  // which weights get an MX image at s3enc_create (per weight, batch-independent)
// gemm_x3.hip: fp32-class GEMM from three bf16 MFMAs per product (opt-in compute mode S3ENC_F32X3)
bool gemm_x3_eligible(const GemmParams& p);
hipError_t launch_gemm_x3(const GemmParams& p, hipStream_t stream);
void pack_x3(const float* w, long N, long K, std::vector<uint16_t>& out);  // host: fp32 (N, K) -> pair-packed bf16 hi / lo
// gemm16.hip: large-tile LDS-DMA kernel for the 16-bit modes (tuning().gemm16_big)
bool gemm16_big_eligible(int dtype, const GemmParams& p);
bool gemm16_res_ln_ok(int dtype, const GemmParams& p);  // may this call carry GemmParams::res_ln_* ?
hipError_t launch_gemm16_big(int dtype, const GemmParams& p, hipStream_t stream);
// gemmt.hip: (256 | 192 | 128 | 64) x 128 tiles, several independent workgroups per CU; fp32 results bit-identical to
// gemm.hip's kernel.  tuning().gemm32_big (fp32) / .gemm_x3_tile (S3ENC_F32X3 = dtype code 3 below: fp32 operands, the
// pair-packed p.W_x3)
bool gemm_tile_eligible(int dtype, const GemmParams& p);
hipError_t launch_gemm_tile(int dtype, const GemmParams& p, hipStream_t stream);
hipError_t launch_gemm(int dtype, const GemmParams& p, hipStream_t stream);

// ---- frontend.hip ---------------------------------------------------------------------------------------
// Waveform table: wav b is `ptrs[b]` with `lens[b]` valid samples; reads beyond are zeros (the padding).
struct WavTable {
    const float* const* ptrs;  // device array [B]
    const long* lens;          // device array [B]
    int B;
    long n_max;
};
constexpr int STAT_K0_MAX = 16;  // conv0 kernel width limit for the closed-form GroupNorm statistics
// per-utterance mean / rstd of the raw waveform (task_cfg.normalize); norm[b] = {mean, rstd}; identity if !normalize
hipError_t launch_wav_norm_stats(const WavTable& w, int normalize, double* partial, float2* norm, hipStream_t s,
                                 float eps = 0.f /* 0 = LN_EPS; Hugging Face feature extractors: 1e-7 */);
// GroupNorm(C,C) statistics of conv0's output from the k0 + k0*(k0+1)/2 lag sums of the waveform, then the
// fused per-(b,c) affine:  gn[b][c] = {scale, shift} with  y = conv0_raw * scale + shift
hipError_t launch_gn_stats(const WavTable& w, const float2* norm, const float* w0 /*[C][k0]*/, const float* gamma,
                           const float* beta, int C, int k0, int s0, long L0, double* partial, double* sums,
                           float2* gn, hipStream_t s);
size_t stats_partial_elems(int B, long n_max);  // doubles needed for `partial`
struct Conv0Params {
    WavTable wav;
    const float2* norm;   // [B] {mean, rstd}
    const float* w0;      // [C][k0]
    const float* bias;    // [C] or null
    const float2* gn;     // [B][C] {scale, shift} (GroupNorm mode) or null
    const float* ln_g;    // [C] (layer_norm mode) or null
    const float* ln_b;
    int C, k0, s0;
    long L0;
    void* out;            // (B, L0, C) compute dtype
    int fast = 0;         // fp32 output with the packed one-transcendental GELU (the split-precision modes)
    int nt = 0;           // fp32 output: non-temporal stores
};
hipError_t launch_conv0(int dtype, const Conv0Params& p, hipStream_t s);

// ---- norm.hip -------------------------------------------------------------------------------------------
// Featurizer term fused into a row kernel: acc[row] (+)= w * state_row  (norm: w * layer_norm(state_row), no affine)
struct LnAcc {
    float* acc = nullptr;  // (rows, C) fp32
    float w = 0.f;
    int mode = 0;  // 0 none; 1: the kernel's INPUT row is the state; 2: its OUTPUT row is
    int norm = 0;
    int init = 0;  // first term: write instead of add
};
// WavLM gate (wavlm/modules.py:535-549) of the attention module that will read this LayerNorm's OUTPUT, fused into the
// LayerNorm pass (the row is in registers): gate[b][h][t] for the (b, t) row; needs C == H * 64.  Replaces a separate
// pass over the (B*T, D) fp32 tensor (and, in the 16-bit modes, the fp32 copy written only for it).
struct LnGate {
    const float* gw = nullptr;  // grep_linear.weight (8, 64)
    const float* gb = nullptr;  // grep_linear.bias (8)
    const float* ga = nullptr;  // grep_a (H)
    float* gate = nullptr;      // (B, H, T) or null: no gate
    int T = 0, H = 0;
};
// act: 0 none, 1 erf-GELU of the mode (16-bit: common.h gelu_fast; fp32: the same unless tuning gelu32 = 0 -> libm erff),
//      2 gelu_fast regardless of dtype and tuning
hipError_t launch_layernorm(int dtype, const float* x, const float* gamma, const float* beta, long rows, int C, int act,
                            float* out32, void* out16, hipStream_t s, const LnAcc& fa = LnAcc(), const LnGate& gt = LnGate(),
                            float2* stats_out = nullptr);  // stats_out[row] = (mean, 1 / sqrt(var + eps)) of the row
// a state produced by a non-LayerNorm kernel: its 16-bit copy (out16, dtype BF16 / F16) and / or its Featurizer term
hipError_t launch_emit_state(int dtype, const float* x, long rows, int C, void* out16, const LnAcc& fa, hipStream_t s);
hipError_t launch_add(const float* a, const float* b, float* out, long n, hipStream_t s);  // out = a + b, n % 4 == 0

// ---- attention.hip ----------------------------------------------------------------------------------------
struct AttnParams {
    const void* qkv;  // (B*T, 3D): q | k | v, q already scaled by head_dim^-0.5
    void* out;        // (B*T, D)
    const int* valid; // [B] keys >= valid[b] are masked
    int B, T, H;
    const float* bias_table;  // WavLM: [H][2R+1], entry (h, clamp(key - query, -R, R) + R), or null
    int table_R = 0;
    const float* gate;        // WavLM: [B][H][T] or null (then gate = 1)
    int out_f32 = 0;          // 16-bit kernels only: write the (B*T, D) result as fp32 (S3ENC_F16X2: out_proj then reads it
                              // through the three-term GEMM — its operand's rounding is the largest non-conv term of the mode's error)
    int probe = 0;            // timing probes (tools/micro/attn_lab.hip builds the kernels with S3_ATTN_PROBE; ignored otherwise)
};
hipError_t launch_attention(int dtype, const AttnParams& p, hipStream_t s);
// WavLM gate (wavlm/modules.py:535-549) from the layer input x (fp32 rows of D): gate[b][h][t]
hipError_t launch_wavlm_gate(const float* x, const float* grep_w /*[8][64]*/, const float* grep_b /*[8]*/,
                             const float* grep_a /*[H]*/, int B, int T, int H, float* gate, hipStream_t s);

// ---- posconv.hip --------------------------------------------------------------------------------------------
struct PosConvParams {
    const float* x;     // (B, T, D) fp32, padded frames already zero
    const void* w;      // fp32 mode: packed [G][K][Dg/16][Dg(n)][16]; 16-bit modes: see launch_posconv16
    const float* bias;  // [D]
    float* out;         // (B, T, D) fp32 = x + gelu(conv(x) + bias)
    int B, T, D, G, K;
    int pad = -1;       // left zero padding in frames; -1 = K / 2 (a K padded with zero taps keeps the real kernel's K / 2)
    int plain = 0;      // 1: out = conv(x) + bias only (data2vec's conv -> LayerNorm -> GELU stack, wav2vec2_model.py:2999-3017)
};
hipError_t launch_posconv(const PosConvParams& p, hipStream_t s);
// fp32 pack: slot permutation inside a 16-float weight row.  ds_read_b128 serves a wave in four groups of 16 lanes ({0-3, 12-15,
// 20-27}, ... — MI355X_MICROARCH.md §LDS); lane (co & 15, s) reads slot s of row co.  Rows are 64 bytes, so rows co and co + 4
// start on the same bank: with the slots of rows 8..15 XORed by 2, the 16 lanes of every group touch 16 different slot positions
// modulo the 256-byte bank row (round 3 measured SQ_LDS_BANK_CONFLICT / active = 0.315 on the linear layout).
__host__ __device__ inline int pc_w_swizzle(int co) { return ((co & 15) >> 3) << 1; }
// 16-bit operand modes: p.w = 16-bit pack [G][Dg][K*Dg] with k = tap*Dg + ci; x / out / bias fp32
hipError_t launch_posconv16(int dtype, const PosConvParams& p, hipStream_t s);

// ---- adapter.hip (multires-HuBERT: the row passes of the conv adapters, multires_hubert/hubert_model.py:970-1266) ------
// Operand geometry of the adapter convolutions: per utterance `total` rows of D — `lead` zero rows, the data rows, zero
// rows to the end — so that a Conv1d / ConvTranspose1d window that hangs over either end reads zeros and the convolution
// is a plain GEMM over contiguous k*D-element rows (multires.hip; weights packed by load_conv in engine.hip).
struct PadCopyParams {
    const float* a;      // (B, >= rows, D) fp32, utterance b at a + b*a_bs
    long a_bs;
    const float* b;      // optional second term added row by row (x + residual, align_size_sum), or null
    long b_bs;
    int B, rows, D;      // data rows copied per utterance
    int lead, total;     // output rows: [0, lead) zero, [lead, lead + rows) data, [lead + rows, total) zero
    const int* zero_from;  // optional [B]: data rows >= zero_from[b] are written as zero
    float* out32;        // (B, total, D) fp32 and / or ...
    void* out16;         // ... the 16-bit compute dtype; either may be null
};
hipError_t launch_pad_copy(int dtype, const PadCopyParams& p, hipStream_t s);
constexpr int GS_BLOCKS = 64;  // partial sums per utterance of the one-group GroupNorm statistics
// partial[b][blk] = {sum, sum of squares} over slice blk of the `count` contiguous fp32 values at x + b*bs
hipError_t launch_group1_stats(const float* x, long bs, long count, int B, double* partial /*[B][GS_BLOCKS][2]*/, hipStream_t s);
struct AdapterApplyParams {
    const float* conv;      // (B, L, D) fp32 conv output, utterance b at conv + b*conv_bs
    long conv_bs;
    const double* partial;  // launch_group1_stats of that output
    double count;           // elements per utterance the statistics run over (L * D: ALL conv frames, kept or not)
    const float* gamma;     // GroupNorm(1, D) affine
    const float* beta;
    const float* r1;        // skip term: row (t * r1_mul) / r1_div of utterance b at r1 + b*r1_bs
    long r1_bs;
    int r1_mul, r1_div;
    const float* r2;        // highway term (ConvAdapter) or null
    long r2_bs;
    int r2_mul, r2_div;
    float scale;            // sqrt(residual_scale)
    int B, rows, D;         // kept output frames per utterance
    int lead, total;        // output geometry as in PadCopyParams (lead = 0, total = rows: a plain (B, rows, D) tensor)
    const int* zero_from;   // optional [B]: frames >= zero_from[b] are written as zero (the next encoder's index_put)
    float* out32;
    void* out16;
    int fast_gelu;          // fp32 instantiation only: gelu_fast (S3ENC_F32X3); the 16-bit ones always use it
};
hipError_t launch_adapter_apply(int dtype, const AdapterApplyParams& p, hipStream_t s);
// state of a block -> (B, rows_out, D) slot: out[b][t] = x[b][t / factor]  (repeat_interleave + cut, expert.py:26-27,93-101)
hipError_t launch_emit_upsampled(int dtype, const float* x, long x_bs, int factor, int B, int rows_out, int D, float* out32,
                                 void* out16, hipStream_t s);

// the same state as its Featurizer term only: fa.acc[b][t] (+)= fa.w * (fa.norm ? layer_norm(x[b][t / factor]) : x[b][t / factor])
hipError_t launch_emit_upsampled_acc(const float* x, long x_bs, int factor, int B, int rows_out, int D, const LnAcc& fa,
                                     hipStream_t s);

// ---- featurizer.hip (weighted sum over layers, the consumer of hidden_states) ---------------------
#define S3_WS_MAX_LAYERS 32
// out[row] = sum_l w[l] * (normalize ? layer_norm(hs[l][row]) : hs[l][row]);  w: HOST array (softmax already applied,
// 0 for unselected layers); hs[l] at hs + l*layer_stride, rows x D fp32
hipError_t launch_weighted_sum(const float* hs, long layer_stride, int L, const float* w_host, int normalize, long rows, int D,
                               float* out, hipStream_t st);
// grad_w[l] = sum <grad_out, hn_l>;  partial: device scratch of weighted_sum_bwd_blocks(rows) * L doubles
int weighted_sum_bwd_blocks(long rows);
hipError_t launch_weighted_sum_bwd(const float* hs, long layer_stride, int L, int normalize, long rows, int D,
                                   const float* grad_out, double* partial, float* grad_w, hipStream_t st);

// ---- fbank.hip (the `fbank` baseline upstream, BASELINE configs[0]) -----------------------------------------------
struct FbankParams {
    int sample_rate = 16000;
    int num_mel_bins = 80;
    float frame_length_ms = 25.f, frame_shift_ms = 10.f;
    float preemph = 0.97f;
    int delta_order = 2, delta_win = 5;
    int use_cmvn = 1;
    float cmvn_eps = 1e-10f;
};
long fbank_num_frames(long n_samples, const FbankParams& c);
// one utterance: wav (device) -> out (device, frames x ldo), columns [0, num_mel_bins * (delta_order + 1))
hipError_t launch_fbank(const FbankParams& c, const float* wav, long n, float* out, int ldo, hipStream_t st);

}  // namespace s3
