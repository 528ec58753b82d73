/*
 * s3enc.h — C ABI of libs3enc.so: the MI355X (gfx950) speech-SSL upstream encoder.
 *
 * One hot path of s3prl, behind plain C:  raw 16 kHz waveforms  ->  the list of per-layer
 * hidden_states of a wav2vec 2.0 / HuBERT / WavLM encoder.  The reference is pure Python on top of
 * PyTorch ATen, so there is no reference FFI to bind; each entry point below names the reference
 * Python interface it stands in for (paths relative to the s3prl tree).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; s3enc_last_error() (thread-local)
 *     describes the last failure.  Nothing throws across this boundary.
 *   - pointers documented "device" are HIP device pointers on the handle's GPU; "host" are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Work is enqueued
 *     asynchronously on it and ordered with the caller's own work on that stream.
 *   - a handle is not re-entrant and serves ONE stream at a time: its workspaces are grown and released in the order of the
 *     stream of the call that grows them, so before moving a handle to another stream the caller orders that stream after
 *     the handle's last forward (an event, or a synchronise).  Distinct handles are independent (one per GPU / rank / stream).
 *   - there is no CPU fallback: without a gfx950 device s3enc_create fails.
 */
#ifndef S3ENC_H
#define S3ENC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3ENC_VERSION 7
#define S3ENC_MAX_CONV 16
#define S3ENC_MAX_RES 4 /* resolutions of a multires-HuBERT U-net: up to 3 rate pairs, 7 encoder blocks */

typedef struct s3enc_encoder* s3enc_handle;

enum { S3ENC_HUBERT = 0, S3ENC_WAV2VEC2 = 1, S3ENC_WAVLM = 2,
       /* DistilHuBERT (upstream/distiller/model.py:83-268): HuBERT-style encoder without the LayerNorm before
        * post_extract_proj, wav2vec2's conv-length frame mask, prediction heads on the last layer */
       S3ENC_DISTILLER = 3,
       /* multi-resolution HuBERT (upstream/multires_hubert/hubert_model.py:337-852): a U-net of TransformerEncoders
        * (encoders[i] -> conv adapter -> ... middle_encoder ... -> conv adapter -> decoders[i] + skip) over the frame rates
        * of mr_ratios; HuBERT's frame mask; the states are every block's layer inputs and output, repeated to the
        * finest rate and cut to their common length (multires_hubert/expert.py:26-27,49-101) */
       S3ENC_MULTIRES = 4 };
/* arithmetic type of the GEMM / attention operands; accumulation, norms, softmax, GELU and the residual
 * stream are always fp32 (the reference's Fp32GroupNorm / Fp32LayerNorm / fp32 softmax guards,
 * wav2vec2_model.py:1826-1853,1899-1900). */
enum { S3ENC_F32 = 0, S3ENC_BF16 = 1, S3ENC_F16 = 2,
       /* fp32 data flow (activations, norms, softmax, attention, positional conv exactly as S3ENC_F32) with the GEMMs
        * computed as three bf16 MFMAs per product on split operands (x = hi + lo): ~1e-5 relative error per GEMM at
        * 3/16 of the exact-fp32 matrix cost.  Opt-in; S3ENC_F32 stays the exact default. */
       S3ENC_F32X3 = 3,
       /* fp16 data flow (exactly S3ENC_F16: fp16 activations / attention, fp32 accumulate, norms, softmax, residual) with
        * every GEMM weight kept as TWO terms (w = hi + lo) and the contraction run over both: the weights' rounding error
        * disappears, leaving the activations'.  What a binder gets BY DEFAULT:
        *   - hi = fp16(w); lo = a second fp16 term — or, for q|k|v, fc1 and fc2 where the weight's shape pays for it, an MX-fp4
        *     image of lo (per row and 32 k an E8M0 scale + 32 e2m1 values, packed at s3enc_create) multiplied on the scaled-MFMA
        *     pipe in the same K step: 4.8e-5 of weight error per GEMM instead of 5e-7.  Tuning key "gemm16_mx" (read at
        *     s3enc_create, default 14; 0 = two fp16 terms everywhere) selects the GEMMs;
        *   - where a rounded fp16 OPERAND carries the error budget the GEMM reads fp32 activations through the three-term
        *     kernel instead: conv2.. of EVERY extractor (GroupNorm and layer-norm: conv1 writes fp32), post_extract_proj (the
        *     fp32 LayerNorm(C) output) and out_proj (the attention kernel writes fp32).  "fp16x2_conv1_f32" = 1 adds conv1.
        * Measured on outputs of the reference itself (profiles/r06_parity_seeds.md, 39 full-dimension fixtures on
        * released-checkpoint-like weight statistics, weight seeds 0-6 x five models): max relative error of a hidden state
        * 2.5e-4 ... 8.6e-4, median 5.3e-4 (8.0e-4 worst with conv1 on fp32 rows) — inside the path's 1e-3 on every seed, where
        * S3ENC_F16 is 0.9e-3 ... 2.8e-3 and S3ENC_BF16 0.7e-2 ... 1.5e-2 — at 3.1x the S3ENC_F32 throughput on HuBERT-base.
        * Its stated limit: an FFN whose hidden activation leaves the fp16 range (GELU(fc1) > 65504: fc1 scaled x1000+ on the
        * pretrained-like models, profiles/r05_fp16_cliff.md) returns non-finite states — reported by s3enc_forward_status; with
        * fc1 x10 ... x1000 WavLM-large settles at 1.2-1.4e-3.  Opt-in. */
       S3ENC_F16X2 = 4 };

/* Hyper-parameters that select kernel variants.
 * Replaces: HubertConfig / HubertPretrainingConfig (upstream/hubert/hubert_model.py:33-278),
 *           Wav2Vec2Config / AudioPretrainingConfig (upstream/wav2vec2/wav2vec2_model.py:2103-2350,3325-3345),
 *           WavLMConfig (upstream/wavlm/WavLM.py:162-245). */
typedef struct s3enc_config {
    int32_t family;                        /* S3ENC_HUBERT / WAV2VEC2 / WAVLM: selects the frame-mask rule */
    int32_t n_conv;                        /* number of conv feature layers (7) */
    int32_t conv_dim;                      /* channels of every conv layer (512) */
    int32_t conv_kernel[S3ENC_MAX_CONV];   /* (10,3,3,3,3,2,2) */
    int32_t conv_stride[S3ENC_MAX_CONV];   /* (5,2,2,2,2,2,2) */
    int32_t extractor_layer_norm;          /* 0: "default" (GroupNorm after conv0); 1: "layer_norm" */
    int32_t conv_bias;
    int32_t encoder_layers;
    int32_t embed_dim;
    int32_t ffn_dim;
    int32_t heads;
    int32_t layer_norm_first;              /* 0: post-LN (base); 1: pre-LN (large) */
    int32_t conv_pos;                      /* positional conv kernel (128) */
    int32_t conv_pos_groups;               /* (16) */
    int32_t normalize;                     /* per-utterance waveform layer-norm (task_cfg.normalize) */
    int32_t rel_pos;                       /* WavLM: relative_position_embedding */
    int32_t num_buckets;
    int32_t max_distance;
    int32_t gru_rel_pos;
    int32_t compute_dtype;                 /* S3ENC_F32 / BF16 / F16 / F32X3 / F16X2 */
    int32_t no_feature_layer_norm;         /* 1: post_extract_proj reads the conv output directly (distiller/model.py:170-176) */
    int32_t pos_conv_depth;                /* data2vec: > 1 = that many {Conv1d(D, D, max(3, conv_pos / depth), groups) -> LayerNorm(no
                                            * affine) -> GELU} blocks instead of the single weight-normed conv (wav2vec2_model.py:2995-3023);
                                            * 0 / 1 = the standard positional conv */
    float wav_norm_eps;                    /* eps of the waveform normalisation; 0 = 1e-5 (F.layer_norm, hubert/expert.py:57-58);
                                            * Hugging Face's Wav2Vec2FeatureExtractor uses 1e-7 (upstream/hf_hubert/expert.py:30-37) */
    int32_t pred_heads;                    /* DistilHuBERT: N prediction heads Linear(D, N*D) -> GELU -> SplitLinear(D, N, D)
                                            * (distiller/model.py:155-161, module.py:55-90); 0 otherwise */
    /* S3ENC_MULTIRES only (MultiresHubertConfig, multires_hubert/hubert_model.py:97-148); encoder_layers = sum(mr_layers) */
    int32_t mr_pairs;                               /* rate pairs = resolutions - 1 (label_rate_ratios has 2 * mr_pairs entries) */
    int32_t mr_ratios[2 * (S3ENC_MAX_RES - 1)];     /* label_rate_ratios: up_0, down_0, up_1, down_1, ... */
    int32_t mr_layers[2 * S3ENC_MAX_RES - 1];       /* layers per block in execution order: encoders..., middle, decoders... */
    int32_t mr_kernel;                              /* conv_adapator_kernal (7): odd, every rate divides mr_kernel - 1 */
    int32_t mr_plain;                               /* use_plain_updownsample: ConvDownsampler / ConvUpsampler, else ConvAdapter */
} s3enc_config;

/* A named fp32 host tensor of the checkpoint, named exactly like the reference state_dict entry
 * ("encoder.layers.3.fc1.weight", ...).  Replaces model.load_state_dict(...)
 * (upstream/hubert/convert.py:37-56, wav2vec2/convert.py:26-39, wavlm/expert.py:37-40). */
typedef struct s3enc_tensor {
    const char* name;
    const float* data;  /* host, contiguous, row-major */
    int32_t ndim;
    int64_t shape[4];
} s3enc_tensor;

int s3enc_version(void);
const char* s3enc_last_error(void);

/* Build an encoder on GPU `device`: packs the weights (folds weight-norm, concatenates q/k/v, folds the
 * 1/sqrt(head_dim) query scale, re-lays conv weights tap-major, converts to the compute dtype) and uploads them.
 * Replaces UpstreamExpert.__init__ (hubert/expert.py:27-51, wav2vec2/expert.py:21-56, wavlm/expert.py:34-54). */
int s3enc_create(const s3enc_config* cfg, const s3enc_tensor* tensors, int32_t n_tensors, int32_t device,
                 s3enc_handle* out);
int s3enc_destroy(s3enc_handle h);

/* T = frames produced for an n-sample input: floor((L-k)/s)+1 through the conv stack
 * (wav2vec2_model.py:2610-2624).  get_downsample_rates() is the product of the strides (320). */
int s3enc_num_frames(s3enc_handle h, int64_t n_samples, int32_t* T);
/* T of the states the forward writes for a batch padded to n_samples: s3enc_num_frames, except for S3ENC_MULTIRES
 * where every state is cut to the common length of the upsampled blocks (multires_hubert/expert.py:93-101). */
int s3enc_num_output_frames(s3enc_handle h, int64_t n_samples, int32_t* T);
int s3enc_downsample_rate(s3enc_handle h, int32_t* rate);
/* Un-masked frames of an utterance of `length` samples in a batch padded to `n_max`:
 * HuBERT/WavLM forward_padding_mask (hubert_model.py:454-464, WavLM.py:339-349), wav2vec2 conv-length rule
 * (wav2vec2_model.py:2652-2669). */
int s3enc_valid_frames(s3enc_handle h, int64_t length, int64_t n_max, int32_t* valid);

/* The forward.  Replaces UpstreamExpert.forward + the hook capture of UpstreamBase.__call__
 * (hubert/expert.py:56-72, upstream/interfaces.py:100-131).
 *   wavs      host array of B device pointers, wavs[b] -> lengths[b] fp32 samples (borrowed, read-only)
 *   lengths   host array of B sample counts
 *   n_max     pad-to length; 0 = max(lengths).  A data-parallel shard passes the GLOBAL batch maximum.
 *   out       device fp32; hidden_states[l] is the contiguous (B, T, D) block at out + l*layer_stride,
 *             l = 0..encoder_layers (layer inputs, then the encoder output)
 *   layer_stride  in elements, >= B*T*D
 */
int s3enc_forward(s3enc_handle h, const float* const* wavs, const int64_t* lengths, int32_t B, int64_t n_max,
                  float* out, int64_t layer_stride, void* stream);

/* Which tensors of the forward are "the states", and how they leave the library.
 * Replaces: the hook selection of the experts (hubert/expert.py:36-43: layer inputs + encoder output), wav2vec2's
 * feature_selection (wav2vec2/expert.py:81-93: layer_results[i][0] / [i][2]), the DistilHuBERT list
 * (distiller/expert.py:43-52), and — with `featurize` — Featurizer._weighted_sum applied as the encoder's epilogue
 * (nn/upstream.py:312-328, upstream/interfaces.py:221-249). */
enum { S3ENC_SEL_HIDDEN = 0,     /* default list: encoder_layers+1 states (DistilHuBERT: 1 + layers + pred_heads) */
       S3ENC_SEL_LAYER_OUT = 1,  /* "fairseq_layers": every layer's output (encoder_layers states) */
       S3ENC_SEL_FFN_OUT = 2 };  /* "fairseq_layers_before_residual": every layer's fc2 output before the residual */
typedef struct s3enc_forward_opts {
    int32_t selection;       /* S3ENC_SEL_* */
    int32_t out_dtype;       /* S3ENC_F32, or the handle's own 16-bit compute dtype: states are then written as 16-bit
                              * values (half the slab bytes and half the data-parallel all-gather); ignored with featurize */
    int32_t featurize;       /* 1: write ONLY  out[b][t][:] = sum_i feat_w[i] * (feat_normalize ? layer_norm(state_i) : state_i)
                              * as one fp32 (B, T, D) block — the states themselves never leave the workspace */
    int32_t feat_normalize;  /* F.layer_norm(state, (D,)) (no affine, eps 1e-5) before the sum */
    const float* feat_w;     /* host, one weight per state of the selection (softmax already applied; 0 = unselected) */
} s3enc_forward_opts;
/* number of states of a selection (the leading extent of `out` when not featurizing) */
int s3enc_num_states(s3enc_handle h, int32_t selection, int32_t* n);
/* s3enc_forward with options; opts == NULL is s3enc_forward.  out: device; state i is the contiguous (B, T, D) block at
 * out + i*layer_stride ELEMENTS of out_dtype (featurize: one fp32 (B, T, D) block, layer_stride ignored). */
int s3enc_forward_ex(s3enc_handle h, const float* const* wavs, const int64_t* lengths, int32_t B, int64_t n_max,
                     const s3enc_forward_opts* opts, void* out, int64_t layer_stride, void* stream);

/* Numerical health of the forwards of this handle (ABI 6).  Replaces: nothing the reference has as a call — its
 * experts return whatever ATen computed and the regression test inspects the tensors (test/test_upstream.py:118-136);
 * a 16-bit compute mode can overflow where fp32 cannot (fp16: |x| > 65504), so the library reports it instead of leaving
 * the caller to scan (NS, B, T, D) states.  `*status` = OR of S3ENC_STATUS_* over every FINISHED forward since the last
 * call; reading clears.  S3ENC_STATUS_NONFINITE: some row LayerNorm met a non-finite mean / variance — every hidden state
 * is a LayerNorm output (post-LN models) or the next LayerNorm's input (pre-LN residual streams), so an inf / NaN that
 * reaches the states is seen; the states of that forward must not be trusted.  wait != 0: block until every forward
 * enqueued so far has finished on its stream; wait == 0: never blocks — forwards still running are reported by the bit
 * S3ENC_STATUS_PENDING and their own bits by a later call (host-side cost of a poll: a few hipEventQuery; each forward
 * copies its word to a pinned ring of 8 slots, so nothing here touches the device). */
enum { S3ENC_STATUS_NONFINITE = 1, S3ENC_STATUS_PENDING = 1 << 30 };
int s3enc_forward_status(s3enc_handle h, int32_t wait, int32_t* status);

/* Optional: `n` = encoder_layers+1 hipEvent_t handles (as void*); the following forwards record events[l] on the
 * launch stream as soon as hidden_states[l] is final, so a communication stream can start the all-gather of layer l
 * while later layers are still computing.  n = 0 clears.  The events stay owned by the caller. */
int s3enc_set_layer_events(s3enc_handle h, void* const* events, int32_t n);

/* ---- multi-GPU: the exchange step of the data-parallel path ---------------------------------------
 * One rank per GPU (process or thread).  Rank r encodes its contiguous block of the batch padded to the GLOBAL n_max
 * (s3enc_forward's n_max argument — the only coupling between utterances) into a (states, shard, T, D) slab; the slabs are
 * re-assembled into (states, world * shard, T, D), rank order = input order, with one RCCL all-gather per state over xGMI.
 * Replaces: nothing in the reference's upstream path (run_downstream.py:166-168 only wraps trainable upstreams in DDP);
 * it is what torch.distributed.all_gather_into_tensor does for the Python binding (s3prl_amd/parallel.py).
 *   s3enc_comm_unique_id   rank 0 draws the 128-byte RCCL id and hands it to the other ranks by any side channel
 *   s3enc_comm_init_rank   collective over all ranks; `device` = the rank's GPU.  world = 1 is valid (gathers are copies).
 *   s3enc_comm_allgather_states
 *       send / recv: device; state l is `bytes_per_state` bytes at send + l * send_state_stride, and lands as rank r's block
 *       at recv + l * recv_state_stride + r * bytes_per_state (strides in BYTES; recv_state_stride >= world * bytes_per_state).
 *       ready_events: NULL, or n_states hipEvent_t (as void*) — the events given to s3enc_set_layer_events: the gather of
 *       state l then starts as soon as the encoder has recorded event l, on the communicator's own stream, while the
 *       remaining layers still compute.  On return the gathers are enqueued and `stream` is ordered after the last of them. */
typedef struct s3enc_comm_s* s3enc_comm;
int s3enc_comm_version(int32_t* version); /* RCCL's version code; fails (with the reason) when librccl cannot be loaded */
int s3enc_comm_unique_id(void* id128);
int s3enc_comm_init_rank(const void* id128, int32_t world, int32_t rank, int32_t device, s3enc_comm* out);
int s3enc_comm_info(s3enc_comm c, int32_t* world, int32_t* rank);
int s3enc_comm_allgather_states(s3enc_comm c, const void* send, int64_t send_state_stride, void* recv, int64_t recv_state_stride,
                                int32_t n_states, int64_t bytes_per_state, void* const* ready_events, void* stream);
/* The same exchange with the algorithm chosen by the caller (ABI 5):
 *   S3ENC_EXCHANGE_COLLECTIVE  one ncclAllGather per state — what s3enc_comm_allgather_states issues; RCCL picks ring / direct;
 *   S3ENC_EXCHANGE_DIRECT      per state one group of world-1 ncclSend / ncclRecv pairs (peer = rank +- p): xGMI is
 *                              point-to-point, every pair of GPUs has its own link, all 7 are driven at once.
 * Identical result bytes; replaces nothing in the reference (see above).  With world = 1 both are one device copy. */
#define S3ENC_EXCHANGE_COLLECTIVE 0
#define S3ENC_EXCHANGE_DIRECT 1
/* S3ENC_EXCHANGE_COPY (ABI 7): the same bytes in the same places moved by the COPY ENGINES — no RCCL kernel, no compute unit beside
 * the encoder's GEMMs.  Every rank registers ONE receive slab (s3enc_comm_copy_export: the `recv` of every later exchange), the ranks
 * swap the S3ENC_COPY_HANDLE_BYTES blobs by any side channel (hipIpcGetMemHandle inside: one PROCESS per GPU) and map each other's slab
 * and mailbox (s3enc_comm_copy_attach).  Per exchange and peer: one hipMemcpyAsync per state into the peer's slab, on that peer's own
 * stream (all xGMI links at once) and behind the encoder's "state l final" event; two 64-bit sequence numbers per pair order the
 * exchange (ready-to-receive, data-has-landed; csrc/comm.hip).  The waits are one-wave polls with a deadline (S3ENC_COPY_DEADLINE_MS,
 * default 5000): s3enc_comm_copy_status reports a missed one instead of a hung GPU.  A communicator for this form alone needs no RCCL:
 * s3enc_comm_init_local. */
#define S3ENC_EXCHANGE_COPY 2
#define S3ENC_COPY_HANDLE_BYTES 256
int s3enc_comm_init_local(int32_t world, int32_t rank, int32_t device, s3enc_comm* out);
int s3enc_comm_copy_export(s3enc_comm c, void* recv_slab, int64_t bytes, void* handle_out);
int s3enc_comm_copy_attach(s3enc_comm c, const void* handles /* world x S3ENC_COPY_HANDLE_BYTES, in rank order */);
int s3enc_comm_copy_status(s3enc_comm c, int32_t* status);
/* optional: marks, on `stream`, the point behind which nobody reads the slab's current contents any more — call it before enqueueing
 * the next forward, and that forward's per-state pushes overlap it (without it the next exchange counts from its own call) */
int s3enc_comm_copy_release(s3enc_comm c, void* stream);
int s3enc_comm_exchange_states(s3enc_comm c, int32_t algo, const void* send, int64_t send_state_stride, void* recv,
                               int64_t recv_state_stride, int32_t n_states, int64_t bytes_per_state, void* const* ready_events,
                               void* stream);
int s3enc_comm_destroy(s3enc_comm c);

/* Same, for a zero-padded (B, row_stride) device buffer (what pad_sequence builds, hubert/expert.py:66). */
int s3enc_forward_padded(s3enc_handle h, const float* pcm, int64_t row_stride, const int64_t* lengths, int32_t B,
                         int64_t n_max, float* out, int64_t layer_stride, void* stream);

/* ---- measurement ------------------------------------------------------------------------------------
 * With profiling on, the kernel launches of the next forwards are bracketed by HIP events on the launch stream
 * (on = 1: every kernel; on = 2: only the GEMM launches — two events cost ~2.6 us, which for ~290 launches per forward
 * is 2 % of an fp32 and 10 % of a bf16 batch, so a TIMED region profiles only its dominant kernel);
 * s3enc_profile_read synchronises and returns per-kernel-kind totals. */
int s3enc_profile_enable(s3enc_handle h, int32_t on);
int s3enc_profile_reset(s3enc_handle h);
typedef struct s3enc_profile_entry {
    char name[48];
    int64_t launches;
    double ms;     /* summed event time */
    double flops;  /* algorithmic FLOPs of those launches (2*M*N*K etc.; 0 for byte-bound kernels) */
    double bytes;  /* algorithmic HBM bytes of those launches (operands read once + outputs written once) */
} s3enc_profile_entry;
int s3enc_profile_read(s3enc_handle h, s3enc_profile_entry* entries, int32_t max_entries, int32_t* n_entries);

/* Copy an intermediate of the LAST forward to the host as fp32 (test hook): the last three conv layers
 * ("conv4".."conv6" for the 7-layer stack), "feat_ln", "proj", "posconv", "qkv0", "attn0" ("qkv0" / "attn0" name the buffers: after a
 * whole forward they hold the LAST layer's contents).  Synchronises.
 * Diagnostic (tools/two_stream_probe.py --taps): with the environment variable S3ENC_DEBUG_STOP = k set when the library loads, every
 * forward ends behind stage k — 1 + i: conv layer i, 20: the feature LayerNorm, 21: post_extract_proj, 22: the positional conv, 30 .. 34:
 * layer 0's q|k|v / attention / out_proj / LayerNorm / fc1, 40 + l: in front of layer l — writes NO states, and keeps "conv0" .. as taps
 * too (a conv tap is whole only for the last two conv layers that ran: the stack ping-pongs between two buffers). */
int s3enc_debug_tap(s3enc_handle h, const char* name, float* host_out, int64_t max_elems, int64_t* n_elems);
/* Measurement hook (no reference counterpart): `workgroups` x `threads` idle threads that hold their CU slots for `milliseconds` on
 * `stream` — a stand-in for a collective's channel kernels running beside the encoder (bench.py --steal-cus). */
int s3enc_debug_occupy_cus(int32_t workgroups, int32_t threads, double milliseconds, void* stream);  /* milliseconds <= 10000 */
/* Measurement hook (no reference counterpart): writes {shader-clock counter, reference-clock counter, reference rate in kHz} of XCD 0
 * to three device uint64 on `stream`.  Two samples around a region give its average shader clock: d(out[0]) / d(out[1]) x out[2]
 * (hipDeviceAttributeWallClockRate, 100 MHz) — bench.py's `clock_ghz`, so that a slow box reads as a slow box. */
int s3enc_debug_clock_sample(uint64_t* out3_device, void* stream);

/* Process-wide tuning knobs (kernel-variant selection for A/B measurements; results are unchanged, except "gelu32"):
 *   "gemm_variant": gemm.hip's 128x128 kernel: bit 0 = 64-byte K stages (else 128), bit 1 = LDS-DMA staging, bit 2 = no
 *                   XCD-aware tile order;
 *   "gemm32_big":   exact-fp32 tile kernel (gemmt.hip): 0 off (gemm.hip), 1 = tile height by shape (default), 2 / 3 / 4 / 5 =
 *                   force 256 / 192 / 128 / 64 rows;  "gemm_x3_tile": the same for S3ENC_F32X3 (1 = only small shapes);
 *   "gemm16_big":   large-tile kernel of the 16-bit modes: 0 off, 1 = one workgroup per CU (256x256 or 192x256 tiles by CU
 *                   utilisation; 5 / 6 force either), 2 = 128x256, 4 = 128x256 with a 3-stage ring (two workgroups per
 *                   CU), 7 = mode 1's tiles walked by one persistent workgroup per CU, 8 = 7 with the epilogue's stores left
 *                   draining under the next tile's first K steps (profiles/r04_gemm16_overlap.md), 9 / 10 = 7 / 8 with the
 *                   row-per-lane epilogue (no LDS transpose) wherever it is legal, 3 = chosen by shape (default: 7);
 *   "gemm16_rows":  1 (default) = under mode 7 the GELU epilogues with a 16-bit output (conv1-5, fc1) take the row-per-lane form
 *                   (profiles/r04_gemm16_epilogue.md), 0 = never;
 *   "gemm16_pp":    under mode 7 / 9, which fragment steps of a K step carry a wave's LDS-DMA pieces: 0 (default) = steps 0 / 1 for
 *                   every wave, non-zero = waves 4-7 (the SIMD partners of waves 0-3) steps 1 / 2 instead — a schedule change only,
 *                   measured within 1-3 % of the default (profiles/r05_gemm16_loop_probe.md);
 *   "gemm16_mx":    S3ENC_F16X2 only, bit mask: which GEMMs take their second weight term as an MX-fp4 image on the scaled-MFMA pipe
 *                   (gemm16.hip MXW: 4.8e-5 of weight error per GEMM instead of 5e-7) where the shape allows — 1 conv1, 2 q|k|v, 4 fc1,
 *                   8 fc2, 16 = (read at s3enc_create) also weights whose 256-row tiling needs fewer CU-rounds at the reference batch — which weights
 *                   take it is decided per weight, never per batch; default 14; 0 = two fp16 terms everywhere.  Results
 *                   differ at the 1e-5 ... 1e-4 level (profiles/r05_mx_second_term.md);
 *                   The mask is read at s3enc_create: only the kinds it names get an image (narrowing it on a live handle works,
 *                   widening needs a new handle);
 *   "forward_chain": 1 (default) = a forward of a 16-bit / split-precision handle (S3ENC_BF16, S3ENC_F16, S3ENC_F16X2, S3ENC_F32X3) starts, on the
 *                   device, behind the previous such forward of ANY handle of the process on that device (hipStreamWaitEvent on one event per
 *                   device; no host wait; with one handle on one stream it adds nothing to stream order).  Forwards of several handles
 *                   overlapping on four or more streams were measured NOT bit-stable in those modes — rare rows a few 16-bit ulps off, never
 *                   with two streams, never in S3ENC_F32 (profiles/r06c_concurrent_forwards.md; round 6's fourth session traced every such row to the first
 *                   conv layer's kernel computing single frames wrong while waves that issue the double-rate 16-bit MFMA (v_mfma_f32_32x32x16) share its SIMD —
 *                   a cross-wave effect; the 16-bit tile-GEMM kernels therefore claim their SIMD's whole register file, r06d) — so a
 *                   serving process that keeps one encoder per model or per worker thread gets every utterance's own bits by default.
 *                   0 = such forwards may overlap (small batches then fill the chip together: four 8 x 10 s forwards 12.1 -> 8.5 ms);
 *                   (round 6, fourth session: the differing rows originate in the first conv layer's kernel of the 16-bit modes and nowhere
 *                   else — profiles/r06d_concurrent_forwards_exclusions.md);
 *   "conv0_fast":   16-bit outputs of the first conv layer: 1 (default) = packed fp32 taps and the packed one-transcendental GELU, 0 = scalar taps
 *                   and libm erff (about twice that kernel's time; results a few 16-bit ulps apart).  With 0 — and forward_chain = 0 — 32 of 32
 *                   measured runs of four / eight overlapping forwards kept their bits; why is not understood, so this is a diagnostic
 *                   switch and the forward chain stays the protection;
 *   "comm_self_p2p": S3ENC_EXCHANGE_DIRECT test hook: 1 = a rank's own block travels as an ncclSend-to-self / ncclRecv-from-self pair
 *                   inside the state's group instead of a device copy (how the all-pairs code executes on a one-GPU box); default 0;
 *   "fp16x2_conv1_f32": S3ENC_F16X2, read at s3enc_create: 1 = conv0 writes fp32 activations and conv1 reads them through the three-term
 *                   GEMM like conv2.. already do (removes the mode's last fp16 rounding inside the conv stack: the worst weight seed of
 *                   profiles/r06_parity_seeds.md moves from 8.3e-4 to ~7e-4, for conv1 at the three-term rate); default 0;
 *   "attn_persist": 1 = the attention kernels run as persistent workgroups that fetch the next (batch, head, query block) item's
 *                   operands under the current item's last key tile, 0 (default) = the one-shot grid; bit-identical results; a measured
 *                   prototype that lost (profiles/r06_attention_persist.md), kept for re-measurement;
 *   "reserve_cus":  CUs the persistent one-workgroup-per-CU GEMM of the 16-bit modes leaves out of its grid (default 0; a measurement
 *                   knob — leaving CUs to a collective's channel kernels costs more than sharing them: profiles/r05_cu_contention.md);
 *   "conv0_nt":     1 (default) = the fp32 conv0 kernel writes its activation with non-temporal stores, 0 = plain stores;
 *   "ws_inplace":   1 (default) = post-LN layers run LayerNorm 1 and fc2 in place on one fp32 workspace buffer, 0 = two buffers;
 *   "ln1_fold":     16-bit modes, post-LN layers: 1 (default) = LayerNorm 1 writes its 16-bit output and the rows' (mean, rstd) only and
 *                   fc2's epilogue rebuilds the fp32 rows it adds from the row it normalised; 0 = LayerNorm 1 writes them; same bits;
 *   "ln_preload":   1 (default) = the row LayerNorm fetches gamma / beta together with the row instead of behind the reductions; same bits;
 *   "ln_rows":      rows per wave of the row LayerNorm: 1 (default), 2 = two rows' loads in flight (launches of >= 8192 rows); same
 *                   bits; measured 2-3 % slower, kept for re-measurement;
 *   "gn_lag_one_block": 1 (default) = the GroupNorm lag sums of the waveform come from one workgroup per (4096-frame chunk,
 *                   utterance) over an LDS-staged window, 0 = the earlier k0-workgroups kernel; same bits
 *                   (profiles/r06b_gn_stats.md);
 *   "gelu32":       S3ENC_F32 only: 1 = the one-transcendental erf-GELU every mode uses (default; csrc/common.h gelu_fast: as
 *                   close to an fp64 erf-GELU as 0.5 x (1 + erff(x / sqrt 2)) evaluated in fp32), 0 = libm erff — results
 *                   differ in the last bits. */
int s3enc_set_tuning(const char* key, int32_t value);
/* The same keys for ONE handle: the handle starts from the process-wide values as they are at the first call and keeps its own
 * copy from then on; its forwards use that copy (per calling thread), other handles and the s3enc_op_* entry points do not. */
int s3enc_set_handle_tuning(s3enc_handle h, const char* key, int32_t value);

/* ---- single-kernel entry points (parity tests of each HIP kernel against the oracle) -------------------
 * All pointers are device pointers; dtype is S3ENC_F32/BF16/F16 for the 16-bit-capable operands
 * (16-bit data are raw uint16). */

/* out[b][m][n] = epilogue( sum_k A[b][m][k] * W[n][k] ):  A rows start at A + b*a_batch_stride + m*lda
 * (elements; lda < K expresses an overlapping strided-conv window), W is (N, K) row-major.
 * epilogue: + bias[n]; GELU if act; + residual (fp32, same indexing as out32); rows m >= row_limit[b] -> 0.
 * Writes out32 (fp32) and/or out16 (dtype) when non-NULL.  dtype S3ENC_F32X3: fp32 A / W / out32, three bf16 MFMAs per
 * product (K % 32 == 0, M, N >= 128; synchronises).  dtype S3ENC_F16X2: fp16 A and out16, W is the (N, 2K) fp16 image
 * [hi(K) | lo(K)] per row with w = hi + lo (K % 64 == 0, else only the hi half is used). */
int s3enc_op_gemm(int32_t dtype, const void* A, int64_t lda, int64_t a_batch_stride, const void* W,
                  const float* bias, int32_t M, int32_t N, int32_t K, int32_t batches, int32_t act,
                  const float* residual, const int32_t* row_limit, float* out32, void* out16, int64_t ldo,
                  int64_t o_batch_stride, void* stream);

/* conv0 of the feature extractor with its normalisation and GELU (wav2vec2_model.py:2879-2906) on raw waveforms:
 * per-utterance waveform layer-norm if `normalize`; GroupNorm(C, C) over all L0 frames incl. the zero padding when
 * gn_gamma != NULL (statistics from the waveform's lag sums), else LayerNorm(C) per frame when ln_gamma != NULL.
 * wavs / lengths as in s3enc_forward; w0 (C, 10), bias (C or NULL), gamma / beta: device fp32; out: device
 * (B, L0, C) of `dtype`, L0 = (n_max - 10) / stride + 1. */
int s3enc_op_conv0(int32_t dtype, const float* const* wavs, const int64_t* lengths, int32_t B, int64_t n_max,
                   int32_t normalize, const float* w0, const float* bias, const float* gn_gamma, const float* gn_beta,
                   const float* ln_gamma, const float* ln_beta, int32_t C, int32_t stride, void* out, void* stream);
/* WavLM gate (wavlm/modules.py:535-549) from the attention input x (device fp32 (B, T, H*64)):
 * gate[b][h][t] = a * (b * grep_a[h] - 1) + 2,  a|b = sigmoid(sum4(grep_linear(x_head))).  grep_w (8, 64), grep_b (8),
 * grep_a (H): device fp32. */
int s3enc_op_wavlm_gate(const float* x, const float* grep_w, const float* grep_b, const float* grep_a, int32_t B, int32_t T,
                        int32_t H, float* gate, void* stream);

/* Row LayerNorm over C (eps 1e-5, biased variance), optional erf-GELU, fp32 in, fp32 and/or dtype out. */
int s3enc_op_layernorm(int32_t dtype, const float* x, const float* gamma, const float* beta, int32_t rows,
                       int32_t C, int32_t act, float* out32, void* out16, void* stream);

/* Multi-head self-attention on a fused (B*T, 3D) q|k|v buffer, head_dim 64, keys >= valid[b] masked.  q arrives
 * pre-scaled by head_dim^-0.5 — for dtype S3ENC_BF16 / S3ENC_F16 by head_dim^-0.5 * log2(e): the 16-bit kernels
 * work on base-2 scores, the engine folds either factor into W_q at s3enc_create; optional WavLM gated relative-position bias: score += gate[b][h][i] * table[h][clamp(j-i, -R, R) + R]
 * with a (H, 2R+1) table (the bucket of wavlm/modules.py:418-446 is constant for |j-i| >= max_distance, so
 * R = max_distance serves every T). */
int s3enc_op_attention(int32_t dtype, const void* qkv, void* out, const int32_t* valid, int32_t B, int32_t T,
                       int32_t H, const float* bias_table, int32_t table_R, const float* gate, void* stream);

/* Convolutional position embedding + residual: out = x + GELU(SamePad(Conv1d(D, D, K, padding=K/2, groups=G)(x)) + bias)
 * (make_conv_pos / SamePad, wav2vec2_model.py:2937-2953,1797-1808).  x, out: device fp32 (B, T, D); w_host: HOST fp32
 * (D, D/G, K), the nn.Conv1d weight with weight_norm already folded; packed and uploaded inside.  Synchronises. */
int s3enc_op_posconv(int32_t dtype, const float* x, const float* w_host, const float* bias, int32_t B, int32_t T, int32_t D,
                     int32_t G, int32_t K, float* out, void* stream);

/* ---- Featurizer: the consumer of hidden_states (next row of the path) --------------------------------
 * Replaces Featurizer._weighted_sum (s3prl/nn/upstream.py:312-328; upstream/interfaces.py:221-249):
 *   out = sum_l w[l] * (normalize ? F.layer_norm(h_l, (D,)) : h_l),   w = softmax(weights) computed by the caller.
 * hs: device fp32, layer l is the (rows, D) block at hs + l*layer_stride (the slab s3enc_forward writes);
 * w: HOST array of L floats (0 = layer not selected); out: device fp32 (rows, D). */
int s3enc_weighted_sum(const float* hs, int64_t layer_stride, int32_t L, const float* w, int32_t normalize, int64_t rows,
                       int32_t D, float* out, void* stream);
/* Gradient of the above w.r.t. w (the upstream is frozen): grad_w[l] = sum <grad_out, hn_l>; grad_w: device, L floats.
 * scratch: device, s3enc_weighted_sum_backward_scratch(rows, L) doubles, owned by the caller (stream-ordered reuse);
 * asynchronous on `stream`. */
int64_t s3enc_weighted_sum_backward_scratch(int64_t rows, int32_t L);
int s3enc_weighted_sum_backward(const float* hs, int64_t layer_stride, int32_t L, int32_t normalize, int64_t rows,
                                int32_t D, const float* grad_out, float* grad_w, double* scratch, void* stream);

/* ---- the `fbank` baseline upstream (BASELINE configs[0]) --------------------------------------------------------
 * Replaces get_extracter(fbank.yaml) + UpstreamExpert.forward of upstream/baseline (extracter.py:32-90,
 * expert.py:46-79): torchaudio.compliance.kaldi.fbank -> delta, delta-delta -> CMVN over time -> pad_sequence. */
typedef struct s3enc_fbank_config {
    int32_t sample_rate;      /* 16000 */
    int32_t num_mel_bins;     /* 80   (fbank.yaml) */
    float frame_length_ms;    /* 25 */
    float frame_shift_ms;     /* 10 */
    float preemphasis;        /* 0.97 (kaldi default) */
    int32_t delta_order;      /* 2 */
    int32_t delta_win_length; /* 5 */
    int32_t use_cmvn;         /* 1 */
    float cmvn_eps;           /* 1e-10 (extracter.py:80) */
} s3enc_fbank_config;
/* frames of an n-sample utterance (snip_edges): 1 + (n - window) / shift, 0 if shorter than one window */
int s3enc_fbank_num_frames(const s3enc_fbank_config* cfg, int64_t n_samples, int32_t* frames);
/* wavs: host array of B device pointers (borrowed); out: device fp32 (B, T_max, num_mel_bins*(delta_order+1)),
 * zero beyond each utterance's frames (pad_sequence); T_max >= the longest utterance's frame count. */
int s3enc_fbank_forward(const s3enc_fbank_config* cfg, const float* const* wavs, const int64_t* lengths, int32_t B,
                        float* out, int64_t T_max, int32_t device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* S3ENC_H */
