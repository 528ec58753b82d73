// engine.hip — the C ABI of libs3enc (include/s3enc.h): handle, weight packer, workspace, forward schedule.
//
// Forward schedule (all launches on the caller's stream, no host synchronisation inside):
//   table upload -> wav stats -> [GroupNorm lag sums] -> conv0 -> conv1..6 (implicit GEMM, GELU fused)
//   -> LayerNorm(C) -> post_extract_proj (+bias, padded frames zeroed) -> pos-conv (+GELU, +residual)
//   -> [post-LN: encoder.layer_norm] -> NL x { q|k|v GEMM, attention, out_proj(+residual), LN, fc1(+GELU),
//   fc2(+residual), LN }  with every hidden-state tap written straight into the caller's (NL+1, B, T, D) slab.
//   (multires-HuBERT continues in multires.hip after post_extract_proj; the single-kernel entry points are in ops.hip)
#include "engine_internal.h"

namespace s3e {

thread_local std::string g_err;

long conv_len(const s3enc_config& c, long n, int upto /*exclusive*/) {
    for (int i = 0; i < upto; ++i) n = n >= c.conv_kernel[i] ? (n - c.conv_kernel[i]) / c.conv_stride[i] + 1 : 0;
    return n;
}

int valid_frames(const s3enc_config& c, long length, long n_max) {
    const long T = conv_len(c, n_max, c.n_conv);
    if (T <= 0) return 0;
    long v;
    if (c.family == S3ENC_WAV2VEC2 || c.family == S3ENC_DISTILLER) {
        v = conv_len(c, length, c.n_conv);  // wav2vec2_model.py:2652-2669; distiller/model.py:271-285
    } else {
        const long chunk = n_max / T;  // hubert_model.py:454-464
        v = (length + chunk - 1) / chunk;
    }
    if (v > T) v = T;
    if (v < 0) v = 0;
    return (int)v;
}
int num_states(const s3enc_config& c, int selection) {
    if (c.family == S3ENC_MULTIRES) return c.encoder_layers + 2 * c.mr_pairs + 1;  // per block: layer inputs + its output
    if (selection == S3ENC_SEL_HIDDEN) return c.encoder_layers + 1 + (c.family == S3ENC_DISTILLER ? c.pred_heads : 0);
    return c.encoder_layers;
}
}  // namespace s3e

namespace {

// ---- checkpoint lookup ---------------------------------------------------------------------------------------
struct Ckpt {
    std::map<std::string, const s3enc_tensor*> m;
    const s3enc_tensor* get(const std::string& n) const {
        auto it = m.find(n);
        return it == m.end() ? nullptr : it->second;
    }
};
long numel(const s3enc_tensor* t) {
    long n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}
bool fetch(const Ckpt& c, const std::string& name, long expect, std::vector<float>& out, std::string& err) {
    const s3enc_tensor* t = c.get(name);
    if (!t) {
        err = "checkpoint is missing tensor '" + name + "'";
        return false;
    }
    if (numel(t) != expect) {
        err = "tensor '" + name + "' has " + std::to_string(numel(t)) + " elements, expected " + std::to_string(expect);
        return false;
    }
    out.assign(t->data, t->data + expect);
    return true;
}
// WavLM bucket table (wavlm/modules.py:418-462): table[h][rel + R] = E[bucket(rel)][h] for rel = key - query in [-R, R]
void build_rel_table(const s3enc_config& c, const std::vector<float>& emb, int R, std::vector<float>& table) {
    const int H = c.heads, nb = c.num_buckets / 2, max_exact = nb / 2;
    const int W = 2 * R + 1;
    table.resize((size_t)H * W);
    const float denom = (float)std::log((double)c.max_distance / (double)max_exact);
    for (int idx = 0; idx < W; ++idx) {
        const int rel = idx - R;
        int bucket = rel > 0 ? nb : 0;
        const int a = rel < 0 ? -rel : rel;
        if (a < max_exact) {
            bucket += a;
        } else {
            float v = std::log((float)a / (float)max_exact);
            v = v / denom;
            v = v * (float)(nb - max_exact);
            int large = max_exact + (int)v;
            if (large > nb - 1) large = nb - 1;
            bucket += large;
        }
        for (int h = 0; h < H; ++h) table[(size_t)h * W + idx] = emb[(size_t)bucket * H + h];
    }
}

int check_config(const s3enc_config& c) {
    if (c.family < 0 || c.family > 4) return fail("config: unknown family");
    if (c.family == S3ENC_MULTIRES) {
        if (c.mr_pairs < 1 || c.mr_pairs > S3ENC_MAX_RES - 1) return fail("config: mr_pairs out of range");
        const int k = c.mr_kernel;
        if (k < 1 || k > 15 || !(k & 1)) return fail("config: mr_kernel must be odd and <= 15");
        int total = 0;
        for (int i = 0; i < 2 * c.mr_pairs + 1; ++i) {
            if (c.mr_layers[i] < 1) return fail("config: every multires block needs at least one layer");
            total += c.mr_layers[i];
        }
        if (total != c.encoder_layers) return fail("config: encoder_layers must equal the sum of mr_layers");
        for (int i = 0; i < 2 * c.mr_pairs; ++i) {
            const int r = c.mr_ratios[i];
            if (r < 1 || r > 4 || (k - 1) % r) return fail("config: every multires rate must divide mr_kernel - 1");
            if (c.mr_plain && !(i & 1) && r != 1) return fail("config: mr_plain needs rate pairs of the form (1, d)");
        }
        if (c.rel_pos || c.pred_heads || c.pos_conv_depth > 1 || c.no_feature_layer_norm)
            return fail("config: multires-HuBERT takes none of the WavLM / DistilHuBERT / data2vec options");
    }
    if (c.pred_heads < 0 || c.pred_heads > 16) return fail("config: pred_heads out of range");
    if (c.pred_heads && c.family != S3ENC_DISTILLER) return fail("config: pred_heads is a DistilHuBERT feature");
    if (c.n_conv < 2 || c.n_conv > S3ENC_MAX_CONV) return fail("config: n_conv out of range");
    if (c.conv_kernel[0] != 10) return fail("config: the conv0 kernel is specialised for kernel width 10");
    if (c.conv_dim % 32 || c.conv_dim > 1024) return fail("config: conv_dim must be a multiple of 32, <= 1024");
    if (c.heads <= 0 || c.embed_dim != c.heads * 64) return fail("config: head_dim must be 64");
    if (c.ffn_dim % 8) return fail("config: ffn_dim must be a multiple of 8");
    if (c.conv_pos_groups <= 0 || c.embed_dim % c.conv_pos_groups) return fail("config: embed_dim % conv_pos_groups != 0");
    const int dg = c.embed_dim / c.conv_pos_groups;
    if (dg != 32 && dg != 48 && dg != 64) return fail("config: embed_dim/conv_pos_groups must be 32, 48 or 64");
    if (c.conv_pos < 1 || c.conv_pos > 256) return fail("config: conv_pos out of range");
    if (c.compute_dtype < 0 || c.compute_dtype > 4) return fail("config: unknown compute_dtype");
    if (c.encoder_layers < 1) return fail("config: encoder_layers < 1");
    if (c.rel_pos && c.family != S3ENC_WAVLM) return fail("config: rel_pos is a WavLM feature");
    if (c.rel_pos && (c.num_buckets < 4 || c.max_distance <= c.num_buckets / 4))
        return fail("config: bad num_buckets / max_distance");
    if (c.rel_pos && c.max_distance > 8192) return fail("config: max_distance > 8192");
    if (c.pos_conv_depth < 0 || c.pos_conv_depth > 16) return fail("config: pos_conv_depth out of range");
    if (!(c.wav_norm_eps >= 0.f) || c.wav_norm_eps > 1.f) return fail("config: wav_norm_eps out of range");
    if (c.pos_conv_depth > 1 && c.family != S3ENC_WAV2VEC2) return fail("config: pos_conv_depth > 1 is the data2vec-audio encoder (wav2vec2 family)");
    return 0;
}
}  // namespace

extern "C" {

int s3enc_version(void) { return S3ENC_VERSION; }
const char* s3enc_last_error(void) { return g_err.c_str(); }

int s3enc_create(const s3enc_config* cfg, const s3enc_tensor* tensors, int32_t n_tensors, int32_t device, s3enc_handle* out) {
    if (!cfg || !tensors || !out) return fail("s3enc_create: null argument");
    *out = nullptr;
    if (check_config(*cfg)) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("s3enc_create: no HIP device visible — libs3enc has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("s3enc_create: device index out of range");
    DeviceGuard dg(device);
    if (!dg.ok) return fail("s3enc_create: hipSetDevice failed");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(std::string("s3enc_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);

    Ckpt ck;
    for (int i = 0; i < n_tensors; ++i)
        if (tensors[i].name && tensors[i].data) ck.m[tensors[i].name] = &tensors[i];

    s3enc_encoder* e = new s3enc_encoder();
    e->cfg = *cfg;
    e->device = device;
    e->x3 = cfg->compute_dtype == 3;  // S3ENC_F32X3: the fp32 data flow, GEMMs through gemm_x3.hip
    e->x2 = cfg->compute_dtype == 4;  // S3ENC_F16X2: the fp16 data flow, GEMM weights split into two fp16 terms
    e->dtype = e->x3 ? (int)F32 : (e->x2 ? (int)F16 : cfg->compute_dtype);
    e->es = e->dtype == F32 ? 4 : 2;
    const s3enc_config& c = e->cfg;
    const int C = c.conv_dim, D = c.embed_dim, F = c.ffn_dim, H = c.heads;
    const bool multires = c.family == S3ENC_MULTIRES;
    const std::string enc0 = multires ? "encoders.0" : "encoder";  // only the first block keeps its positional conv
    std::string err;
    std::vector<float> t, t2;
#define GET(name, n, vec)                  \
    if (!fetch(ck, name, n, vec, err)) {   \
        delete e;                          \
        return fail(err);                  \
    }
#define UP(call)                                                     \
    do {                                                             \
        hipError_t _e = (call);                                      \
        if (_e != hipSuccess) {                                      \
            delete e;                                                \
            return fail(std::string("weight upload failed: ") + hipGetErrorString(_e)); \
        }                                                            \
    } while (0)

    // a GEMM weight in the handle's operand format; S3ENC_F16X2: + the MX-fp4 image of its lo term for the weights gemm16.hip's MXW K
    // step pays for (gemm16_mx_weight_rule: decided per weight, never per batch; wsplit_of finds the image by the weight's pointer).
    // Only the kinds the tuning key gemm16_mx names AT CREATE TIME get an image (default 14: conv1's would never be read, and with
    // gemm16_mx = 0 none would — seconds of host packing and the device bytes on a large model): the mask can be narrowed after
    // s3enc_create, widening it needs a new handle.
#define UPW(buf, vec, N_, K_, KIND)                                   \
    do {                                                              \
        UP(upload_gemm_w(buf, vec, N_, K_, e->dtype, e->x2));         \
        if (e->x2 && (tuning().gemm16_mx & ((KIND) | 16)) &&          \
            (gemm16_mx_weight_rule(N_, K_) || ((tuning().gemm16_mx & 16) && !((K_) & 127) && (N_) >= 128))) { \
            std::unique_ptr<MxImage> img(new MxImage());              \
            UP(upload_mx4_lo(*img, vec, N_, K_));                     \
            img->kind = KIND;                                         \
            e->mx_images[(buf).p] = std::move(img);                   \
        }                                                             \
    } while (0)

    // ---- conv feature extractor ----
    // (round 5: layer-norm extractors too — their conv outputs are LayerNorm'd + GELU'd, which renormalises the scale but not the
    // relative rounding noise: six stacked fp16 roundings are 6.0e-4 on HuBERT-large and, amplified by the bias-sharpened
    // attention of WavLM-large, 1.07e-3 alone on pretrained-like statistics — tools/fp16_error_budget.py, profiles/r05_fp16_cliff.md)
    if (e->x2 && !c.no_feature_layer_norm && c.n_conv >= 3) {
        bool ok = true;  // every conv from the second on must be a shape the three-term GEMM takes
        for (int i = 2; i < c.n_conv; ++i) ok = ok && x3_shape_ok(C, (long)c.conv_kernel[i] * C, (long)c.conv_stride[i] * C);
        if (ok) e->x2_conv_f32_from = 2;
        // round 6, tuning key fp16x2_conv1_f32 (read here): conv0 writes fp32 and conv1 — half of the conv stack's FLOPs — takes the
        // three-term GEMM as well.  What is left of the conv term after the hybrid above is conv0's own fp16 output (the largest single
        // site on the worst seed of the weight-seed sweep after the LayerNorm outputs: WavLM-large seed 1, 8.3e-4 -> 7.2e-4 emulated,
        // profiles/r06_parity_seeds.md); it costs conv1 at the three-term rate on fp32 rows — opt-in, the default stays inside 1e-3.
        if (ok && tuning().fp16x2_conv1_f32 && x3_shape_ok(C, (long)c.conv_kernel[1] * C, (long)c.conv_stride[1] * C)) e->x2_conv_f32_from = 1;
    }
    e->conv.resize(c.n_conv);
    for (int i = 0; i < c.n_conv; ++i) {
        const std::string p = "feature_extractor.conv_layers." + std::to_string(i);
        const int cin = i == 0 ? 1 : C, k = c.conv_kernel[i];
        GET(p + ".0.weight", (long)C * cin * k, t);
        if (i == 0) {
            UP(upload_f32(e->conv[i].w, t));  // [C][k]
        } else {
            // re-lay (Cout, Cin, k) tap-major: W'[co][j*Cin + ci], the K order of the channel-last implicit GEMM
            t2.resize(t.size());
            for (int co = 0; co < C; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int j = 0; j < k; ++j) t2[((long)co * k + j) * cin + ci] = t[((long)co * cin + ci) * k + j];
            if (e->x2_conv_f32_from && i >= e->x2_conv_f32_from) {
                UP(upload_gemm_w(e->conv[i].w, t2, C, (long)k * cin, e->dtype, e->x2));  // (read only if the three-term path declines)
            } else {
                UPW(e->conv[i].w, t2, C, (long)k * cin, 1);
            }
            if (e->x3 || (e->x2_conv_f32_from && i >= e->x2_conv_f32_from)) UP(upload_x3(e->conv[i].w3, t2, C, (long)k * cin));
        }
        if (c.conv_bias) {
            GET(p + ".0.bias", C, t);
            UP(upload_f32(e->conv[i].bias, t));
            e->conv[i].has_bias = true;
        }
        if (c.extractor_layer_norm) {
            GET(p + ".2.1.weight", C, t);
            UP(upload_f32(e->conv[i].lng, t));
            GET(p + ".2.1.bias", C, t);
            UP(upload_f32(e->conv[i].lnb, t));
        } else if (i == 0) {
            GET(p + ".2.weight", C, t);
            UP(upload_f32(e->gn_g, t));
            GET(p + ".2.bias", C, t);
            UP(upload_f32(e->gn_b, t));
        }
    }
    if (!c.no_feature_layer_norm) {
        GET("layer_norm.weight", C, t);
        UP(upload_f32(e->fln_g, t));
        GET("layer_norm.bias", C, t);
        UP(upload_f32(e->fln_b, t));
    }
    GET("post_extract_proj.weight", (long)D * C, t);
    UP(upload_gemm_w(e->proj_w, t, D, C, e->dtype, e->x2));
    e->x2_proj_f32 = e->x2 && !c.no_feature_layer_norm && c.family != S3ENC_DISTILLER && x3_shape_ok(D, C, C);
    if (e->x3 || e->x2_proj_f32) UP(upload_x3(e->proj_w3, t, D, C));
    GET("post_extract_proj.bias", D, t);
    UP(upload_f32(e->proj_b, t));

    // ---- positional conv: fold weight_norm(dim=2), then pack for the kernel of the compute dtype ----
    if (c.pos_conv_depth > 1) {
        // data2vec (wav2vec2_model.py:2995-3023): plain grouped convs of width max(3, conv_pos / depth), no weight_norm
        const int G = c.conv_pos_groups, Dg = D / G;
        int k = c.conv_pos / c.pos_conv_depth;
        k = k < 3 ? 3 : k;
        const bool k16 = e->dtype != F32 || e->x3;  // the implicit-GEMM kernel: even K with K * Dg a multiple of 128
        int kp = k;
        if (k16)
            while ((kp & 1) || ((long)kp * Dg) % 128) ++kp;
        e->pos_k = kp;
        e->pos_pad = k / 2;
        e->pos_ws.resize(c.pos_conv_depth);
        e->pos_bs.resize(c.pos_conv_depth);
        for (int i = 0; i < c.pos_conv_depth; ++i) {
            const std::string p = "encoder.pos_conv." + std::to_string(i) + ".0";
            std::vector<float> w, wp((size_t)D * Dg * kp, 0.f);
            GET(p + ".weight", (long)D * Dg * k, w);
            for (long r = 0; r < (long)D * Dg; ++r)
                for (int j = 0; j < k; ++j) wp[r * kp + j] = w[r * k + j];
            if (e->x3) {
                UP(upload_posconv_x3(e->pos_ws[i], wp, D, G, kp));
            } else {
                pack_posconv(wp, D, G, kp, e->dtype, t2);
                UP(upload_cvt(e->pos_ws[i], t2, e->dtype));
            }
            GET(p + ".bias", D, t);
            UP(upload_f32(e->pos_bs[i], t));
        }
        UP(upload_f32(e->ones, std::vector<float>(D, 1.f)));
        UP(upload_f32(e->zeros, std::vector<float>(D, 0.f)));
    } else {
        const int K = c.conv_pos, G = c.conv_pos_groups, Dg = D / G;
        std::vector<float> g, v;
        GET(enc0 + ".pos_conv.0.weight_g", K, g);
        GET(enc0 + ".pos_conv.0.weight_v", (long)D * Dg * K, v);
        std::vector<double> nrm(K, 0.0);
        for (long i = 0; i < (long)D * Dg; ++i)
            for (int k = 0; k < K; ++k) nrm[k] += (double)v[i * K + k] * v[i * K + k];
        for (int k = 0; k < K; ++k) nrm[k] = (double)g[k] / std::sqrt(nrm[k]);
        for (long i = 0; i < (long)D * Dg; ++i)
            for (int k = 0; k < K; ++k) v[i * K + k] = (float)(v[i * K + k] * nrm[k]);
        // the 16-bit / split-precision implicit GEMM wants an even tap count with K * Dg a multiple of 128: append zero
        // taps (the real kernel keeps its K / 2 left padding), as for the data2vec stack above
        int kp = K;
        if (e->dtype != F32 || e->x3)
            while ((kp & 1) || ((long)kp * Dg) % 128) ++kp;
        e->pos_k = kp;
        e->pos_pad = K / 2;
        if (kp != K) {
            std::vector<float> vp((size_t)D * Dg * kp, 0.f);
            for (long r = 0; r < (long)D * Dg; ++r)
                for (int j = 0; j < K; ++j) vp[r * kp + j] = v[r * K + j];
            v.swap(vp);
        }
        pack_posconv(v, D, G, kp, e->dtype, t2);
        UP(upload_cvt(e->pos_w, t2, e->dtype));
        if (e->x3) UP(upload_posconv_x3(e->pos_w3, v, D, G, kp));
        GET(enc0 + ".pos_conv.0.bias", D, t);
        UP(upload_f32(e->pos_b, t));
    }
    if (!multires) {
        GET("encoder.layer_norm.weight", D, t);
        UP(upload_f32(e->eln_g, t));
        GET("encoder.layer_norm.bias", D, t);
        UP(upload_f32(e->eln_b, t));
    }

    // ---- transformer layers ----
    // q *= head_dim^-0.5 is folded into W_q, b_q; the 16-bit attention kernels additionally want log2-domain scores
    // (attention.hip: softmax as exp2 with no per-score multiply), so those handles fold log2(e) in as well — one rounding
    // of the weight to the operand type either way.  The exact-fp32 and split-precision handles keep the power-of-two scale.
    const float qscale = (1.0f / std::sqrt((float)(D / H))) * (e->dtype != F32 ? 1.44269504088896340736f : 1.0f);
    e->x2_attn_f32 = e->x2 && !multires && x3_shape_ok(D, D, D);  // (the U-net keeps 16-bit)
    // one TransformerSentenceEncoderLayer named `p` (…layers.N); returns non-zero after fail() (e is already deleted)
    auto load_layer = [&](const std::string& p, LayerW& L) -> int {
        std::vector<float> w(3L * D * D), bb(3L * D);
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        for (int s = 0; s < 3; ++s) {
            GET(p + ".self_attn." + names[s] + ".weight", (long)D * D, t);
            GET(p + ".self_attn." + names[s] + ".bias", D, t2);
            const float sc = s == 0 ? qscale : 1.f;  // q *= head_dim^-0.5 folded into W_q, b_q
            for (long i = 0; i < (long)D * D; ++i) w[(long)s * D * D + i] = t[i] * sc;
            for (int i = 0; i < D; ++i) bb[(long)s * D + i] = t2[i] * sc;
        }
        UPW(L.wqkv, w, 3L * D, D, 2);
        if (e->x3) UP(upload_x3(L.wqkv3, w, 3L * D, D));
        UP(upload_f32(L.bqkv, bb));
        GET(p + ".self_attn.out_proj.weight", (long)D * D, t);
        UP(upload_gemm_w(L.wo, t, D, D, e->dtype, e->x2));
        if (e->x3 || e->x2_attn_f32) UP(upload_x3(L.wo3, t, D, D));
        GET(p + ".self_attn.out_proj.bias", D, t);
        UP(upload_f32(L.bo, t));
        GET(p + ".self_attn_layer_norm.weight", D, t);
        UP(upload_f32(L.ln1g, t));
        GET(p + ".self_attn_layer_norm.bias", D, t);
        UP(upload_f32(L.ln1b, t));
        GET(p + ".fc1.weight", (long)F * D, t);
        UPW(L.w1, t, F, D, 4);
        if (e->x3) UP(upload_x3(L.w13, t, F, D));
        GET(p + ".fc1.bias", F, t);
        UP(upload_f32(L.b1, t));
        GET(p + ".fc2.weight", (long)D * F, t);
        UPW(L.w2, t, D, F, 8);
        if (e->x3) UP(upload_x3(L.w23, t, D, F));
        GET(p + ".fc2.bias", D, t);
        UP(upload_f32(L.b2, t));
        GET(p + ".final_layer_norm.weight", D, t);
        UP(upload_f32(L.ln2g, t));
        GET(p + ".final_layer_norm.bias", D, t);
        UP(upload_f32(L.ln2b, t));
        if (c.rel_pos && c.gru_rel_pos) {
            GET(p + ".self_attn.grep_linear.weight", 8 * 64, t);
            UP(upload_f32(L.grep_w, t));
            GET(p + ".self_attn.grep_linear.bias", 8, t);
            UP(upload_f32(L.grep_b, t));
            GET(p + ".self_attn.grep_a", H, t);
            UP(upload_f32(L.grep_a, t));
        }
        return 0;
    };
    if (!multires) {
        e->layers.resize(c.encoder_layers);
        for (int l = 0; l < c.encoder_layers; ++l)
            if (load_layer("encoder.layers." + std::to_string(l), e->layers[l])) return 1;
    } else {
        // U-net blocks in execution order (hubert_model.py:399-507) and the conv adapters between them
        const int R = c.mr_pairs + 1, NB = 2 * R - 1, k = c.mr_kernel;
        e->mr_blocks.resize(NB);
        for (int bi = 0; bi < NB; ++bi) {
            const std::string bp = bi < R - 1 ? "encoders." + std::to_string(bi)
                                 : bi == R - 1 ? std::string("middle_encoder") : "decoders." + std::to_string(bi - R);
            BlockW& bw = e->mr_blocks[bi];
            GET(bp + ".layer_norm.weight", D, t);
            UP(upload_f32(bw.eln_g, t));
            GET(bp + ".layer_norm.bias", D, t);
            UP(upload_f32(bw.eln_b, t));
            bw.layers.resize(c.mr_layers[bi]);
            for (int l = 0; l < c.mr_layers[bi]; ++l)
                if (load_layer(bp + ".layers." + std::to_string(l), bw.layers[l])) return 1;
        }
        // A conv over all D channels becomes a GEMM whose A rows are k_eff * D contiguous elements of a zero-bordered
        // frame buffer (kernels.h, PadCopyParams):
        //   Conv1d(D, D, k, stride s, padding (k-1)/2), weight (co, ci, j):  W'[co][j*D + ci], row t starts at frame t*s - pad
        //   ConvTranspose1d(D, D, k, stride s), weight (ci, co, j): s interleaved stride-1 convs of KT = ceil(k/s) taps —
        //     out[s*q + r] = sum_m x[q - m] . w[:, :, r + s*m]  ->  W'[r*D + co][j'*D + ci] = w[ci][co][r + s*(KT-1-j')]
        //     (0 where that tap is >= k), row q starts at frame q - (KT-1); the (Q, s*D) output IS the (s*Q, D) sequence
        auto load_conv = [&](const std::string& cp, bool transposed, int stride, AdapterConvW& cw) -> int {
            std::vector<float> w, pk;
            GET(cp + ".0.weight", (long)D * D * k, w);
            long N, K;
            if (!transposed) {
                N = D;
                K = (long)k * D;
                pk.assign((size_t)N * K, 0.f);
                for (int co = 0; co < D; ++co)
                    for (int ci = 0; ci < D; ++ci)
                        for (int j = 0; j < k; ++j) pk[(long)co * K + (long)j * D + ci] = w[((long)co * D + ci) * k + j];
            } else {
                const int KT = (k + stride - 1) / stride;
                N = (long)stride * D;
                K = (long)KT * D;
                pk.assign((size_t)N * K, 0.f);
                for (int r = 0; r < stride; ++r)
                    for (int jp = 0; jp < KT; ++jp) {
                        const int tap = r + stride * (KT - 1 - jp);
                        if (tap >= k) continue;
                        for (int co = 0; co < D; ++co)
                            for (int ci = 0; ci < D; ++ci)
                                pk[((long)r * D + co) * K + (long)jp * D + ci] = w[((long)ci * D + co) * k + tap];
                    }
            }
            UP(upload_gemm_w(cw.w, pk, N, K, e->dtype, e->x2));
            if (e->x3) UP(upload_x3(cw.w3, pk, N, K));
            GET(cp + ".2.weight", D, t);
            UP(upload_f32(cw.g, t));
            GET(cp + ".2.bias", D, t);
            UP(upload_f32(cw.b, t));
            return 0;
        };
        e->mr_adapters.resize(2 * (R - 1));
        for (int i = 0; i < R - 1; ++i) {
            const int u = c.mr_ratios[2 * i], d = c.mr_ratios[2 * i + 1];
            AdapterW& dn = e->mr_adapters[i];            // downsample_modules[i]: label_rate (u, d)
            AdapterW& upm = e->mr_adapters[R - 1 + i];   // upsample_modules[i]: the inverted pair (d, u) (:474-507)
            dn.kind = c.mr_plain ? 1 : 0;
            dn.up_rate = u;
            dn.down_rate = d;
            upm.kind = c.mr_plain ? 2 : 0;
            upm.up_rate = d;
            upm.down_rate = u;
            const std::string dp = "downsample_modules." + std::to_string(i), upp = "upsample_modules." + std::to_string(i);
            if (dn.kind != 1 && load_conv(dp + ".upsample_conv", true, dn.up_rate, dn.up)) return 1;
            if (load_conv(dp + ".downsample_conv", false, dn.down_rate, dn.down)) return 1;
            if (load_conv(upp + ".upsample_conv", true, upm.up_rate, upm.up)) return 1;
            if (upm.kind != 2 && load_conv(upp + ".downsample_conv", false, upm.down_rate, upm.down)) return 1;
        }
    }
    if (c.rel_pos) {
        // bucket(rel) (wavlm/modules.py:418-446) is constant for |rel| >= max_distance, so ONE (H, 2R+1) table with
        // R = max_distance serves every sequence length: nothing is rebuilt when T changes between batches
        std::vector<float> emb, table;
        GET("encoder.layers.0.self_attn.relative_attention_bias.weight", (long)c.num_buckets * H, emb);
        e->rel_R = c.max_distance;
        build_rel_table(c, emb, e->rel_R, table);
        UP(upload_f32(e->rel_table, table));
    }
    if (c.pred_heads) {
        // output_layer = Linear(D, N*D) -> GELU -> SplitLinear(D, N, D) (distiller/model.py:155-161): SplitLinear's
        // weight is (N, Din, Dout) (module.py:66-68); each head becomes an (out, in) row-major GEMM operand
        const int NH = c.pred_heads;
        GET("output_layer.0.weight", (long)NH * D * D, t);
        UP(upload_gemm_w(e->head_w1, t, (long)NH * D, D, e->dtype, e->x2));
        if (e->x3) UP(upload_x3(e->head_w13, t, (long)NH * D, D));
        GET("output_layer.0.bias", (long)NH * D, t);
        UP(upload_f32(e->head_b1, t));
        GET("output_layer.2.weight", (long)NH * D * D, t);
        t2.resize(t.size());
        for (int k = 0; k < NH; ++k)
            for (int i = 0; i < D; ++i)
                for (int n = 0; n < D; ++n) t2[((long)k * D + n) * D + i] = t[((long)k * D + i) * D + n];
        UP(upload_gemm_w(e->head_w2, t2, (long)NH * D, D, e->dtype, e->x2));
        if (e->x3) UP(upload_x3(e->head_w23, t2, (long)NH * D, D));
        GET("output_layer.2.bias", (long)NH * D, t);
        UP(upload_f32(e->head_b2, t));
    }
#undef GET
#undef UPW
#undef UP
    for (int i = 0; i < s3enc_encoder::RING; ++i) {
        if (hipEventCreateWithFlags(&e->slot_ev[i], hipEventDisableTiming) != hipSuccess) {
            delete e;
            return fail("hipEventCreate failed");
        }
    }
    // the status word of s3enc_forward_status: device int + pinned host copy + the event behind the copy
    bool st_ok = e->status_dev.ensure(64) == hipSuccess && hipMemset(e->status_dev.p, 0, 64) == hipSuccess &&
                 hipHostMalloc((void**)&e->status_host, s3enc_encoder::STATUS_RING * sizeof(int), hipHostMallocDefault) == hipSuccess;
    for (int i = 0; st_ok && i < s3enc_encoder::STATUS_RING; ++i)
        st_ok = hipEventCreateWithFlags(&e->status_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!st_ok) {
        delete e;
        return fail("s3enc_create: status word allocation failed");
    }
    *out = e;
    return 0;
}

int s3enc_destroy(s3enc_handle h) {
    if (!h) return 0;
    DeviceGuard dg(h->device);
    (void)hipDeviceSynchronize();
    delete h;
    return 0;
}

int s3enc_num_frames(s3enc_handle h, int64_t n_samples, int32_t* T) {
    if (!h || !T) return fail("s3enc_num_frames: null argument");
    *T = (int32_t)conv_len(h->cfg, n_samples, h->cfg.n_conv);
    return 0;
}
int s3enc_num_output_frames(s3enc_handle h, int64_t n_samples, int32_t* T) {
    if (!h || !T) return fail("s3enc_num_output_frames: null argument");
    *T = (int32_t)output_frames(h->cfg, n_samples);
    return 0;
}
int s3enc_downsample_rate(s3enc_handle h, int32_t* rate) {
    if (!h || !rate) return fail("s3enc_downsample_rate: null argument");
    int r = 1;
    for (int i = 0; i < h->cfg.n_conv; ++i) r *= h->cfg.conv_stride[i];
    *rate = r;
    return 0;
}
int s3enc_valid_frames(s3enc_handle h, int64_t length, int64_t n_max, int32_t* valid) {
    if (!h || !valid) return fail("s3enc_valid_frames: null argument");
    *valid = valid_frames(h->cfg, length, n_max);
    return 0;
}

}  // extern "C"

namespace {

// Where the selected states go: straight into the caller's fp32 slab (the producing kernel writes the slot, nothing is
// copied), as 16-bit copies next to an internal fp32 residual stream, or only as their term of the Featurizer sum.
struct Sink {
    s3enc_encoder* e;
    hipStream_t st;
    int mode;  // 0 fp32 slab, 1 16-bit slab, 2 featurize
    void* out;
    long stride;
    int dt;  // the handle's compute dtype (the 16-bit slab's type)
    long M;
    int D;
    const float* w;
    int norm;
    bool first = true;

    float* slot32(int si) const { return (mode == 0 && si >= 0) ? (float*)out + (long)si * stride : nullptr; }
    void* slot16(int si) const { return (mode == 1 && si >= 0) ? (void*)((u16*)out + (long)si * stride) : nullptr; }
    bool wanted(int si) const { return si >= 0 && (mode != 2 || w[si] != 0.f); }
    // Featurizer term of state `si`, fused into a row kernel that holds the state as its input (1) or output (2)
    LnAcc acc(int si, int where) {
        LnAcc a;
        if (mode != 2 || si < 0 || w[si] == 0.f) return a;
        a.acc = (float*)out;
        a.w = w[si];
        a.mode = where;
        a.norm = norm;
        a.init = first;
        first = false;
        return a;
    }
    // a state a non-LayerNorm kernel left in fp32 outside the slab: its 16-bit copy / its Featurizer term
    hipError_t emit(int si, const float* x, bool copy16 = true) {
        if (si < 0) return hipSuccess;
        if (mode == 1 && copy16) return launch_emit_state(dt, x, M, D, slot16(si), LnAcc(), st);
        if (mode == 2 && w[si] != 0.f) return launch_emit_state(F32, x, M, D, nullptr, acc(si, 1), st);
        return hipSuccess;
    }
    hipError_t done(int si) {
        if (si < 0 || mode == 2 || e->layer_events.empty() || si >= (int)e->layer_events.size()) return hipSuccess;
        return hipEventRecord(e->layer_events[si], st);
    }
};

int forward_body(s3enc_handle e, const float* const* wav_ptrs_host, const int64_t* lengths, int32_t B, int64_t n_max_in,
                 const FwdOpts& fo, void* out, int64_t layer_stride, hipStream_t st);

// ---- forward chain: one event per device, re-recorded behind every forward; the next forward (any handle, any stream) waits for it on
// the device.  hipStreamWaitEvent captures the record that is current when it is called, so re-using one event is safe; a stream that
// waits for an event recorded on itself waits for nothing new.  Host cost: two runtime calls per forward under one mutex.
namespace {
struct ForwardChain {
    std::mutex mu;
    hipEvent_t ev[64] = {};
    bool recorded[64] = {};
};
ForwardChain& forward_chain() {
    static ForwardChain* c = new ForwardChain();  // (never destroyed: events must not outlive the runtime at process exit)
    return *c;
}
hipError_t forward_chain_wait(int dev, hipStream_t st) {
    if (dev < 0 || dev >= 64) return hipSuccess;
    ForwardChain& c = forward_chain();
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.recorded[dev]) return hipSuccess;
    return hipStreamWaitEvent(st, c.ev[dev], 0);
}
hipError_t forward_chain_record(int dev, hipStream_t st) {
    if (dev < 0 || dev >= 64) return hipSuccess;
    ForwardChain& c = forward_chain();
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.ev[dev]) {
        hipError_t e = hipEventCreateWithFlags(&c.ev[dev], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipEventRecord(c.ev[dev], st);
    if (e == hipSuccess) c.recorded[dev] = true;
    return e;
}
}  // namespace

// The forward proper runs with the handle's tuning and status word current for this thread.  The word is cleared in front of
// it and copied to the next pinned slot behind it, with an event, so that s3enc_forward_status never has to touch the device.
int forward_impl(s3enc_handle e, const float* const* wav_ptrs_host, const int64_t* lengths, int32_t B, int64_t n_max_in,
                 const FwdOpts& fo, void* out, int64_t layer_stride, hipStream_t st) {
    TuningScope tuning_scope(e->has_tuning ? &e->tun : nullptr);
    StatusScope status_scope((int*)e->status_dev.p);
    // forward chain (tuning key forward_chain, kernels.h): this forward starts behind the previous forward of any handle on this device
    // (exact fp32 — the tile kernel waits vmcnt(0) for every buffer_load ... lds — stayed bit-stable in every concurrent run and is left free to overlap)
    const bool chain = tuning().forward_chain != 0 && (e->dtype != F32 || e->x3);
    if (chain) HIP_TRY(forward_chain_wait(e->device, st));
    HIP_TRY(hipMemsetAsync(e->status_dev.p, 0, sizeof(int), st));
    const int rc = forward_body(e, wav_ptrs_host, lengths, B, n_max_in, fo, out, layer_stride, st);
    if (chain) HIP_TRY(forward_chain_record(e->device, st));  // (also behind a failed forward: whatever it enqueued is in the stream)
    if (rc == 0) {
        const int slot = e->status_next;
        if (e->status_busy[slot]) {  // STATUS_RING forwards ago: finished long since unless the host ran far ahead
            HIP_TRY(hipEventSynchronize(e->status_ev[slot]));
            e->status_sticky |= ((volatile int*)e->status_host)[slot];
            e->status_busy[slot] = false;
        }
        HIP_TRY(hipMemcpyAsync(e->status_host + slot, e->status_dev.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(e->status_ev[slot], st));
        e->status_busy[slot] = true;
        e->status_next = (slot + 1) % s3enc_encoder::STATUS_RING;
    }
    return rc;
}

int forward_body(s3enc_handle e, const float* const* wav_ptrs_host, const int64_t* lengths, int32_t B, int64_t n_max_in,
                 const FwdOpts& fo, void* out, int64_t layer_stride, hipStream_t st) {
    const s3enc_config& c = e->cfg;
    const int C = c.conv_dim, D = c.embed_dim, F = c.ffn_dim, H = c.heads, NL = c.encoder_layers;
    const int dt = e->dtype, es = e->es;
    static const int dbg_stop = getenv("S3ENC_DEBUG_STOP") ? atoi(getenv("S3ENC_DEBUG_STOP")) : 0;
    const bool dist = c.family == S3ENC_DISTILLER;
    const bool mr = c.family == S3ENC_MULTIRES;
    const int NH = dist ? c.pred_heads : 0;
    if (B <= 0) return fail("s3enc_forward: B must be positive");
    if (fo.selection < 0 || fo.selection > 2) return fail("s3enc_forward: unknown selection");
    if ((dist || mr) && fo.selection != S3ENC_SEL_HIDDEN)
        return fail("s3enc_forward: DistilHuBERT / multires-HuBERT have one selection (their hidden_states list)");
    const int NS = num_states(c, fo.selection);
    if (fo.featurize && !fo.w) return fail("s3enc_forward: featurize needs feat_w");
    if (!fo.featurize && fo.out_dtype != F32 && (fo.out_dtype != dt || dt == F32))
        return fail("s3enc_forward: out_dtype must be S3ENC_F32 or the handle's own 16-bit compute dtype");
    long n_max = 0;
    for (int b = 0; b < B; ++b) {
        if (lengths[b] <= 0) return fail("s3enc_forward: empty utterance");
        if (lengths[b] > n_max) n_max = lengths[b];
        if (!wav_ptrs_host[b]) return fail("s3enc_forward: null waveform pointer");
    }
    if (n_max_in > 0) {
        if (n_max_in < n_max) return fail("s3enc_forward: n_max is smaller than the longest utterance");
        n_max = n_max_in;
    }
    std::vector<long> L(c.n_conv);
    for (int i = 0; i < c.n_conv; ++i) L[i] = conv_len(c, n_max, i + 1);
    const long T = L[c.n_conv - 1];
    if (T < 1) return fail("s3enc_forward: input shorter than the receptive field of the conv stack");
    const long M = (long)B * T;
    if (!out) return fail("s3enc_forward: null output");
    MrPlan plan;  // multires-HuBERT: the frame geometry of the U-net; the states are (B, plan.T_out, D)
    if (mr) mr_plan(c, T, plan);
    for (const auto& bp : plan.blocks)
        if (bp.T < 1 || plan.T_out < 1) return fail("s3enc_forward: input too short for the coarsest resolution of the U-net");
    if (!fo.featurize) {
        if (layer_stride < (mr ? (long)B * plan.T_out : M) * D) return fail("s3enc_forward: layer_stride < B*T*D");
        if (layer_stride & 3) return fail("s3enc_forward: layer_stride must be a multiple of 4 elements (vector stores)");
    }
    if ((uintptr_t)out & 15) return fail("s3enc_forward: out must be 16-byte aligned");
    if (c.rel_pos && T > 6000) return fail("s3enc_forward: WavLM relative-position window limited to 6000 frames (120 s) per batch");
    std::vector<int> valid(B);
    for (int b = 0; b < B; ++b) {
        valid[b] = valid_frames(c, lengths[b], n_max);
        if (valid[b] < 1) return fail("s3enc_forward: an utterance is too short to produce a valid frame");
    }
    // multires-HuBERT: un-masked frames per block — the padding mask follows every adapter as
    // repeat_interleave(up)[::down][:T'] (hubert_model.py:1080-1084,1169-1172,1256-1258) and align_size_sum's cut (:777-783)
    std::vector<int> valid_blk;  // blocks 1.., B entries each (block 0 uses `valid`)
    if (mr) {
        std::vector<int> v = valid;
        for (size_t bi = 1; bi < plan.blocks.size(); ++bi) {
            const MrBlockPlan& bp = plan.blocks[bi];
            const AdapterW& aw = e->mr_adapters[bp.adapter];
            const long up = aw.kind == 1 ? 1 : aw.up_rate, down = aw.kind == 2 ? 1 : aw.down_rate;
            for (int b = 0; b < B; ++b) {
                long nv = ((long)v[b] * up + down - 1) / down;
                v[b] = (int)std::min(nv, bp.T);
            }
            valid_blk.insert(valid_blk.end(), v.begin(), v.end());
            for (int b = 0; b < B; ++b) v[b] = (int)std::min<long>(v[b], bp.T_sum);
        }
    }
    DeviceGuard dg(e->device);
    if (!dg.ok) return fail("s3enc_forward: hipSetDevice failed");

    // ---- small device state: tables + stats ----
    const size_t tbl_bytes = (size_t)B * (8 + 8 + 4) + valid_blk.size() * 4;
    const size_t part_elems = stats_partial_elems(B, n_max);
    {
        Bump sb(nullptr);
        sb.take(tbl_bytes);
        sb.take((size_t)B * sizeof(float2));
        sb.take((size_t)B * C * sizeof(float2));
        sb.take(part_elems * 8);
        HIP_TRY(e->small.ensure_on_stream(sb.off + 1024, st));
    }
    Bump sb(e->small.p);
    char* d_tbl = (char*)sb.take(tbl_bytes);
    float2* d_norm = (float2*)sb.take((size_t)B * sizeof(float2));
    float2* d_gn = (float2*)sb.take((size_t)B * C * sizeof(float2));
    double* d_part = (double*)sb.take(part_elems * 8);
    const float* const* d_ptrs = (const float* const*)d_tbl;
    const long* d_lens = (const long*)(d_tbl + (size_t)B * 8);
    const int* d_valid = (const int*)(d_tbl + (size_t)B * 16);

    // pinned staging ring (so the async H2D copy never reads freed / overwritten host memory)
    if (tbl_bytes > e->slot_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (e->pinned) HIP_TRY(hipHostFree(e->pinned));
        e->pinned = nullptr;
        e->slot_bytes = tbl_bytes * 4 + 4096;
        HIP_TRY(hipHostMalloc(&e->pinned, e->slot_bytes * s3enc_encoder::RING, hipHostMallocDefault));
    }
    {
        const int slot = e->slot_next;
        e->slot_next = (slot + 1) % s3enc_encoder::RING;
        HIP_TRY(hipEventSynchronize(e->slot_ev[slot]));
        char* hp = (char*)e->pinned + (size_t)slot * e->slot_bytes;
        memcpy(hp, wav_ptrs_host, (size_t)B * 8);
        for (int b = 0; b < B; ++b) ((long*)(hp + (size_t)B * 8))[b] = (long)lengths[b];
        memcpy(hp + (size_t)B * 16, valid.data(), (size_t)B * 4);
        if (!valid_blk.empty()) memcpy(hp + (size_t)B * 20, valid_blk.data(), valid_blk.size() * 4);
        HIP_TRY(hipMemcpyAsync(d_tbl, hp, tbl_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(e->slot_ev[slot], st));
    }

    // ---- workspace ----
    const bool lnmode = c.extractor_layer_norm != 0;
    const bool prel = c.layer_norm_first != 0;
    const bool gated = c.rel_pos && c.gru_rel_pos;
    const bool featln = !c.no_feature_layer_norm;
    const bool ffn_tap = fo.selection == S3ENC_SEL_FFN_OUT;
    const long HW = std::max<long>(F, (long)NH * D);  // widest row of the FFN / prediction-head intermediate
    void *actA, *actB, *tmp32, *feat32, *featT, *x32, *xpc, *xT, *qkv, *attn, *tmp1, *tmp2, *hbuf, *gate, *ffnbuf, *lnst;
    for (int pass = 0; pass < 2; ++pass) {
        Bump wb(pass ? e->ws.p : nullptr);
        // conv0's output in the compute dtype; in the fp16x2 hybrid conv2, conv4, ... write fp32 rows back here: L[2] <= L[0] / 2
        // only when conv1 * conv2 stride >= 2, which no config validation promises
        actA = wb.take(std::max((size_t)B * L[0] * C * (e->x2_conv_f32_from == 1 ? 4 : es),
                                (e->x2_conv_f32_from && c.n_conv > 2) ? (size_t)B * L[2] * C * 4 : (size_t)0));
        actB = wb.take((size_t)B * L[1] * C * (e->x2_conv_f32_from ? 4 : es));  // (fp32 activations between the later convs)
        tmp32 = lnmode ? wb.take((size_t)B * L[1] * C * 4) : nullptr;
        feat32 = featln ? wb.take((size_t)M * C * 4) : nullptr;
        featT = wb.take((size_t)M * C * (e->x2_proj_f32 ? 4 : es));
        x32 = wb.take((size_t)M * D * 4);
        xpc = wb.take((size_t)M * D * 4);
        xT = wb.take((size_t)M * D * es);
        qkv = wb.take((size_t)M * 3 * D * es);
        attn = wb.take((size_t)M * D * (e->x2_attn_f32 ? 4 : es));
        tmp1 = wb.take((size_t)M * D * 4);
        tmp2 = wb.take((size_t)M * D * 4);
        lnst = wb.take((size_t)M * sizeof(float2));  // (mean, rstd) per row: LayerNorm 1 -> fc2's epilogue (ln1_fold)
        hbuf = wb.take((size_t)M * HW * es);
        gate = gated ? wb.take((size_t)B * H * T * 4) : nullptr;
        ffnbuf = (ffn_tap && (fo.featurize || fo.out_dtype != F32)) ? wb.take((size_t)M * D * 4) : nullptr;
        if (!pass) HIP_TRY(e->ws.ensure_on_stream(wb.off + 4096, st));
        // (diagnostic, profiles/r06d_concurrent_forwards_exclusions.md: S3ENC_DEBUG_POISON=1 fills the workspace with NaN patterns in front of
        //  every forward — every buffer is written before it is read, so a correct forward does not change; a read that overtakes its
        //  producer shows as NaN rows instead of plausible numbers)
        static const int dbg_poison = getenv("S3ENC_DEBUG_POISON") ? atoi(getenv("S3ENC_DEBUG_POISON")) : 0;
        if (pass && dbg_poison) HIP_TRY(hipMemsetAsync(e->ws.p, 0xFF, wb.off, st));
    }
    e->taps.clear();

    Sink sink{e, st, fo.featurize ? 2 : (fo.out_dtype != F32 ? 1 : 0), out, (long)layer_stride, dt, M, D, fo.w, fo.feat_norm};
    if (fo.featurize) {
        bool any = false;
        for (int i = 0; i < NS; ++i) any = any || fo.w[i] != 0.f;
        if (!any) HIP_TRY(hipMemsetAsync(out, 0, (size_t)(mr ? (long)B * plan.T_out : M) * D * 4, st));
    }
    // state index of each tensor of the forward under this selection (-1: not a state)
    const bool hid = fo.selection == S3ENC_SEL_HIDDEN && !dist;
    const bool lay = fo.selection == S3ENC_SEL_LAYER_OUT;
    auto si_hidden = [&](int l) { return hid ? l : -1; };                             // input of layer l / encoder output
    auto si_layer_out = [&](int l) { return lay ? l : (dist ? 1 + l : -1); };         // output of layer l
    auto si_stream = [&](int l) {  // the tensor "residual stream after layer l" as the next layer's input
        const int a = (l + 1 < NL || !prel) ? si_hidden(l + 1) : -1;  // pre-LN: the encoder output is the normed stream
        return a >= 0 ? a : si_layer_out(l);
    };
    auto other = [&](const float* busy) { return busy == (const float*)x32 ? (float*)xpc : (float*)x32; };

    WavTable wt{d_ptrs, d_lens, B, n_max};
    // (diagnostic: S3ENC_DEBUG_KEEP=1 copies conv0 and the early conv outputs aside behind their kernel, so that a FULL forward keeps them as
    //  taps — the workspace reuses their buffers; the copies are never freed: a debugging process)
    static const int dbg_keep = getenv("S3ENC_DEBUG_KEEP") ? atoi(getenv("S3ENC_DEBUG_KEEP")) : 0;
    auto keep_tap = [&](const char* name, int idx, const void* src, long elems, int tdt) -> hipError_t {
        if (!dbg_keep) return hipSuccess;
        static std::map<std::pair<const void*, int>, std::pair<void*, size_t>> kept;
        static std::mutex kept_mu;
        const size_t bytes = (size_t)elems * (tdt == F32 ? 4 : 2);
        void* kp = nullptr;
        {
            std::lock_guard<std::mutex> lk(kept_mu);
            auto& slot = kept[{(const void*)e, idx}];
            if (slot.second < bytes) {
                hipError_t er = hipMalloc(&slot.first, bytes);
                if (er != hipSuccess) return er;
                slot.second = bytes;
            }
            kp = slot.first;
        }
        hipError_t er = hipMemcpyAsync(kp, src, bytes, hipMemcpyDeviceToDevice, st);
        if (er != hipSuccess) return er;
        e->taps[name] = {kp, elems, tdt};
        return hipSuccess;
    };
    {
        Prof pr(e, st, "wav_stats", 0, 4.0 * B * n_max);
        HIP_TRY(launch_wav_norm_stats(wt, c.normalize, d_part, d_norm, st, c.wav_norm_eps));
    }
    if (!lnmode) {
        Prof pr(e, st, "gn_stats", 0, 4.0 * B * n_max);
        HIP_TRY(launch_gn_stats(wt, d_norm, (const float*)e->conv[0].w.p, (const float*)e->gn_g.p, (const float*)e->gn_b.p, C,
                                c.conv_kernel[0], c.conv_stride[0], L[0], d_part, nullptr, d_gn, st));
    }
    {
        Conv0Params p{};
        p.wav = wt;
        p.norm = d_norm;
        p.w0 = (const float*)e->conv[0].w.p;
        p.bias = e->conv[0].has_bias ? (const float*)e->conv[0].bias.p : nullptr;
        p.gn = lnmode ? nullptr : d_gn;
        p.ln_g = lnmode ? (const float*)e->conv[0].lng.p : nullptr;
        p.ln_b = lnmode ? (const float*)e->conv[0].lnb.p : nullptr;
        p.C = C;
        p.k0 = c.conv_kernel[0];
        p.s0 = c.conv_stride[0];
        p.L0 = L[0];
        p.out = actA;
        const bool c0f32 = e->x2_conv_f32_from == 1;  // (fp16x2_conv1_f32: conv1 reads fp32 rows)
        p.fast = e->x3 || c0f32;
        p.nt = tuning().conv0_nt;
        Prof pr(e, st, "conv0", 2.0 * B * L[0] * C * p.k0, 4.0 * B * n_max + (double)B * L[0] * C * (c0f32 ? 4 : es));
        HIP_TRY(launch_conv0(c0f32 ? (int)F32 : dt, p, st));
        // (diagnostic, tools/two_stream_probe.py --taps: S3ENC_DEBUG_STOP = k ends the forward behind conv(k - 1) and keeps its output as a tap)
        if (dbg_stop) e->taps["conv0"] = {actA, (long)B * L[0] * C, c0f32 ? (int)F32 : dt};
        else HIP_TRY(keep_tap("conv0", 0, actA, (long)B * L[0] * C, c0f32 ? (int)F32 : dt));
        if (dbg_stop == 1) return 0;
    }
    // conv1..: implicit GEMM on channel-last activations.  The last one feeds LayerNorm(C) in fp32, or — without that
    // norm (DistilHuBERT) — post_extract_proj directly, in the compute dtype.
    void* cur = actA;
    for (int i = 1; i < c.n_conv; ++i) {
        const bool last = i == c.n_conv - 1;
        // S3ENC_F16X2, GroupNorm extractor: conv `x2_conv_f32_from`.. read fp32 activations through the three-term GEMM
        const bool x3in = e->x2_conv_f32_from && i >= e->x2_conv_f32_from;
        const bool x3next = e->x2_conv_f32_from && i + 1 >= e->x2_conv_f32_from && !last;
        const bool f32out = dt == F32 || (last && featln) || x3next;
        void* dst = last ? (featln ? feat32 : featT) : (cur == actA ? actB : actA);
        GemmParams g{};
        g.A = cur;
        g.lda = (long)c.conv_stride[i] * C;
        g.a_bs = L[i - 1] * C;
        g.W = e->conv[i].w.p;
        g.W_x3 = e->conv[i].w3.p;
        g.bias = e->conv[i].has_bias ? (const float*)e->conv[i].bias.p : nullptr;
        g.M = (int)L[i];
        g.N = C;
        g.K = c.conv_kernel[i] * C;
        g.batches = B;
        g.ldo = C;
        g.o_bs = L[i] * C;
        const double fl = 2.0 * B * L[i] * C * g.K;
        const double by = ((double)B * L[i - 1] * C + (double)C * g.K) * (x3in ? 4 : es) + (double)B * L[i] * C * (f32out ? 4 : es);
        char kind[32];
        snprintf(kind, sizeof(kind), "gemm:conv%d", i);
        if (!lnmode) {
            g.act = 1;
            if (f32out) g.out32 = (float*)dst; else g.out16 = dst;
            Prof pr(e, st, kind, fl, by);
            if (x3in) {
                if (!gemm_x3_eligible(g)) return fail("conv layer is not a shape of the three-term GEMM (internal)");
                HIP_TRY(launch_gemm(F32, g, st));
            } else {
                HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            }
        } else {
            g.act = 0;
            g.out32 = (float*)tmp32;
            {
                Prof pr(e, st, kind, fl, by);
                if (x3in) {  // fp32 LayerNorm output of the previous conv -> three-term GEMM (S3ENC_F16X2, see x2_conv_f32_from)
                    if (!gemm_x3_eligible(g)) return fail("conv layer is not a shape of the three-term GEMM (internal)");
                    HIP_TRY(launch_gemm(F32, g, st));
                } else {
                    HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
                }
            }
            Prof pr(e, st, "layernorm:conv", 0, (double)B * L[i] * C * (4 + (f32out ? 4 : es)));
            HIP_TRY(launch_layernorm(dt, (const float*)tmp32, (const float*)e->conv[i].lng.p, (const float*)e->conv[i].lnb.p,
                                     (long)B * L[i], C, e->x3 ? 2 : 1, f32out ? (float*)dst : nullptr, f32out ? nullptr : dst, st));
        }
        char tn[16];
        snprintf(tn, sizeof(tn), "conv%d", i);
        if (i >= c.n_conv - 3 || dbg_stop) e->taps[tn] = {dst, (long)B * L[i] * C, f32out ? (int)F32 : dt};  // earlier ones get overwritten
        else HIP_TRY(keep_tap(tn, i, dst, (long)B * L[i] * C, f32out ? (int)F32 : dt));
        if (dbg_stop == 1 + i) return 0;
        cur = dst;
    }
    // LayerNorm(C) -> post_extract_proj (+ zero padded frames)
    const bool proj32 = dt == F32 || (e->x2_proj_f32 && featln);  // the projection's operand stays fp32
    if (featln) {
        Prof pr(e, st, "layernorm:feat", 0, (double)M * C * (4 + (proj32 ? 4 : es)));
        HIP_TRY(launch_layernorm(dt, (const float*)feat32, (const float*)e->fln_g.p, (const float*)e->fln_b.p, M, C, 0,
                                 proj32 ? (float*)featT : nullptr, proj32 ? nullptr : featT, st));
        e->taps["feat_ln"] = {featT, M * C, proj32 ? (int)F32 : dt};
        if (dbg_stop == 20) return 0;
    }
    const int si_proj = dist ? 0 : -1;  // DistilHuBERT: hidden_states[0] = feat_final, padded frames zeroed in place
    float* xproj = sink.slot32(si_proj) ? sink.slot32(si_proj) : (float*)x32;
    {
        GemmParams g{};
        g.A = featT;
        g.lda = C;
        g.a_bs = T * C;
        g.W = e->proj_w.p;
        g.W_x3 = e->proj_w3.p;
        g.bias = (const float*)e->proj_b.p;
        g.M = (int)T;
        g.N = D;
        g.K = C;
        g.batches = B;
        g.row_limit = d_valid;
        g.out32 = xproj;
        g.out16 = sink.slot16(si_proj);
        g.ldo = D;
        g.o_bs = T * D;
        {
            Prof pr(e, st, "gemm:proj", 2.0 * M * D * C, ((double)M * C + (double)D * C) * es + (double)M * D * 4);
            if (proj32 && dt != F32) {
                if (!gemm_x3_eligible(g)) return fail("post_extract_proj is not a shape of the three-term GEMM (internal)");
                HIP_TRY(launch_gemm(F32, g, st));
            } else {
                HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            }
        }
        e->taps["proj"] = {xproj, M * D, F32};
        if (dbg_stop == 21) return 0;
        HIP_TRY(sink.emit(si_proj, xproj, false));
        HIP_TRY(sink.done(si_proj));
    }
    if (mr) {
        std::vector<const int*> dv(plan.blocks.size(), d_valid);
        for (size_t bi = 1; bi < plan.blocks.size(); ++bi) dv[bi] = (const int*)(d_tbl + (size_t)B * 20) + (bi - 1) * (size_t)B;
        return multires_tail(e, st, B, plan, dv, xproj, out, (long)layer_stride, fo) ? 1 : 0;
    }
    // positional conv + residual; hidden_states[0]
    float* x_cur;          // the fp32 residual stream entering the layer loop
    void* a16_cur = nullptr;  // post-LN 16-bit modes: its 16-bit copy (the q|k|v operand)
    {
        const int si0 = si_hidden(0);
        float* pc_out = prel ? (sink.slot32(si0) ? sink.slot32(si0) : other(xproj)) : other(xproj);
        PosConvParams p{};
        p.x = xproj;
        p.w = e->pos_w.p;
        p.bias = (const float*)e->pos_b.p;
        p.out = pc_out;
        p.B = B;
        p.T = (int)T;
        p.D = D;
        p.G = c.conv_pos_groups;
        p.K = e->pos_k;  // conv_pos, or conv_pos + zero taps in the 16-bit / fp32x3 modes
        p.pad = e->pos_pad;
        auto run_conv = [&](PosConvParams& q) -> hipError_t {
            if (e->x3) return launch_posconv16(3, q, st);
            return dt == F32 ? launch_posconv(q, st) : launch_posconv16(dt, q, st);
        };
        if (c.pos_conv_depth > 1) {
            // data2vec: x + block_n(...block_1(x)), block = conv -> LayerNorm(no affine) -> GELU (wav2vec2_model.py:2999-3017)
            const float* cur_in = xproj;
            const int gelu_act = (dt != F32 || e->x3) ? 2 : 1;
            for (int i = 0; i < c.pos_conv_depth; ++i) {
                p.x = cur_in;
                p.w = e->pos_ws[i].p;
                p.bias = (const float*)e->pos_bs[i].p;
                p.out = (float*)tmp1;
                p.K = e->pos_k;
                p.pad = e->pos_pad;
                p.plain = 1;
                {
                    Prof pr(e, st, "posconv", 2.0 * M * D * (D / p.G) * (2 * e->pos_pad + 1), (double)M * D * 8);
                    HIP_TRY(run_conv(p));
                }
                Prof pr(e, st, "layernorm:posconv", 0, (double)M * D * 8);
                HIP_TRY(launch_layernorm(F32, (const float*)tmp1, (const float*)e->ones.p, (const float*)e->zeros.p, M, D, gelu_act,
                                         (float*)tmp2, nullptr, st));
                cur_in = (const float*)tmp2;
            }
            Prof pr(e, st, "residual_add", 0, (double)M * D * 12);
            HIP_TRY(launch_add(xproj, (const float*)tmp2, pc_out, M * D, st));
            p.out = pc_out;
        } else {
            Prof pr(e, st, "posconv", 2.0 * M * D * (D / p.G) * c.conv_pos, (double)M * D * 8 + (double)D * (D / p.G) * c.conv_pos * 4);
            if (e->x3) p.w = e->pos_w3.p;
            HIP_TRY(run_conv(p));
        }
        e->taps["posconv"] = {p.out, M * D, F32};
        if (dbg_stop == 22) return 0;
        if (prel) {
            x_cur = pc_out;
            if (sink.mode == 1) HIP_TRY(sink.emit(si0, pc_out));  // featurize: ln1 of layer 0 adds this state's term
            HIP_TRY(sink.done(si0));
        } else {
            float* h0 = sink.slot32(si0) ? sink.slot32(si0) : other(pc_out);
            void* h16 = dt == F32 ? nullptr : (sink.slot16(si0) ? sink.slot16(si0) : xT);
            Prof pr(e, st, "layernorm:enc", 0, (double)M * D * (8 + (dt == F32 ? 0 : es)));
            LnGate g0;  // post-LN WavLM: the gate of layer 0 reads this LayerNorm's output
            if (gated) {
                g0.gw = (const float*)e->layers[0].grep_w.p;
                g0.gb = (const float*)e->layers[0].grep_b.p;
                g0.ga = (const float*)e->layers[0].grep_a.p;
                g0.gate = (float*)gate;
                g0.T = (int)T;
                g0.H = H;
            }
            HIP_TRY(launch_layernorm(dt, pc_out, (const float*)e->eln_g.p, (const float*)e->eln_b.p, M, D, 0, h0, h16, st,
                                     sink.acc(si0, 2), g0));
            x_cur = h0;
            a16_cur = h16;
            HIP_TRY(sink.done(si0));
        }
    }
    const float* d_table = c.rel_pos ? (const float*)e->rel_table.p : nullptr;
    // WavLM gate of layer l's attention, computed by the LayerNorm that produces that attention's input
    auto gate_of = [&](int l) {
        LnGate g;
        if (gated && l < NL) {
            const LayerW& w = e->layers[l];
            g.gw = (const float*)w.grep_w.p;
            g.gb = (const float*)w.grep_b.p;
            g.ga = (const float*)w.grep_a.p;
            g.gate = (float*)gate;
            g.T = (int)T;
            g.H = H;
        }
        return g;
    };

    const double gM = (double)M;
    int si_cur = si_hidden(0);  // state index of x_cur (pre-LN featurize: its term is added by the LayerNorm that reads it)
    for (int l = 0; l < NL; ++l) {
        LayerW& Lw = e->layers[l];
        if (dbg_stop == 40 + l) return 0;  // (before layer l)
        const bool lastl = l == NL - 1;
        const void* a_in;       // operand of the q|k|v GEMM
        const float* gate_src;  // WavLM: the attention module's input
        if (prel) {
            Prof pr(e, st, "layernorm:ln1", 0, gM * D * (4 + es));
            // WavLM: the gate reads LN1's output — computed inside this pass (no fp32 copy, no separate kernel)
            HIP_TRY(launch_layernorm(dt, x_cur, (const float*)Lw.ln1g.p, (const float*)Lw.ln1b.p, M, D, 0,
                                     dt == F32 ? (float*)xT : nullptr, dt == F32 ? nullptr : xT, st, sink.acc(si_cur, 1),
                                     gate_of(l)));
            a_in = xT;
            gate_src = nullptr;
        } else {
            a_in = dt == F32 ? (const void*)x_cur : (const void*)a16_cur;
            gate_src = x_cur;
        }
        (void)gate_src;  // the gate of this layer was written by the LayerNorm that produced its attention input
        {
            GemmParams g{};
            g.A = a_in;
            g.lda = D;
            g.W = Lw.wqkv.p;
            g.W_x3 = Lw.wqkv3.p;
            g.bias = (const float*)Lw.bqkv.p;
            g.M = (int)M;
            g.N = 3 * D;
            g.K = D;
            g.batches = 1;
            g.ldo = 3 * D;
            if (dt == F32) g.out32 = (float*)qkv; else g.out16 = qkv;
            Prof pr(e, st, "gemm:qkv", 2.0 * gM * 3 * D * D, (gM * D + 3.0 * D * D + gM * 3 * D) * es);
            HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            if (l == 0) e->taps["qkv0"] = {qkv, M * 3 * D, dt};
            if (dbg_stop == 30 && l == 0) return 0;
        }
        {
            AttnParams a{};
            a.qkv = qkv;
            a.out = attn;
            a.valid = d_valid;
            a.B = B;
            a.T = (int)T;
            a.H = H;
            a.bias_table = d_table;
            a.table_R = e->rel_R;
            a.gate = gated ? (const float*)gate : nullptr;
            a.out_f32 = e->x2_attn_f32 ? 1 : 0;
            Prof pr(e, st, "attention", 4.0 * B * H * (double)T * T * 64, gM * 4 * D * es);
            HIP_TRY(launch_attention(e->x3 ? 3 : dt, a, st));
            if (l == 0) e->taps["attn0"] = {attn, M * D, e->x2_attn_f32 ? (int)F32 : dt};
            if (dbg_stop == 31 && l == 0) return 0;
        }
        {   // out_proj + bias + residual
            GemmParams g{};
            g.A = attn;
            g.lda = D;
            g.W = Lw.wo.p;
            g.W_x3 = Lw.wo3.p;
            g.bias = (const float*)Lw.bo.p;
            g.M = (int)M;
            g.N = D;
            g.K = D;
            g.batches = 1;
            g.ldo = D;
            g.residual = x_cur;
            g.out32 = (float*)tmp1;
            Prof pr(e, st, "gemm:out_proj", 2.0 * gM * D * D, (gM * D + (double)D * D) * es + gM * D * 8);
            if (e->x2_attn_f32) {
                if (!gemm_x3_eligible(g)) return fail("out_proj is not a shape of the three-term GEMM (internal)");
                HIP_TRY(launch_gemm(F32, g, st));
            } else {
                HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            }
        }
        const float* ffn_res;
        const void* ffn_in;
        if (dbg_stop == 32 && l == 0) return 0;  // (behind out_proj)
        // round 6 (second session): in the 16-bit modes a post-LN layer's LayerNorm 1 need not write its fp32 output — that tensor is only
        // fc2's residual, and fc2's epilogue can rebuild it from the row it reads anyway and two numbers per row (GemmParams::res_ln_*;
        // -49 MB of stores per layer at the reference batch).  Decided from fc2's shape class (gemm16_res_ln_ok), never from M
        bool ln1_fold = false;
        if (!prel && dt != F32 && !ffn_tap && tuning().ln1_fold) {
            GemmParams q{};
            q.A = hbuf;
            q.lda = F;
            q.W = Lw.w2.p;
            q.W_x3 = Lw.w23.p;
            q.bias = (const float*)Lw.b2.p;
            q.M = (int)M;
            q.N = D;
            q.K = F;
            q.batches = 1;
            q.ldo = D;
            q.residual = (const float*)tmp1;
            q.out32 = (float*)tmp1;
            ln1_fold = gemm16_res_ln_ok(dt, wsplit_of(e, q));
        }
        if (prel) {  // b = LN2(y) feeds fc1; residual is y itself
            Prof pr(e, st, "layernorm:ln2", 0, gM * D * (4 + es));
            HIP_TRY(launch_layernorm(dt, (const float*)tmp1, (const float*)Lw.ln2g.p, (const float*)Lw.ln2b.p, M, D, 0,
                                     dt == F32 ? (float*)xT : nullptr, dt == F32 ? nullptr : xT, st));
            ffn_res = (const float*)tmp1;
            ffn_in = xT;
        } else if (ln1_fold) {  // 16-bit modes: LN1 writes the fc1 operand and the rows' (mean, rstd); fc2's epilogue rebuilds x1 = LN1(y)
            Prof pr(e, st, "layernorm:ln1", 0, gM * D * (4 + es));
            HIP_TRY(launch_layernorm(dt, (const float*)tmp1, (const float*)Lw.ln1g.p, (const float*)Lw.ln1b.p, M, D, 0,
                                     nullptr, xT, st, LnAcc(), LnGate(), (float2*)lnst));
            ffn_res = (const float*)tmp1;  // (y itself: fc2 normalises the rows it adds, then overwrites them in place)
            ffn_in = xT;
        } else {  // x1 = LN1(y): both the fc1 operand and the FFN residual
            Prof pr(e, st, "layernorm:ln1", 0, gM * D * (8 + (dt == F32 ? 0 : es)));
            // (ws_inplace: LayerNorm 1 overwrites its input — a wave reads its whole row before it writes — and fc2 then adds its
            //  product onto that buffer in place: every element is read and written by the same lane)
            float* x1 = tuning().ws_inplace ? (float*)tmp1 : (float*)tmp2;
            HIP_TRY(launch_layernorm(dt, (const float*)tmp1, (const float*)Lw.ln1g.p, (const float*)Lw.ln1b.p, M, D, 0,
                                     x1, dt == F32 ? nullptr : xT, st));
            ffn_res = (const float*)x1;
            ffn_in = dt == F32 ? (const void*)x1 : (const void*)xT;
        }
        {   // fc1 + bias + GELU
        if (dbg_stop == 33 && l == 0) return 0;  // (behind LayerNorm 1 / 2)
            GemmParams g{};
            g.A = ffn_in;
            g.lda = D;
            g.W = Lw.w1.p;
            g.W_x3 = Lw.w13.p;
            g.bias = (const float*)Lw.b1.p;
            g.M = (int)M;
            g.N = F;
            g.K = D;
            g.batches = 1;
            g.act = 1;
            g.ldo = F;
            if (dt == F32) g.out32 = (float*)hbuf; else g.out16 = hbuf;
            Prof pr(e, st, "gemm:fc1", 2.0 * gM * F * D, (gM * D + (double)F * D + gM * F) * es);
            HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
        }
        if (dbg_stop == 34 && l == 0) return 0;  // (behind fc1)
        // fc2 + bias + residual.  pre-LN: the result IS the residual stream after the layer (a state for l < NL-1, and
        // for the fairseq_layers / DistilHuBERT selections); post-LN: it feeds final_layer_norm.
        const int si_next = si_stream(l);
        // (the last pre-LN layer's raw stream goes to tmp2 when it is not a state, which keeps the "proj" debug tap in x32
        // alive through a default forward)
        float* x_next = sink.slot32(si_next) ? sink.slot32(si_next) : ((prel && lastl) ? (float*)tmp2 : other(x_cur));
        float* fc2_dst = prel ? x_next : (float*)tmp1;
        {
            GemmParams g{};
            g.A = hbuf;
            g.lda = F;
            g.W = Lw.w2.p;
            g.W_x3 = Lw.w23.p;
            g.bias = (const float*)Lw.b2.p;
            g.M = (int)M;
            g.N = D;
            g.K = F;
            g.batches = 1;
            g.ldo = D;
            if (ffn_tap) {
                // "fairseq_layers_before_residual": the GEMM exports fc2(x) + bias, the residual is re-applied by an
                // elementwise add in the same order as the fused epilogue (bit-identical stream)
                float* f_out = sink.slot32(l) ? sink.slot32(l) : (float*)ffnbuf;
                g.out32 = f_out;
                g.out16 = sink.slot16(l);
                {
                    Prof pr(e, st, "gemm:fc2", 2.0 * gM * D * F, (gM * F + (double)D * F) * es + gM * D * 4);
                    HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
                }
                HIP_TRY(sink.emit(l, f_out, false));
                HIP_TRY(sink.done(l));
                Prof pr(e, st, "residual_add", 0, gM * D * 12);
                HIP_TRY(launch_add(f_out, ffn_res, fc2_dst, M * D, st));
            } else {
                g.residual = ffn_res;
                g.out32 = fc2_dst;
                g.out16 = prel ? sink.slot16(si_next) : nullptr;
                if (ln1_fold) {
                    g.res_ln_stats = (const float2*)lnst;
                    g.res_ln_g = (const float*)Lw.ln1g.p;
                    g.res_ln_b = (const float*)Lw.ln1b.p;
                }
                Prof pr(e, st, "gemm:fc2", 2.0 * gM * D * F, (gM * F + (double)D * F) * es + gM * D * 8);
                HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            }
        }
        if (!prel) {
            void* h16 = dt == F32 ? nullptr : (sink.slot16(si_next) ? sink.slot16(si_next) : xT);
            Prof pr(e, st, "layernorm:ln2", 0, gM * D * (8 + (dt == F32 ? 0 : es)));
            HIP_TRY(launch_layernorm(dt, (const float*)tmp1, (const float*)Lw.ln2g.p, (const float*)Lw.ln2b.p, M, D, 0, x_next,
                                     h16, st, sink.acc(si_next, 2), gate_of(l + 1)));
            a16_cur = h16;
        }
        HIP_TRY(sink.done(si_next));
        x_cur = x_next;
        si_cur = si_next;
    }
    const float* enc_out = x_cur;  // what the DistilHuBERT heads read
    const void* enc_out16 = a16_cur;
    if (prel) {
        // encoder.layer_norm on the last residual stream (wav2vec2_model.py:3049-3050): the encoder output
        const int si_fin = si_hidden(NL);
        float* y = sink.slot32(si_fin) ? sink.slot32(si_fin) : other(x_cur);
        void* y16 = dt == F32 ? nullptr : (sink.slot16(si_fin) ? sink.slot16(si_fin) : xT);
        // featurize: this LayerNorm adds the term of its output (hidden list) or of its input (fairseq_layers / distiller)
        LnAcc fa = sink.wanted(si_fin) ? sink.acc(si_fin, 2) : sink.acc(si_cur, 1);
        Prof pr(e, st, "layernorm:enc", 0, gM * D * 8);
        HIP_TRY(launch_layernorm(dt, x_cur, (const float*)e->eln_g.p, (const float*)e->eln_b.p, M, D, 0, y, y16, st, fa));
        HIP_TRY(sink.done(si_fin));
        enc_out = y;
        enc_out16 = y16;
    }
    if (NH) {
        // DistilHuBERT prediction heads (distiller/model.py:155-161,245-258): Linear(D, NH*D) + GELU, then per head k
        // the (D x D) slice of SplitLinear on columns [k*D, (k+1)*D) of the intermediate
        {
            GemmParams g{};
            g.A = dt == F32 ? (const void*)enc_out : enc_out16;
            g.lda = D;
            g.W = e->head_w1.p;
            g.W_x3 = e->head_w13.p;
            g.bias = (const float*)e->head_b1.p;
            g.M = (int)M;
            g.N = NH * D;
            g.K = D;
            g.batches = 1;
            g.act = 1;
            g.ldo = (long)NH * D;
            if (dt == F32) g.out32 = (float*)hbuf; else g.out16 = hbuf;
            Prof pr(e, st, "gemm:head1", 2.0 * gM * NH * D * D, (gM * D + (double)NH * D * D + gM * NH * D) * es);
            HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
        }
        for (int k = 0; k < NH; ++k) {
            const int si = 1 + NL + k;
            float* dst = sink.slot32(si) ? sink.slot32(si) : (float*)tmp1;
            GemmParams g{};
            g.A = (const char*)hbuf + (size_t)k * D * es;
            g.lda = (long)NH * D;
            g.W = (const char*)e->head_w2.p + (size_t)k * D * D * es * (e->x2 ? 2 : 1);  // x2: rows are [hi | lo]
            g.W_x3 = e->head_w23.p ? (const char*)e->head_w23.p + (size_t)k * D * D * 4 : nullptr;
            g.bias = (const float*)e->head_b2.p + (long)k * D;
            g.M = (int)M;
            g.N = D;
            g.K = D;
            g.batches = 1;
            g.ldo = D;
            g.out32 = dst;
            g.out16 = sink.slot16(si);
            {
                Prof pr(e, st, "gemm:head2", 2.0 * gM * D * D, (gM * D + (double)D * D) * es + gM * D * 4);
                HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            }
            HIP_TRY(sink.emit(si, dst, false));
            HIP_TRY(sink.done(si));
        }
    }
    return 0;
}
}  // namespace

extern "C" {

static int parse_opts(s3enc_handle h, const s3enc_forward_opts* o, FwdOpts& fo) {
    if (!o) return 0;
    fo.selection = o->selection;
    fo.out_dtype = o->out_dtype;
    fo.featurize = o->featurize != 0;
    fo.feat_norm = o->feat_normalize != 0;
    fo.w = o->feat_w;
    if (fo.out_dtype < 0 || fo.out_dtype > 2) return fail("s3enc_forward_ex: out_dtype must be S3ENC_F32 / BF16 / F16");
    return 0;
}

int s3enc_forward(s3enc_handle h, const float* const* wavs, const int64_t* lengths, int32_t B, int64_t n_max, float* out,
                  int64_t layer_stride, void* stream) {
    if (!h || !wavs || !lengths) return fail("s3enc_forward: null argument");
    return forward_impl(h, wavs, lengths, B, n_max, FwdOpts(), out, layer_stride, (hipStream_t)stream);
}

int s3enc_forward_ex(s3enc_handle h, const float* const* wavs, const int64_t* lengths, int32_t B, int64_t n_max,
                     const s3enc_forward_opts* opts, void* out, int64_t layer_stride, void* stream) {
    if (!h || !wavs || !lengths) return fail("s3enc_forward_ex: null argument");
    FwdOpts fo;
    if (parse_opts(h, opts, fo)) return 1;
    return forward_impl(h, wavs, lengths, B, n_max, fo, out, layer_stride, (hipStream_t)stream);
}

int s3enc_forward_status(s3enc_handle h, int32_t wait, int32_t* status) {
    if (!h || !status) return fail("s3enc_forward_status: null argument");
    DeviceGuard dg(h->device);
    const int running = h->status_collect(wait != 0);
    *status = h->status_sticky | (running ? S3ENC_STATUS_PENDING : 0);
    h->status_sticky = 0;
    return 0;
}

int s3enc_num_states(s3enc_handle h, int32_t selection, int32_t* n) {
    if (!h || !n) return fail("s3enc_num_states: null argument");
    if (selection < 0 || selection > 2) return fail("s3enc_num_states: unknown selection");
    if ((h->cfg.family == S3ENC_DISTILLER || h->cfg.family == S3ENC_MULTIRES) && selection != S3ENC_SEL_HIDDEN)
        return fail("s3enc_num_states: DistilHuBERT / multires-HuBERT have one selection (their hidden_states list)");
    *n = num_states(h->cfg, selection);
    return 0;
}

int s3enc_forward_padded(s3enc_handle h, const float* pcm, int64_t row_stride, const int64_t* lengths, int32_t B, int64_t n_max,
                         float* out, int64_t layer_stride, void* stream) {
    if (!h || !pcm || !lengths) return fail("s3enc_forward_padded: null argument");
    if (B <= 0) return fail("s3enc_forward_padded: B must be positive");
    std::vector<const float*> ptrs(B);
    for (int b = 0; b < B; ++b) {
        if (lengths[b] > row_stride) return fail("s3enc_forward_padded: length exceeds row_stride");
        ptrs[b] = pcm + (long)b * row_stride;
    }
    return forward_impl(h, ptrs.data(), lengths, B, n_max, FwdOpts(), out, layer_stride, (hipStream_t)stream);
}

int s3enc_set_layer_events(s3enc_handle h, void* const* events, int32_t n) {
    if (!h) return fail("null handle");
    if (n == 0) {
        h->layer_events.clear();
        return 0;
    }
    if (!events || n < h->cfg.encoder_layers || n > num_states(h->cfg, S3ENC_SEL_HIDDEN))
        return fail("s3enc_set_layer_events: pass one event per state of the selection the forwards will use");
    h->layer_events.assign((hipEvent_t const*)events, (hipEvent_t const*)events + n);
    return 0;
}

int s3enc_profile_enable(s3enc_handle h, int32_t on) {
    if (!h) return fail("null handle");
    h->prof = on == 2 ? 2 : (on != 0);
    return 0;
}
int s3enc_profile_reset(s3enc_handle h) {
    if (!h) return fail("null handle");
    DeviceGuard dg(h->device);
    HIP_TRY(hipDeviceSynchronize());
    for (auto& r : h->recs) {
        h->ev_pool.push_back(r.a);
        h->ev_pool.push_back(r.b);
    }
    h->recs.clear();
    h->kinds.clear();
    h->kflops.clear();
    h->kbytes.clear();
    h->klaunches.clear();
    return 0;
}
int s3enc_profile_read(s3enc_handle h, s3enc_profile_entry* entries, int32_t max_entries, int32_t* n_entries) {
    if (!h || !entries || !n_entries) return fail("s3enc_profile_read: null argument");
    DeviceGuard dg(h->device);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<double> ms(h->kinds.size(), 0.0);
    for (auto& r : h->recs) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.kind] += t;
    }
    int n = 0;
    for (size_t k = 0; k < h->kinds.size() && n < max_entries; ++k, ++n) {
        memset(&entries[n], 0, sizeof(entries[n]));
        strncpy(entries[n].name, h->kinds[k].c_str(), sizeof(entries[n].name) - 1);
        entries[n].launches = h->klaunches[k];
        entries[n].ms = ms[k];
        entries[n].flops = h->kflops[k];
        entries[n].bytes = h->kbytes[k];
    }
    *n_entries = n;
    return 0;
}

int s3enc_debug_tap(s3enc_handle h, const char* name, float* host_out, int64_t max_elems, int64_t* n_elems) {
    if (!h || !name || !n_elems) return fail("s3enc_debug_tap: null argument");
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return fail(std::string("s3enc_debug_tap: unknown tap '") + name + "'");
    *n_elems = it->second.elems;
    if (!host_out) return 0;
    if (max_elems < it->second.elems) return fail("s3enc_debug_tap: buffer too small");
    DeviceGuard dg(h->device);
    HIP_TRY(hipDeviceSynchronize());
    if (it->second.dtype == F32) {
        HIP_TRY(hipMemcpy(host_out, it->second.p, (size_t)it->second.elems * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> tmp((size_t)it->second.elems);
        HIP_TRY(hipMemcpy(tmp.data(), it->second.p, tmp.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) host_out[i] = h_from16(tmp[i], it->second.dtype);
    }
    return 0;
}
}  // extern "C"
