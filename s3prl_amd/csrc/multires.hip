// multires.hip — multi-resolution HuBERT (upstream/multires_hubert/hubert_model.py:337-852,970-1266): the frame geometry of
// the U-net and its schedule behind post_extract_proj.  The blocks run the same GEMM / attention / LayerNorm kernels as the
// single-resolution encoders (engine.hip); the conv adapters are GEMMs over zero-bordered frame buffers + the row kernels
// of adapter.hip.
#include "engine_internal.h"

namespace s3e {

// output frames of a conv adapter on T frames (hubert_model.py:1038-1095,1146-1180,1232-1266): each stage is cut to
// min(conv length, skip-connection length); kind 0 ConvAdapter, 1 ConvDownsampler, 2 ConvUpsampler
long mr_adapter_frames(int k, long T, int up, int down, int kind) {
    long n = T;
    if (kind != 1) n = std::min<long>((long)up * T + k - 1, (long)up * T);
    if (kind != 2) {
        const long ld = (n + 2 * ((k - 1) / 2) - k) / down + 1;
        const long n2 = std::min(ld, (n + down - 1) / down);
        n = kind == 0 ? std::min(n2, ((long)up * T + down - 1) / down) : n2;
    }
    return n;
}

void mr_plan(const s3enc_config& c, long T0, MrPlan& plan) {
    const int R = c.mr_pairs + 1, k = c.mr_kernel;
    long ds[S3ENC_MAX_RES], lcm = 1;
    ds[0] = 1;
    for (int i = 0; i < c.n_conv; ++i) ds[0] *= c.conv_stride[i];
    for (int i = 0; i < R - 1; ++i) ds[i + 1] = ds[i] * c.mr_ratios[2 * i + 1] / c.mr_ratios[2 * i];  // hubert_model.py:512-533
    for (int i = 0; i < R; ++i) {
        long a = lcm, b = ds[i];
        while (b) {
            const long r = a % b;
            a = b;
            b = r;
        }
        lcm = lcm / a * ds[i];
    }
    int upf[S3ENC_MAX_RES], rev[S3ENC_MAX_RES];
    for (int i = 0; i < R; ++i) upf[i] = (int)(lcm / ds[R - 1 - i]);  // (sic) the expert reverses the list, expert.py:44-45
    for (int i = 0; i + 1 < R; ++i) rev[i] = upf[R - 2 - i];           // upsample_factor[::-1][1:]
    plan.blocks.clear();
    long T = T0, encT[S3ENC_MAX_RES];
    int ad = -1;
    long t_in = 0;
    for (int i = 0; i < R - 1; ++i) {
        plan.blocks.push_back({c.mr_layers[i], T, upf[i], ad, t_in, T});
        encT[i] = T;
        ad = i;
        t_in = T;
        T = mr_adapter_frames(k, T, c.mr_ratios[2 * i], c.mr_ratios[2 * i + 1], c.mr_plain ? 1 : 0);
    }
    plan.blocks.push_back({c.mr_layers[R - 1], T, upf[R - 1], ad, t_in, T});
    for (int i = 0; i < R - 1; ++i) {
        t_in = T;
        T = mr_adapter_frames(k, T, c.mr_ratios[2 * i + 1], c.mr_ratios[2 * i], c.mr_plain ? 2 : 0);
        const long res = encT[R - 2 - i];
        plan.blocks.push_back({c.mr_layers[R + i], T, rev[i], R - 1 + i, t_in, std::min(T, res)});
        T = std::min(T, res);
    }
    plan.T_out = -1;
    for (const auto& b : plan.blocks) {
        const long a = b.T * b.factor, p2 = (b.T + (b.T & 1)) * b.factor;  // outputs; layer inputs are padded to even T
        const long m = std::min(a, p2);
        if (plan.T_out < 0 || m < plan.T_out) plan.T_out = m;
    }
}

long output_frames(const s3enc_config& c, long n_samples) {
    const long T = conv_len(c, n_samples, c.n_conv);
    if (c.family != S3ENC_MULTIRES || T < 1) return T;
    MrPlan plan;
    mr_plan(c, T, plan);
    return plan.T_out;
}
// ---- multires-HuBERT behind post_extract_proj (multires_hubert/hubert_model.py:786-822) -------------------------------
// x (B, T0, D) fp32 with the padded frames zeroed -> encoders[i] -> conv adapter (down) ... middle_encoder (+ its input) ...
// conv adapter (up) -> decoders[i] (+ the matching encoder's output).  Every block's layer inputs and its output are
// states; each is written to its (B, T_out, D) slot with its frames repeated `factor` times (expert.py:26-27,93-101).
// The blocks run the same GEMM / attention / LayerNorm kernels as the single-resolution encoders; the adapter
// convolutions are GEMMs over zero-bordered frame buffers, their GroupNorm / GELU / skip passes are adapter.hip.
int multires_tail(s3enc_handle e, hipStream_t st, int B, const MrPlan& plan, const std::vector<const int*>& d_valid, float* xproj,
                  void* out, long layer_stride, const FwdOpts& fo) {
    const s3enc_config& c = e->cfg;
    const int D = c.embed_dim, F = c.ffn_dim, H = c.heads;
    const int dt = e->dtype, es = e->es;
    const bool prel = c.layer_norm_first != 0;
    const int R = c.mr_pairs + 1, NB = 2 * R - 1, k = c.mr_kernel, PADR = k - 1;
    const bool out16 = !fo.featurize && fo.out_dtype != F32;
    const float scale = std::sqrt(0.4f);  // sqrt(residual_scale), hubert_model.py:429,1036

    // capacities: frames of the widest block, rows of the widest zero-bordered operand, frames of the longest conv output
    long Tc = 0, Pc = 0, Lc = 0;
    for (const auto& bp : plan.blocks) {
        Tc = std::max(Tc, bp.T);
        if (bp.adapter < 0) continue;
        const AdapterW& aw = e->mr_adapters[bp.adapter];
        long rows = bp.T_in;
        Pc = std::max(Pc, rows + 2 * PADR);
        if (aw.kind != 1) {
            Lc = std::max(Lc, (rows + (k - 1) / aw.up_rate) * aw.up_rate);
            rows *= aw.up_rate;
            Pc = std::max(Pc, rows + 2 * PADR);
        }
        if (aw.kind != 2) Lc = std::max(Lc, (rows - 1) / aw.down_rate + 1);
    }
    const long Mc = (long)B * Tc;
    float *hA, *hB, *yM, *bufX, *tmp1, *tmp2, *pad32[2], *convout, *res[S3ENC_MAX_RES] = {};
    void *xT, *qkv, *attn, *hbuf, *pad16[2] = {};
    double* gpart;
    for (int pass = 0; pass < 2; ++pass) {
        Bump wb(pass ? e->ws_mr.p : nullptr);
        hA = (float*)wb.take((size_t)Mc * D * 4);
        hB = (float*)wb.take((size_t)Mc * D * 4);
        yM = (float*)wb.take((size_t)Mc * D * 4);
        bufX = (float*)wb.take((size_t)Mc * D * 4);
        tmp1 = (float*)wb.take((size_t)Mc * D * 4);
        tmp2 = (float*)wb.take((size_t)Mc * D * 4);
        xT = wb.take((size_t)Mc * D * 4);
        qkv = wb.take((size_t)Mc * 3 * D * es);
        attn = wb.take((size_t)Mc * D * es);
        hbuf = wb.take((size_t)Mc * F * es);
        for (int i = 0; i < R - 1; ++i) res[i] = (float*)wb.take((size_t)B * plan.blocks[i].T * D * 4);
        for (int i = 0; i < 2; ++i) {
            pad32[i] = (float*)wb.take((size_t)B * Pc * D * 4);
            if (dt != F32) pad16[i] = wb.take((size_t)B * Pc * D * 2);
        }
        convout = (float*)wb.take((size_t)B * Lc * D * 4);
        gpart = (double*)wb.take((size_t)B * GS_BLOCKS * 2 * 8);
        if (!pass) HIP_TRY(e->ws_mr.ensure_on_stream(wb.off + 4096, st));
    }

    int si = 0;
    bool first_term = true;
    // a state (B, T, D) -> slot si of the caller's slab at the finest frame rate; featurize: only its term of the sum
    auto emit = [&](const float* x, long T, int factor) -> int {
        if (fo.featurize) {
            if (fo.w[si] != 0.f) {
                LnAcc fa;
                fa.acc = (float*)out;
                fa.w = fo.w[si];
                fa.mode = 1;
                fa.norm = fo.feat_norm;
                fa.init = first_term;
                first_term = false;
                Prof pr(e, st, "emit_state", 0, (double)B * plan.T_out * D * (4.0 / factor + 8));
                HIP_TRY(launch_emit_upsampled_acc(x, T * D, factor, B, (int)plan.T_out, D, fa, st));
            }
            ++si;
            return 0;
        }
        float* o32 = out16 ? nullptr : (float*)out + (long)si * layer_stride;
        void* o16 = out16 ? (void*)((u16*)out + (long)si * layer_stride) : nullptr;
        {
            Prof pr(e, st, "emit_state", 0, (double)B * plan.T_out * D * (4.0 / factor + (out16 ? 2 : 4)));
            HIP_TRY(launch_emit_upsampled(out16 ? dt : (int)F32, x, T * D, factor, B, (int)plan.T_out, D, o32, o16, st));
        }
        if (si < (int)e->layer_events.size()) HIP_TRY(hipEventRecord(e->layer_events[si], st));
        ++si;
        return 0;
    };

    // one TransformerEncoder of the U-net (wav2vec2_model.py:3046-3121 with skip_pos_conv / override_encoder_layer):
    // x (B, T, D) fp32, padded frames zero (the producer wrote them so); the block's output lands in y_out
    auto run_block = [&](int bi, float* x, float* y_out) -> int {
        BlockW& bw = e->mr_blocks[bi];
        const MrBlockPlan& bp = plan.blocks[bi];
        const long T = bp.T, M = (long)B * T;
        const double gM = (double)M;
        const int NLb = (int)bw.layers.size();
        float* cur = x;
        auto pick = [&](const float* busy) { return busy == hA ? hB : hA; };
        if (bi == 0) {  // only encoders[0] keeps the positional conv (hubert_model.py:434-449)
            PosConvParams p{};
            p.x = x;
            p.w = e->x3 ? e->pos_w3.p : e->pos_w.p;
            p.bias = (const float*)e->pos_b.p;
            p.out = hA;
            p.B = B;
            p.T = (int)T;
            p.D = D;
            p.G = c.conv_pos_groups;
            p.K = e->pos_k;
            p.pad = e->pos_pad;
            Prof pr(e, st, "posconv", 2.0 * gM * D * (D / p.G) * c.conv_pos, gM * D * 8 + (double)D * (D / p.G) * p.K * 4);
            HIP_TRY(e->x3 ? launch_posconv16(3, p, st) : (dt == F32 ? launch_posconv(p, st) : launch_posconv16(dt, p, st)));
            cur = hA;
        }
        if (!prel) {
            float* h0 = pick(cur);
            Prof pr(e, st, "layernorm:enc", 0, gM * D * (8 + (dt == F32 ? 0 : es)));
            HIP_TRY(launch_layernorm(dt, cur, (const float*)bw.eln_g.p, (const float*)bw.eln_b.p, M, D, 0, h0,
                                     dt == F32 ? nullptr : xT, st));
            cur = h0;
        }
        if (emit(cur, T, bp.factor)) return 1;  // the input of the block's first layer
        for (int l = 0; l < NLb; ++l) {
            LayerW& Lw = bw.layers[l];
            const bool lastl = l == NLb - 1;
            const void* a_in;
            if (prel) {
                Prof pr(e, st, "layernorm:ln1", 0, gM * D * (4 + es));
                HIP_TRY(launch_layernorm(dt, cur, (const float*)Lw.ln1g.p, (const float*)Lw.ln1b.p, M, D, 0,
                                         dt == F32 ? (float*)xT : nullptr, dt == F32 ? nullptr : xT, st));
                a_in = xT;
            } else {
                a_in = dt == F32 ? (const void*)cur : (const void*)xT;
            }
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
            }
            {
                AttnParams a{};
                a.qkv = qkv;
                a.out = attn;
                a.valid = d_valid[bi];
                a.B = B;
                a.T = (int)T;
                a.H = H;
                Prof pr(e, st, "attention", 4.0 * B * H * (double)T * T * 64, gM * 4 * D * es);
                HIP_TRY(launch_attention(e->x3 ? 3 : dt, a, st));
            }
            {
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
                g.residual = cur;
                g.out32 = tmp1;
                Prof pr(e, st, "gemm:out_proj", 2.0 * gM * D * D, (gM * D + (double)D * D) * es + gM * D * 8);
                HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            }
            const float* ffn_res;
            const void* ffn_in;
            if (prel) {
                Prof pr(e, st, "layernorm:ln2", 0, gM * D * (4 + es));
                HIP_TRY(launch_layernorm(dt, tmp1, (const float*)Lw.ln2g.p, (const float*)Lw.ln2b.p, M, D, 0,
                                         dt == F32 ? (float*)xT : nullptr, dt == F32 ? nullptr : xT, st));
                ffn_res = tmp1;
                ffn_in = xT;
            } else {
                Prof pr(e, st, "layernorm:ln1", 0, gM * D * (8 + (dt == F32 ? 0 : es)));
                HIP_TRY(launch_layernorm(dt, tmp1, (const float*)Lw.ln1g.p, (const float*)Lw.ln1b.p, M, D, 0, tmp2,
                                         dt == F32 ? nullptr : xT, st));
                ffn_res = tmp2;
                ffn_in = dt == F32 ? (const void*)tmp2 : (const void*)xT;
            }
            {
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
            float* nxt = (!prel && lastl) ? y_out : pick(cur);
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
                g.residual = ffn_res;
                g.out32 = prel ? nxt : tmp1;
                Prof pr(e, st, "gemm:fc2", 2.0 * gM * D * F, (gM * F + (double)D * F) * es + gM * D * 8);
                HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
            }
            if (!prel) {
                Prof pr(e, st, "layernorm:ln2", 0, gM * D * (8 + (dt == F32 ? 0 : es)));
                HIP_TRY(launch_layernorm(dt, tmp1, (const float*)Lw.ln2g.p, (const float*)Lw.ln2b.p, M, D, 0, nxt,
                                         dt == F32 ? nullptr : xT, st));
            }
            cur = nxt;
            // post-LN: the layer output is the next layer's input / the block output; pre-LN: the last stream is not a state
            if (!prel || !lastl)
                if (emit(cur, T, bp.factor)) return 1;
        }
        if (prel) {  // encoder.layer_norm on the last residual stream (wav2vec2_model.py:3049-3050): the block output
            {
                Prof pr(e, st, "layernorm:enc", 0, gM * D * 8);
                HIP_TRY(launch_layernorm(F32, cur, (const float*)bw.eln_g.p, (const float*)bw.eln_b.p, M, D, 0, y_out, nullptr, st));
            }
            if (emit(y_out, T, bp.factor)) return 1;
        }
        return 0;
    };

    // One convolution + GroupNorm(1, D) statistics of a conv adapter stage: A rows are k_eff * D contiguous elements of the
    // zero-bordered buffer (lead = PADR rows), output (B, L, D) fp32 in convout; returns L through `frames`
    auto run_conv = [&](const AdapterConvW& cw, bool transposed, int stride, int which, long rows, long* frames) -> int {
        const long total = rows + 2 * PADR;
        GemmParams g{};
        const char* base = dt == F32 ? (const char*)pad32[which] : (const char*)pad16[which];
        long Mrows;
        if (transposed) {
            const int KT = (k + stride - 1) / stride;
            Mrows = rows + (k - 1) / stride;  // Q; the (Q, stride * D) output is the (stride * Q, D) sequence
            g.A = base + (size_t)(PADR - (KT - 1)) * D * es;
            g.lda = D;
            g.N = stride * D;
            g.K = KT * D;
            *frames = Mrows * stride;
        } else {
            const int pd = (k - 1) / 2;
            Mrows = (rows + 2 * pd - k) / stride + 1;
            g.A = base + (size_t)(PADR - pd) * D * es;
            g.lda = (long)stride * D;
            g.N = D;
            g.K = k * D;
            *frames = Mrows;
        }
        g.a_bs = total * D;
        g.W = cw.w.p;
        g.W_x3 = cw.w3.p;
        g.M = (int)Mrows;
        g.batches = B;
        g.out32 = convout;
        g.ldo = g.N;
        g.o_bs = *frames * D;
        {
            // algorithmic flops: the k real taps (the zero taps that square up the transposed conv's phases are not counted)
            Prof pr(e, st, "gemm:adapter", 2.0 * B * (transposed ? (double)rows : (double)Mrows) * D * D * k,
                    ((double)B * total * D + (double)g.N * g.K) * es + (double)B * *frames * D * 4);
            HIP_TRY(launch_gemm(dt, wsplit_of(e, g), st));
        }
        Prof pr(e, st, "adapter_stats", 0, (double)B * *frames * D * 4);
        HIP_TRY(launch_group1_stats(convout, *frames * D, *frames * D, B, gpart, st));
        return 0;
    };

    // a conv adapter (hubert_model.py:1038-1078 ConvAdapter, :1146-1167 ConvDownsampler, :1232-1250 ConvUpsampler):
    // input rows a[t] (+ b2[t]) of T_in frames -> bufX (B, n_out, D) fp32 with the frames >= zero_next[b] zeroed
    auto run_adapter = [&](const AdapterW& aw, const float* a, long a_bs, const float* b2, long b_bs, long T_in, const int* zero_next,
                           long n_expect) -> int {
        {
            PadCopyParams pc{};
            pc.a = a;
            pc.a_bs = a_bs;
            pc.b = b2;
            pc.b_bs = b_bs;
            pc.B = B;
            pc.rows = (int)T_in;
            pc.D = D;
            pc.lead = PADR;
            pc.total = (int)(T_in + 2 * PADR);
            pc.out32 = pad32[0];
            pc.out16 = pad16[0];
            Prof pr(e, st, "adapter_pad", 0, (double)B * T_in * D * (b2 ? 8 : 4) + (double)B * pc.total * D * (4 + (dt == F32 ? 0 : 2)));
            HIP_TRY(launch_pad_copy(dt, pc, st));
        }
        const long total0 = T_in + 2 * PADR;
        long rows = T_in, frames = 0;
        int cur = 0;  // operand buffer holding the current stage's input
        AdapterApplyParams ap{};
        ap.conv = convout;
        ap.partial = gpart;
        ap.scale = scale;
        ap.B = B;
        ap.D = D;
        ap.fast_gelu = e->x3;
        if (aw.kind != 1) {  // upsample_conv + skip from repeat_interleave(x, up)
            if (run_conv(aw.up, true, aw.up_rate, cur, rows, &frames)) return 1;
            const long n1 = std::min(frames, rows * aw.up_rate);
            ap.conv_bs = frames * D;
            ap.count = (double)frames * D;
            ap.gamma = (const float*)aw.up.g.p;
            ap.beta = (const float*)aw.up.b.p;
            ap.r1 = pad32[0] + (long)PADR * D;
            ap.r1_bs = total0 * D;
            ap.r1_mul = 1;
            ap.r1_div = aw.up_rate;
            ap.r2 = nullptr;
            ap.rows = (int)n1;
            const bool fin = aw.kind == 2;
            ap.lead = fin ? 0 : PADR;
            ap.total = (int)(fin ? n1 : n1 + 2 * PADR);
            ap.zero_from = fin ? zero_next : nullptr;
            ap.out32 = fin ? bufX : pad32[1];
            ap.out16 = fin ? nullptr : pad16[1];
            Prof pr(e, st, "adapter_apply", 0, (double)B * n1 * D * 12);
            HIP_TRY(launch_adapter_apply(dt, ap, st));
            rows = n1;
            cur = 1;
        }
        if (aw.kind != 2) {  // downsample_conv + skip x[::down] (+ highway repeat_interleave(x0, up)[::down])
            if (run_conv(aw.down, false, aw.down_rate, cur, rows, &frames)) return 1;
            const long n2 = std::min(frames, (rows + aw.down_rate - 1) / aw.down_rate);
            const long n3 = aw.kind == 0 ? std::min(n2, (T_in * aw.up_rate + aw.down_rate - 1) / aw.down_rate) : n2;
            ap.conv_bs = frames * D;
            ap.count = (double)frames * D;
            ap.gamma = (const float*)aw.down.g.p;
            ap.beta = (const float*)aw.down.b.p;
            ap.r1 = pad32[cur] + (long)PADR * D;
            ap.r1_bs = (rows + 2 * PADR) * D;
            ap.r1_mul = aw.down_rate;
            ap.r1_div = 1;
            if (aw.kind == 0) {
                ap.r2 = pad32[0] + (long)PADR * D;
                ap.r2_bs = total0 * D;
                ap.r2_mul = aw.down_rate;
                ap.r2_div = aw.up_rate;
            } else {
                ap.r2 = nullptr;
            }
            ap.rows = (int)n3;
            ap.lead = 0;
            ap.total = (int)n3;
            ap.zero_from = zero_next;
            ap.out32 = bufX;
            ap.out16 = nullptr;
            Prof pr(e, st, "adapter_apply", 0, (double)B * n3 * D * (aw.kind == 0 ? 16 : 12));
            HIP_TRY(launch_adapter_apply(dt, ap, st));
            rows = n3;
        }
        if (rows != n_expect) return fail("multires: adapter length does not match the plan (internal error)");
        return 0;
    };

    float* x = xproj;
    for (int bi = 0; bi < NB; ++bi) {
        const MrBlockPlan& bp = plan.blocks[bi];
        if (bp.adapter >= 0) {
            const AdapterW& aw = e->mr_adapters[bp.adapter];
            const float *a, *b2 = nullptr;
            long a_bs, b_bs = 0;
            if (bi <= R - 1) {  // an encoder's output -> downsample_modules[bi - 1]
                a = res[bi - 1];
                a_bs = plan.blocks[bi - 1].T * D;
            } else if (bi == R) {  // x = x + middle_encoder(x) (hubert_model.py:801-802): x is the middle block's zeroed input
                a = bufX;
                a_bs = plan.blocks[bi - 1].T * D;
                b2 = yM;
                b_bs = a_bs;
            } else {  // align_size_sum(decoder output, the matching encoder output) (:816)
                const int ri = R - 2 - (bi - 1 - R);
                a = yM;
                a_bs = plan.blocks[bi - 1].T * D;
                b2 = res[ri];
                b_bs = plan.blocks[ri].T * D;
            }
            if (run_adapter(aw, a, a_bs, b2, b_bs, bp.T_in, d_valid[bi], bp.T)) return 1;
            x = bufX;
        }
        if (run_block(bi, x, bi < R - 1 ? res[bi] : yM)) return 1;
    }
    if (si != num_states(c, S3ENC_SEL_HIDDEN)) return fail("multires: state count mismatch (internal error)");
    return 0;
}
}  // namespace s3e
