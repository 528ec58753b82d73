"""Second, independent oracle: Hugging Face ``transformers.HubertModel`` — a third-party implementation
of the same architecture the reference itself wraps (upstream/hf_hubert/expert.py:12-41) and cross-checks
(utility/compare_wav2vec2.py).  Our synthetic weights are mapped name-for-name into the HF module and its
``output_hidden_states`` compared with oracle/encoder_oracle.py on an equal-length batch (HF derives the frame mask of
padded batches with the wav2vec2 conv-length rule, not HuBERT's chunk rule — SURVEY A.2 — so padded batches are not
comparable).  Post-LN (base-style) and pre-LN (large-style, layer-norm extractor) variants."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

from oracle import encoder_oracle as O  # noqa: E402


def _hf_model(cfg, weights):
    from transformers import HubertConfig, HubertModel

    hc = HubertConfig(
        hidden_size=cfg.encoder_embed_dim, num_hidden_layers=cfg.encoder_layers,
        num_attention_heads=cfg.encoder_attention_heads, intermediate_size=cfg.encoder_ffn_embed_dim,
        conv_dim=[c for c, _, _ in cfg.conv_layers], conv_kernel=[k for _, k, _ in cfg.conv_layers],
        conv_stride=[s for _, _, s in cfg.conv_layers], conv_bias=cfg.conv_bias,
        num_conv_pos_embeddings=cfg.conv_pos, num_conv_pos_embedding_groups=cfg.conv_pos_groups,
        feat_extract_norm="group" if cfg.extractor_mode == "default" else "layer",
        do_stable_layer_norm=cfg.layer_norm_first, hidden_act="gelu", feat_extract_activation="gelu",
        hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
        feat_proj_layer_norm=True, attn_implementation="eager", layer_norm_eps=1e-5)
    m = HubertModel(hc).eval()
    t = lambda k: torch.from_numpy(np.ascontiguousarray(weights[k]))
    sd = {}
    for i in range(len(cfg.conv_layers)):
        p, q = f"feature_extractor.conv_layers.{i}", f"feature_extractor.conv_layers.{i}"
        sd[f"{q}.conv.weight"] = t(f"{p}.0.weight")
        if cfg.conv_bias:
            sd[f"{q}.conv.bias"] = t(f"{p}.0.bias")
        if cfg.extractor_mode == "layer_norm":
            sd[f"{q}.layer_norm.weight"], sd[f"{q}.layer_norm.bias"] = t(f"{p}.2.1.weight"), t(f"{p}.2.1.bias")
        elif i == 0:
            sd[f"{q}.layer_norm.weight"], sd[f"{q}.layer_norm.bias"] = t(f"{p}.2.weight"), t(f"{p}.2.bias")
    sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"] = t("layer_norm.weight"), t("layer_norm.bias")
    sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"] = \
        t("post_extract_proj.weight"), t("post_extract_proj.bias")
    sd["encoder.pos_conv_embed.conv.bias"] = t("encoder.pos_conv.0.bias")
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = t("encoder.pos_conv.0.weight_g")
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = t("encoder.pos_conv.0.weight_v")
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = t("encoder.layer_norm.weight"), t("encoder.layer_norm.bias")
    for l in range(cfg.encoder_layers):
        p, q = f"encoder.layers.{l}", f"encoder.layers.{l}"
        for n in ("q", "k", "v", "out"):
            sd[f"{q}.attention.{n}_proj.weight"] = t(f"{p}.self_attn.{n}_proj.weight")
            sd[f"{q}.attention.{n}_proj.bias"] = t(f"{p}.self_attn.{n}_proj.bias")
        sd[f"{q}.layer_norm.weight"], sd[f"{q}.layer_norm.bias"] = t(f"{p}.self_attn_layer_norm.weight"), t(f"{p}.self_attn_layer_norm.bias")
        sd[f"{q}.feed_forward.intermediate_dense.weight"], sd[f"{q}.feed_forward.intermediate_dense.bias"] = t(f"{p}.fc1.weight"), t(f"{p}.fc1.bias")
        sd[f"{q}.feed_forward.output_dense.weight"], sd[f"{q}.feed_forward.output_dense.bias"] = t(f"{p}.fc2.weight"), t(f"{p}.fc2.bias")
        sd[f"{q}.final_layer_norm.weight"], sd[f"{q}.final_layer_norm.bias"] = t(f"{p}.final_layer_norm.weight"), t(f"{p}.final_layer_norm.bias")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert set(missing) <= {"masked_spec_embed"}, missing
    return m


@pytest.mark.parametrize("name", ["tiny_hubert", "tiny_hubert_large"])
def test_oracle_matches_huggingface_hubert(name):
    from s3prl_amd.synth import named_config, synth_weights, synth_wavs

    cfg = named_config(name)
    weights = synth_weights(cfg, 7)
    wavs = synth_wavs([8000, 8000, 8000], 3)
    m = _hf_model(cfg, weights)
    x = torch.from_numpy(np.stack(wavs))
    if cfg.normalize:  # Wav2Vec2FeatureExtractor(do_normalize=True) == F.layer_norm(wav, wav.shape) (hubert/expert.py:57-58)
        x = torch.nn.functional.layer_norm(x, (x.shape[1],))
    with torch.no_grad():
        hf = m(x, output_hidden_states=True).hidden_states
    ours = O.forward(cfg, weights, wavs, dtype=np.float32)
    assert len(hf) == len(ours) == cfg.encoder_layers + 1
    for l, (a, b) in enumerate(zip(hf, ours)):
        # pre-LN HF models return the un-normalised stream for l < NL and the normalised last state, like the reference
        assert O.rel_err(a.numpy(), b) < 2e-4, f"layer {l}: {O.rel_err(a.numpy(), b):.2e}"


# ---- WavLM: the gated relative-position bias against a third-party implementation -------------------------------
# transformers.WavLMModel re-implements wavlm/modules.py:418-462 (bucketed relative positions, one embedding table in layer 0,
# reused by every later layer) and :535-551 (the GRU-style gate from the layer input split into heads).  Weight names per SURVEY
# A.10: relative_attention_bias -> rel_attn_embed, grep_linear -> gru_rel_pos_linear, grep_a -> gru_rel_pos_const.

def _hf_wavlm(cfg, weights):
    from transformers import WavLMConfig, WavLMModel

    hc = WavLMConfig(
        hidden_size=cfg.encoder_embed_dim, num_hidden_layers=cfg.encoder_layers,
        num_attention_heads=cfg.encoder_attention_heads, intermediate_size=cfg.encoder_ffn_embed_dim,
        conv_dim=[c for c, _, _ in cfg.conv_layers], conv_kernel=[k for _, k, _ in cfg.conv_layers],
        conv_stride=[s for _, _, s in cfg.conv_layers], conv_bias=cfg.conv_bias,
        num_conv_pos_embeddings=cfg.conv_pos, num_conv_pos_embedding_groups=cfg.conv_pos_groups,
        feat_extract_norm="group" if cfg.extractor_mode == "default" else "layer",
        do_stable_layer_norm=cfg.layer_norm_first, hidden_act="gelu", feat_extract_activation="gelu",
        hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
        num_buckets=cfg.num_buckets, max_bucket_distance=cfg.max_distance, layer_norm_eps=1e-5,
        mask_time_prob=0.0, mask_feature_prob=0.0)
    m = WavLMModel(hc).eval()
    t = lambda k: torch.from_numpy(np.ascontiguousarray(weights[k]))
    sd = {}
    for i in range(len(cfg.conv_layers)):
        p = f"feature_extractor.conv_layers.{i}"
        sd[f"{p}.conv.weight"] = t(f"{p}.0.weight")
        if cfg.conv_bias:
            sd[f"{p}.conv.bias"] = t(f"{p}.0.bias")
        if cfg.extractor_mode == "layer_norm":
            sd[f"{p}.layer_norm.weight"], sd[f"{p}.layer_norm.bias"] = t(f"{p}.2.1.weight"), t(f"{p}.2.1.bias")
        elif i == 0:
            sd[f"{p}.layer_norm.weight"], sd[f"{p}.layer_norm.bias"] = t(f"{p}.2.weight"), t(f"{p}.2.bias")
    sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"] = t("layer_norm.weight"), t("layer_norm.bias")
    sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"] = \
        t("post_extract_proj.weight"), t("post_extract_proj.bias")
    sd["encoder.pos_conv_embed.conv.bias"] = t("encoder.pos_conv.0.bias")
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = t("encoder.pos_conv.0.weight_g")
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = t("encoder.pos_conv.0.weight_v")
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = t("encoder.layer_norm.weight"), t("encoder.layer_norm.bias")
    for l in range(cfg.encoder_layers):
        p = f"encoder.layers.{l}"
        for n in ("q", "k", "v", "out"):
            sd[f"{p}.attention.{n}_proj.weight"] = t(f"{p}.self_attn.{n}_proj.weight")
            sd[f"{p}.attention.{n}_proj.bias"] = t(f"{p}.self_attn.{n}_proj.bias")
        if l == 0:
            sd[f"{p}.attention.rel_attn_embed.weight"] = t(f"{p}.self_attn.relative_attention_bias.weight")
        sd[f"{p}.attention.gru_rel_pos_linear.weight"] = t(f"{p}.self_attn.grep_linear.weight")
        sd[f"{p}.attention.gru_rel_pos_linear.bias"] = t(f"{p}.self_attn.grep_linear.bias")
        sd[f"{p}.attention.gru_rel_pos_const"] = t(f"{p}.self_attn.grep_a")
        sd[f"{p}.layer_norm.weight"], sd[f"{p}.layer_norm.bias"] = t(f"{p}.self_attn_layer_norm.weight"), t(f"{p}.self_attn_layer_norm.bias")
        sd[f"{p}.feed_forward.intermediate_dense.weight"], sd[f"{p}.feed_forward.intermediate_dense.bias"] = t(f"{p}.fc1.weight"), t(f"{p}.fc1.bias")
        sd[f"{p}.feed_forward.output_dense.weight"], sd[f"{p}.feed_forward.output_dense.bias"] = t(f"{p}.fc2.weight"), t(f"{p}.fc2.bias")
        sd[f"{p}.final_layer_norm.weight"], sd[f"{p}.final_layer_norm.bias"] = t(f"{p}.final_layer_norm.weight"), t(f"{p}.final_layer_norm.bias")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert set(missing) <= {"masked_spec_embed"}, missing
    return m


def _ragged_lengths_both_mask_rules_agree_on(cfg, n_max, want=2):
    """Hugging Face masks padded frames by the conv-length rule, the reference's WavLM by the chunk rule; they
    differ by a frame for some lengths.  Pick short lengths where both give the same count, so a RAGGED batch is comparable."""
    out = []
    n = int(n_max * 0.55)
    while len(out) < want and n < n_max:
        if cfg.valid_frames(n, n_max) == cfg.conv_lengths(n)[-1]:
            out.append(n)
            n += n_max // 5
        n += 1
    assert len(out) == want
    return out


@pytest.mark.parametrize("name,profile", [("tiny_wavlm", "synthetic"), ("tiny_wavlm_large", "synthetic"), ("tiny_wavlm", "pretrained_like")])
def test_oracle_matches_huggingface_wavlm_on_a_ragged_batch(name, profile):
    from s3prl_amd.synth import named_config, synth_weights, synth_wavs

    cfg = named_config(name)
    weights = synth_weights(cfg, 9, profile)
    n_max = 9000
    lengths = [n_max] + _ragged_lengths_both_mask_rules_agree_on(cfg, n_max)
    wavs = synth_wavs(lengths, 4)
    m = _hf_wavlm(cfg, weights)
    x = torch.zeros(len(wavs), n_max)
    mask = torch.zeros(len(wavs), n_max, dtype=torch.long)
    for b, w in enumerate(wavs):
        w = torch.from_numpy(w)
        if cfg.normalize:  # per utterance over its own samples (wavlm/expert.py:71-73 -> F.layer_norm(wav, wav.shape))
            w = torch.nn.functional.layer_norm(w, w.shape)
        x[b, : len(w)] = w
        mask[b, : len(w)] = 1
    with torch.no_grad():
        hf = m(x, attention_mask=mask, output_hidden_states=True).hidden_states
    ours = O.forward(cfg, weights, wavs, dtype=np.float32)
    assert len(hf) == len(ours) == cfg.encoder_layers + 1
    T = ours[0].shape[1]
    valid = [cfg.valid_frames(n, n_max) for n in lengths]
    assert valid[1] < T and valid[2] < T  # really ragged
    for l, (a, b) in enumerate(zip(hf, ours)):
        a = a.numpy()
        assert a.shape == b.shape
        for u in range(len(wavs)):  # padded frames carry implementation-defined values in both; the valid ones must agree
            e = O.rel_err(a[u, : valid[u]], b[u, : valid[u]])
            assert e < 1e-4, f"layer {l}, utterance {u}: {e:.2e}"
