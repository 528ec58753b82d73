"""Writes a Hugging Face checkpoint directory (config.json + model.safetensors + preprocessor_config.json) for one of our
synthetic configurations — the format ``transformers``' ``HubertModel`` / ``Wav2Vec2Model.from_pretrained`` reads.  Used by
tests/golden/make_golden.py (where the REFERENCE's hf_hubert / hf_wav2vec2 experts load it through ``transformers``, which
also validates the format) and by the GPU tests (where ``s3prl_amd.hf`` reads it without ``transformers``)."""

import json
import os

import numpy as np


def hf_names(cfg, weights):
    """our (fairseq) parameter names -> Hugging Face names"""
    sd = {}
    for i in range(len(cfg.conv_layers)):
        p, q = f"feature_extractor.conv_layers.{i}", f"feature_extractor.conv_layers.{i}"
        sd[f"{q}.conv.weight"] = weights[f"{p}.0.weight"]
        if cfg.conv_bias:
            sd[f"{q}.conv.bias"] = weights[f"{p}.0.bias"]
        if cfg.extractor_mode == "layer_norm":
            sd[f"{q}.layer_norm.weight"], sd[f"{q}.layer_norm.bias"] = weights[f"{p}.2.1.weight"], weights[f"{p}.2.1.bias"]
        elif i == 0:
            sd[f"{q}.layer_norm.weight"], sd[f"{q}.layer_norm.bias"] = weights[f"{p}.2.weight"], weights[f"{p}.2.bias"]
    sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"] = weights["layer_norm.weight"], weights["layer_norm.bias"]
    sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"] = \
        weights["post_extract_proj.weight"], weights["post_extract_proj.bias"]
    sd["encoder.pos_conv_embed.conv.bias"] = weights["encoder.pos_conv.0.bias"]
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = weights["encoder.pos_conv.0.weight_g"]
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = weights["encoder.pos_conv.0.weight_v"]
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = weights["encoder.layer_norm.weight"], weights["encoder.layer_norm.bias"]
    for l in range(cfg.encoder_layers):
        p, q = f"encoder.layers.{l}", f"encoder.layers.{l}"
        for n in ("q", "k", "v", "out"):
            sd[f"{q}.attention.{n}_proj.weight"] = weights[f"{p}.self_attn.{n}_proj.weight"]
            sd[f"{q}.attention.{n}_proj.bias"] = weights[f"{p}.self_attn.{n}_proj.bias"]
        sd[f"{q}.layer_norm.weight"], sd[f"{q}.layer_norm.bias"] = weights[f"{p}.self_attn_layer_norm.weight"], weights[f"{p}.self_attn_layer_norm.bias"]
        sd[f"{q}.feed_forward.intermediate_dense.weight"], sd[f"{q}.feed_forward.intermediate_dense.bias"] = weights[f"{p}.fc1.weight"], weights[f"{p}.fc1.bias"]
        sd[f"{q}.feed_forward.output_dense.weight"], sd[f"{q}.feed_forward.output_dense.bias"] = weights[f"{p}.fc2.weight"], weights[f"{p}.fc2.bias"]
        sd[f"{q}.final_layer_norm.weight"], sd[f"{q}.final_layer_norm.bias"] = weights[f"{p}.final_layer_norm.weight"], weights[f"{p}.final_layer_norm.bias"]
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in sd.items()}


def write_hf_dir(path, cfg, weights, model_type="hubert", do_normalize=None):
    from safetensors.numpy import save_file

    os.makedirs(path, exist_ok=True)
    if do_normalize is None:
        do_normalize = cfg.extractor_mode == "layer_norm"
    hc = dict(
        model_type=model_type, architectures=["HubertModel" if model_type == "hubert" else "Wav2Vec2Model"],
        hidden_size=cfg.encoder_embed_dim, num_hidden_layers=cfg.encoder_layers,
        num_attention_heads=cfg.encoder_attention_heads, intermediate_size=cfg.encoder_ffn_embed_dim,
        conv_dim=[c for c, _, _ in cfg.conv_layers], conv_kernel=[k for _, k, _ in cfg.conv_layers],
        conv_stride=[s for _, _, s in cfg.conv_layers], num_feat_extract_layers=len(cfg.conv_layers), conv_bias=bool(cfg.conv_bias),
        num_conv_pos_embeddings=cfg.conv_pos, num_conv_pos_embedding_groups=cfg.conv_pos_groups,
        feat_extract_norm="group" if cfg.extractor_mode == "default" else "layer",
        do_stable_layer_norm=bool(cfg.layer_norm_first), hidden_act="gelu", feat_extract_activation="gelu",
        hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0,
        layer_norm_eps=1e-5, vocab_size=32, mask_time_prob=0.0, mask_feature_prob=0.0)
    if model_type == "hubert":
        hc["feat_proj_layer_norm"] = True
    else:  # keep the quantiser / projection heads of Wav2Vec2Config tiny; unused by Wav2Vec2Model
        hc.update(num_codevectors_per_group=2, num_codevector_groups=2, codevector_dim=8, proj_codevector_dim=8)
    json.dump(hc, open(os.path.join(path, "config.json"), "w"), indent=1)
    save_file(hf_names(cfg, weights), os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    json.dump(dict(feature_extractor_type="Wav2Vec2FeatureExtractor", do_normalize=bool(do_normalize), feature_size=1,
                   padding_side="right", padding_value=0.0, return_attention_mask=True, sampling_rate=16000),
              open(os.path.join(path, "preprocessor_config.json"), "w"), indent=1)
    return path
