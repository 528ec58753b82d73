"""Reader for Hugging Face ``transformers`` checkpoints of the same architectures (``HubertModel`` / ``Wav2Vec2Model``),
what the reference's ``hf_hubert`` / ``hf_wav2vec2`` upstreams load with ``from_pretrained`` (upstream/hf_hubert/expert.py:12-41,
upstream/hf_wav2vec2/expert.py).  No ``transformers`` import: a checkpoint directory is ``config.json`` +
``model.safetensors`` (or ``pytorch_model.bin``) + optionally ``preprocessor_config.json``; the tensors are renamed to the
fairseq names the rest of the package (and libs3enc's weight packer) use.

Semantics of the HF forward that differ from the fairseq experts and are reproduced (transformers ``modeling_hubert`` /
``modeling_wav2vec2``): the frame mask is ALWAYS wav2vec2's conv-length rule (``_get_feature_vector_attention_mask``),
also for HuBERT; the waveform normalisation of ``Wav2Vec2FeatureExtractor(do_normalize=True)`` uses eps 1e-7
(``zero_mean_unit_var_norm``), not ``F.layer_norm``'s 1e-5.
"""

from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np

from .config import EncoderConfig


def _load_state(path: str) -> Dict[str, np.ndarray]:
    st = os.path.join(path, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.numpy import load_file

        return dict(load_file(st))
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.isfile(pt):
        import torch

        return {k: v.float().numpy() for k, v in torch.load(pt, map_location="cpu", weights_only=True).items()}
    raise ValueError(f"{path}: neither model.safetensors nor pytorch_model.bin found")


def load_hf_checkpoint(path: str) -> Tuple[EncoderConfig, Dict[str, np.ndarray], dict]:
    """(config, weights under the fairseq names, preprocessing {"do_normalize", "norm_eps"}) from a local HF directory."""
    if str(path).startswith("http") or not os.path.isdir(path):
        raise RuntimeError(f"{path}: no network in this build — pass a local Hugging Face checkpoint directory "
                           f"(config.json + model.safetensors)")
    hc = json.load(open(os.path.join(path, "config.json")))
    mt = hc.get("model_type")
    if mt not in ("hubert", "wav2vec2"):
        raise ValueError(f"{path}: model_type {mt!r} is not one of the architectures on this path (hubert, wav2vec2)")
    for key, want in (("hidden_act", "gelu"), ("feat_extract_activation", "gelu")):
        if hc.get(key, "gelu") != want:
            raise ValueError(f"{path}: {key}={hc.get(key)!r}; only 'gelu' is on the hot path")
    if abs(float(hc.get("layer_norm_eps", 1e-5)) - 1e-5) > 1e-12:
        raise ValueError(f"{path}: layer_norm_eps {hc.get('layer_norm_eps')} (the kernels use 1e-5)")
    if hc.get("add_adapter", False) or hc.get("position_embeddings_type", None) not in (None, "conv"):
        raise ValueError(f"{path}: adapters / non-convolutional position embeddings are not on this path")
    convs = list(zip(hc["conv_dim"], hc["conv_kernel"], hc["conv_stride"]))
    cfg = EncoderConfig(
        family="wav2vec2",  # HF derives the frame mask with the conv-length rule for every architecture
        conv_layers=[(int(c), int(k), int(s)) for c, k, s in convs],
        extractor_mode="layer_norm" if hc.get("feat_extract_norm", "group") == "layer" else "default",
        conv_bias=bool(hc.get("conv_bias", False)), encoder_layers=int(hc["num_hidden_layers"]),
        encoder_embed_dim=int(hc["hidden_size"]), encoder_ffn_embed_dim=int(hc["intermediate_size"]),
        encoder_attention_heads=int(hc["num_attention_heads"]), layer_norm_first=bool(hc.get("do_stable_layer_norm", False)),
        conv_pos=int(hc.get("num_conv_pos_embeddings", 128)), conv_pos_groups=int(hc.get("num_conv_pos_embedding_groups", 16)),
        feature_layer_norm=bool(hc.get("feat_proj_layer_norm", True)) if mt == "hubert" else True,
    )
    # without a preprocessor_config.json the reference's hf_hubert expert falls back to the facebook/hubert-base-ls960
    # feature extractor (do_normalize=False, hf_hubert/expert.py:20-27), whatever the model's extractor mode; wav2vec2
    # checkpoints ship the file (hf_wav2vec2 has no fallback), so the extractor mode decides there
    pre = {"do_normalize": cfg.extractor_mode == "layer_norm" and mt != "hubert", "norm_eps": 1e-7}
    pp = os.path.join(path, "preprocessor_config.json")
    if os.path.isfile(pp):
        pre["do_normalize"] = bool(json.load(open(pp)).get("do_normalize", True))
    cfg.normalize = pre["do_normalize"]
    cfg.wav_norm_eps = pre["norm_eps"]
    cfg.validate()

    sd = _load_state(path)
    sd = {(k[len(mt) + 1:] if k.startswith(mt + ".") else k): v for k, v in sd.items()}  # *ForCTC etc. prefix the base model

    def get(name):
        if name not in sd:
            raise ValueError(f"{path}: missing parameter {name}")
        return np.ascontiguousarray(sd[name], dtype=np.float32)

    w: Dict[str, np.ndarray] = {}
    for i in range(len(cfg.conv_layers)):
        q, p = f"feature_extractor.conv_layers.{i}", f"feature_extractor.conv_layers.{i}"
        w[f"{p}.0.weight"] = get(f"{q}.conv.weight")
        if cfg.conv_bias:
            w[f"{p}.0.bias"] = get(f"{q}.conv.bias")
        if cfg.extractor_mode == "layer_norm":
            w[f"{p}.2.1.weight"], w[f"{p}.2.1.bias"] = get(f"{q}.layer_norm.weight"), get(f"{q}.layer_norm.bias")
        elif i == 0:
            w[f"{p}.2.weight"], w[f"{p}.2.bias"] = get(f"{q}.layer_norm.weight"), get(f"{q}.layer_norm.bias")
    if cfg.feature_layer_norm:
        w["layer_norm.weight"], w["layer_norm.bias"] = get("feature_projection.layer_norm.weight"), get("feature_projection.layer_norm.bias")
    w["post_extract_proj.weight"], w["post_extract_proj.bias"] = get("feature_projection.projection.weight"), get("feature_projection.projection.bias")
    pc = "encoder.pos_conv_embed.conv"
    w["encoder.pos_conv.0.bias"] = get(f"{pc}.bias")
    g_name = f"{pc}.parametrizations.weight.original0" if f"{pc}.parametrizations.weight.original0" in sd else f"{pc}.weight_g"
    v_name = f"{pc}.parametrizations.weight.original1" if f"{pc}.parametrizations.weight.original1" in sd else f"{pc}.weight_v"
    w["encoder.pos_conv.0.weight_g"], w["encoder.pos_conv.0.weight_v"] = get(g_name).reshape(1, 1, -1), get(v_name)
    w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"] = get("encoder.layer_norm.weight"), get("encoder.layer_norm.bias")
    for l in range(cfg.encoder_layers):
        q, p = f"encoder.layers.{l}", f"encoder.layers.{l}"
        for n in ("q", "k", "v", "out"):
            w[f"{p}.self_attn.{n}_proj.weight"] = get(f"{q}.attention.{n}_proj.weight")
            w[f"{p}.self_attn.{n}_proj.bias"] = get(f"{q}.attention.{n}_proj.bias")
        w[f"{p}.self_attn_layer_norm.weight"], w[f"{p}.self_attn_layer_norm.bias"] = get(f"{q}.layer_norm.weight"), get(f"{q}.layer_norm.bias")
        w[f"{p}.fc1.weight"], w[f"{p}.fc1.bias"] = get(f"{q}.feed_forward.intermediate_dense.weight"), get(f"{q}.feed_forward.intermediate_dense.bias")
        w[f"{p}.fc2.weight"], w[f"{p}.fc2.bias"] = get(f"{q}.feed_forward.output_dense.weight"), get(f"{q}.feed_forward.output_dense.bias")
        w[f"{p}.final_layer_norm.weight"], w[f"{p}.final_layer_norm.bias"] = get(f"{q}.final_layer_norm.weight"), get(f"{q}.final_layer_norm.bias")
    from .synth import param_shapes

    for name, shape in param_shapes(cfg).items():
        if tuple(w[name].shape) != tuple(shape):
            raise ValueError(f"{path}: parameter {name} has shape {tuple(w[name].shape)}, expected {tuple(shape)}")
    return cfg, w, pre
