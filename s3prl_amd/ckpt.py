"""Readers for the three converted-checkpoint formats the reference experts load:

* HuBERT   ``{"task_cfg", "model_cfg", "model_weight", "dictionaries_symbols"}``  (hubert/convert.py:37-56)
* wav2vec2 ``{"task_cfg", "model_cfg", "model_weight"}``                          (wav2vec2/convert.py:26-39)
* WavLM    ``{"cfg", "model"}``                                                    (wavlm/expert.py:37-40)

Only the hot-path tensors are kept (``mask_emb``, ``label_embs_concat``, ``final_proj``, ``quantizer.*``,
``project_q`` … are unused at inference).  Also writes the same formats from synthetic weights.
"""

from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .config import EncoderConfig, config_from_dicts
from .synth import param_shapes

_REQUIRED = {
    "hubert": ["task_cfg", "model_cfg", "model_weight", "dictionaries_symbols"],
    "wav2vec2": ["task_cfg", "model_cfg", "model_weight"],
    "wavlm": ["cfg", "model"],
}


def _plain(d):
    """dataclass / namespace / omegaconf-ish → plain dict."""
    if isinstance(d, dict):
        return d
    import dataclasses

    if dataclasses.is_dataclass(d):
        return dataclasses.asdict(d)
    return dict(vars(d))


def load_checkpoint(ckpt: str, family: str) -> Tuple[EncoderConfig, Dict[str, np.ndarray]]:
    import torch

    state = torch.load(ckpt, map_location="cpu", weights_only=False)
    for key in _REQUIRED[family]:
        if key not in state:
            # same message shape as hubert/convert.py:46-49
            raise ValueError(f"{ckpt} is not a valid checkpoint since the required key: {key} is missing")
    if family == "wavlm":
        cfg = config_from_dicts("wavlm", _plain(state["cfg"]))
        sd = state["model"]
    else:
        cfg = config_from_dicts(family, _plain(state["model_cfg"]), _plain(state["task_cfg"]))
        sd = state["model_weight"]
    weights = {}
    for name, shape in param_shapes(cfg).items():
        if name not in sd:
            raise ValueError(f"{ckpt}: missing parameter {name}")
        w = sd[name]
        w = w.detach().cpu().float().numpy() if hasattr(w, "detach") else np.asarray(w, dtype=np.float32)
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{ckpt}: parameter {name} has shape {tuple(w.shape)}, expected {tuple(shape)}")
        weights[name] = np.ascontiguousarray(w)
    return cfg, weights


def save_checkpoint(path: str, cfg: EncoderConfig, weights: Dict[str, np.ndarray]) -> None:
    """Write ``weights`` in the reference's converted format for ``cfg.family`` (what ``*_local(ckpt=...)`` reads)."""
    import torch

    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    model_cfg = dict(
        extractor_mode=cfg.extractor_mode, conv_bias=cfg.conv_bias, encoder_layers=cfg.encoder_layers,
        encoder_embed_dim=cfg.encoder_embed_dim, encoder_ffn_embed_dim=cfg.encoder_ffn_embed_dim,
        encoder_attention_heads=cfg.encoder_attention_heads, layer_norm_first=cfg.layer_norm_first,
        conv_pos=cfg.conv_pos, conv_pos_groups=cfg.conv_pos_groups, activation_fn="gelu",
        conv_feature_layers=str([tuple(t) for t in cfg.conv_layers]),
    )
    if cfg.family == "wavlm":
        model_cfg.update(normalize=cfg.normalize, relative_position_embedding=cfg.relative_position_embedding,
                         num_buckets=cfg.num_buckets, max_distance=cfg.max_distance, gru_rel_pos=cfg.gru_rel_pos)
        torch.save({"cfg": model_cfg, "model": sd}, path)
    elif cfg.family == "hubert":
        torch.save({"task_cfg": {"normalize": cfg.normalize, "label_rate": 50.0}, "model_cfg": model_cfg,
                    "model_weight": sd, "dictionaries_symbols": [["a"] * 8]}, path)
    else:
        torch.save({"task_cfg": {"normalize": cfg.normalize}, "model_cfg": model_cfg, "model_weight": sd}, path)
