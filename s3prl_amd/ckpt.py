"""Readers for the three converted-checkpoint formats the reference experts load:

* HuBERT   ``{"task_cfg", "model_cfg", "model_weight", "dictionaries_symbols"}``  (hubert/convert.py:37-56)
* wav2vec2 ``{"task_cfg", "model_cfg", "model_weight"}``                          (wav2vec2/convert.py:26-39)
* WavLM / UniSpeech-SAT ``{"cfg", "model"}``                      (wavlm/expert.py:37-40, unispeech_sat/expert.py:36-39)
* DistilHuBERT ``{"Config": {"distiller": {...}}, "Distiller": state_dict}``      (distiller/builder.py:41-58,119-122)
* multires-HuBERT: the HuBERT layout with ``MultiresHubertConfig`` in ``model_cfg``  (multires_hubert/convert.py:21-62)

plus the fairseq layout ``{"cfg": {"task", "model"}, "model"}`` that ``hubert_custom(fairseq=True)`` /
``wav2vec2_custom(fairseq=True)`` convert first (hubert/convert.py:22-40, wav2vec2/convert.py:14-24).

Only the hot-path tensors are kept (``mask_emb``, ``label_embs_concat``, ``final_proj``, ``quantizer.*``,
``project_q`` … are unused at inference).  Also writes the same formats from synthetic weights.
"""

from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .config import EncoderConfig, config_from_dicts, config_from_distiller, config_from_multires
from .synth import param_shapes

_REQUIRED = {
    "hubert": ["task_cfg", "model_cfg", "model_weight", "dictionaries_symbols"],
    "wav2vec2": ["task_cfg", "model_cfg", "model_weight"],
    "wavlm": ["cfg", "model"],
    "distiller": ["Config", "Distiller"],
    "multires_hubert": ["task_cfg", "model_cfg", "model_weight", "dictionaries_symbols"],
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
    elif family == "distiller":
        cfg = config_from_distiller(_plain(_plain(state["Config"])["distiller"]))
        sd = state["Distiller"]
    elif family == "multires_hubert":
        cfg = config_from_multires(_plain(state["model_cfg"]), _plain(state["task_cfg"]))
        sd = state["model_weight"]
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
        conv_pos=cfg.conv_pos, conv_pos_groups=cfg.conv_pos_groups, activation_fn="gelu", pos_conv_depth=cfg.pos_conv_depth,
        conv_feature_layers=str([tuple(t) for t in cfg.conv_layers]),
    )
    if cfg.family == "distiller":
        d = dict(extractor_mode=cfg.extractor_mode, extractor_conv_feature_layers=model_cfg["conv_feature_layers"],
                 conv_pos=cfg.conv_pos, conv_pos_groups=cfg.conv_pos_groups, encoder_layers=cfg.encoder_layers,
                 encoder_embed_dim=cfg.encoder_embed_dim, encoder_ffn_embed_dim=cfg.encoder_ffn_embed_dim,
                 encoder_attention_heads=cfg.encoder_attention_heads, layer_norm_first=cfg.layer_norm_first,
                 final_dim=cfg.encoder_embed_dim, n_tasks=cfg.pred_heads, task_emb_type="expand-last",
                 out_layer_type="expand-last")
        torch.save({"Config": {"distiller": d}, "Distiller": sd}, path)
    elif cfg.family == "multires_hubert":
        R = len(cfg.rate_pairs) + 1
        bl = cfg.block_layers
        model_cfg.update(label_rate_ratios=list(cfg.label_rate_ratios), encoder_layers=bl[0],
                         override_encoder_layers=str(bl[:R] + [bl[2 * R - 2 - i] for i in range(R - 1)]),
                         conv_adapator_kernal=cfg.conv_adapter_kernel, use_plain_updownsample=cfg.use_plain_updownsample)
        model_cfg.pop("pos_conv_depth")
        torch.save({"task_cfg": {"normalize": cfg.normalize, "label_rate": 50.0}, "model_cfg": model_cfg,
                    "model_weight": sd, "dictionaries_symbols": [["a"] * 8] * R}, path)
    elif cfg.family == "wavlm":
        model_cfg.update(normalize=cfg.normalize, relative_position_embedding=cfg.relative_position_embedding,
                         num_buckets=cfg.num_buckets, max_distance=cfg.max_distance, gru_rel_pos=cfg.gru_rel_pos)
        torch.save({"cfg": model_cfg, "model": sd}, path)
    elif cfg.family == "hubert":
        torch.save({"task_cfg": {"normalize": cfg.normalize, "label_rate": 50.0}, "model_cfg": model_cfg,
                    "model_weight": sd, "dictionaries_symbols": [["a"] * 8]}, path)
    else:
        torch.save({"task_cfg": {"normalize": cfg.normalize}, "model_cfg": model_cfg, "model_weight": sd}, path)


def convert_fairseq_checkpoint(path: str, family: str, refresh: bool = False) -> str:
    """``<stem>.converted.pt`` next to a fairseq checkpoint, in the reference's converted format — what
    ``load_and_convert_fairseq_ckpt`` writes (hubert/convert.py:22-40, wav2vec2/convert.py:14-24), without importing
    the ``fairseq`` package: the file is unpickled with torch and ``cfg`` may be a dict or an OmegaConf container
    (unpickling the latter needs ``omegaconf`` to be importable; if it is not, the error says so)."""
    import os

    import torch

    src = os.path.abspath(path)
    name = os.path.splitext(os.path.basename(src))[0] + ".converted.pt"
    out_dir = os.path.dirname(src)
    if not os.access(out_dir, os.W_OK):  # read-only checkpoint store: convert into a per-user cache instead
        import hashlib

        out_dir = os.path.join(os.environ.get("S3PRL_AMD_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "s3prl_amd")),
                               hashlib.sha1(src.encode()).hexdigest()[:12])
        os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name)

    def fresh():  # converted file present and not older than its source (refresh re-converts once, not once per rank)
        return os.path.isfile(out) and os.path.getmtime(out) >= os.path.getmtime(src)

    if fresh() and not refresh:
        return out
    # one converter at a time (every rank of a data-parallel job calls this at once): exclusive lock on a side file, then
    # re-check — the ranks that waited find the finished file (the reference serialises the same step with FileLock,
    # hubert/hubconf.py:46-56) — and the result appears atomically (temp file + os.replace), never half-written
    import fcntl
    import time

    t_call = time.time()
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if fresh() and (not refresh or os.path.getmtime(out) >= t_call - 1.0):
                return out
            try:
                state = torch.load(src, map_location="cpu", weights_only=False)
            except ModuleNotFoundError as e:  # pragma: no cover - depends on the pickled classes
                raise RuntimeError(f"{path}: unpickling this fairseq checkpoint needs the module {e.name!r}; convert it "
                                   f"with s3prl's upstream/{family}/convert.py where fairseq is installed") from e
            if "cfg" not in state or "model" not in state:
                raise ValueError(f"{path} is not a fairseq checkpoint (needs 'cfg' and 'model')")
            cfg = state["cfg"]
            if not isinstance(cfg, dict):
                from omegaconf import OmegaConf  # only reachable when omegaconf unpickled the object above

                cfg = OmegaConf.to_container(cfg)
            conv = {"task_cfg": _plain(cfg["task"]), "model_cfg": _plain(cfg["model"]), "model_weight": state["model"]}
            if family == "hubert":
                dicts = (state.get("task_state") or {}).get("dictionaries") or []
                conv["dictionaries_symbols"] = [list(getattr(d, "symbols", d)) for d in dicts]
            tmp = f"{out}.tmp.{os.getpid()}"
            try:
                torch.save(conv, tmp)
                os.replace(tmp, out)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out
