"""Deterministic synthetic checkpoints and waveforms.

There is no network for real checkpoints, so benchmarks, tests and the golden
fixtures all use weights drawn from a seeded numpy generator, named exactly like the reference
``state_dict`` entries on the hot path, so that the very same tensors can be loaded
into the reference classes (``tests/golden/make_golden.py``) and into the HIP encoder.

The scales are chosen so activations stay O(1) through 12-24 layers and so that every bias /
affine parameter is exercised (none is left at its 0/1 default).
"""

from __future__ import annotations

from typing import Dict, List

import numpy as np

from .config import EncoderConfig


def param_shapes(cfg: EncoderConfig) -> Dict[str, tuple]:
    """Hot-path parameter names → shapes (reference naming)."""
    s: Dict[str, tuple] = {}
    cin = 1
    for i, (dim, k, _) in enumerate(cfg.conv_layers):
        p = f"feature_extractor.conv_layers.{i}"
        s[f"{p}.0.weight"] = (dim, cin, k)
        if cfg.conv_bias:
            s[f"{p}.0.bias"] = (dim,)
        if cfg.extractor_mode == "layer_norm":
            s[f"{p}.2.1.weight"] = (dim,)
            s[f"{p}.2.1.bias"] = (dim,)
        elif i == 0:
            s[f"{p}.2.weight"] = (dim,)
            s[f"{p}.2.bias"] = (dim,)
        cin = dim
    C, D, F, H = cfg.conv_dim, cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.encoder_attention_heads
    if cfg.feature_layer_norm:
        s["layer_norm.weight"] = (C,)
        s["layer_norm.bias"] = (C,)
    s["post_extract_proj.weight"] = (D, C)
    s["post_extract_proj.bias"] = (D,)
    if cfg.family == "multires_hubert":
        return _multires_shapes(cfg, s)
    if cfg.pos_conv_depth > 1:  # data2vec: plain convs, no weight_norm (wav2vec2_model.py:3001-3007)
        for i in range(cfg.pos_conv_depth):
            s[f"encoder.pos_conv.{i}.0.weight"] = (D, D // cfg.conv_pos_groups, cfg.pos_conv_kernel)
            s[f"encoder.pos_conv.{i}.0.bias"] = (D,)
    else:
        s["encoder.pos_conv.0.bias"] = (D,)
        s["encoder.pos_conv.0.weight_g"] = (1, 1, cfg.conv_pos)
        s["encoder.pos_conv.0.weight_v"] = (D, D // cfg.conv_pos_groups, cfg.conv_pos)
    s["encoder.layer_norm.weight"] = (D,)
    s["encoder.layer_norm.bias"] = (D,)
    for l in range(cfg.encoder_layers):
        p = f"encoder.layers.{l}"
        _layer_shapes(s, p, D, F)
        if cfg.family == "wavlm":
            if cfg.relative_position_embedding and l == 0:
                s[f"{p}.self_attn.relative_attention_bias.weight"] = (cfg.num_buckets, H)
            if cfg.gru_rel_pos:
                s[f"{p}.self_attn.grep_linear.weight"] = (8, cfg.head_dim)
                s[f"{p}.self_attn.grep_linear.bias"] = (8,)
                s[f"{p}.self_attn.grep_a"] = (1, H, 1, 1)
    if cfg.pred_heads:  # distiller/model.py:155-161: Linear(D, D*N) -> GELU -> SplitLinear(D, N, D)
        N = cfg.pred_heads
        s["output_layer.0.weight"] = (D * N, D)
        s["output_layer.0.bias"] = (D * N,)
        s["output_layer.2.weight"] = (N, D, D)
        s["output_layer.2.bias"] = (1, 1, N, D)
    return s


def _layer_shapes(s: Dict[str, tuple], p: str, D: int, F: int) -> None:
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
        s[f"{p}.self_attn.{n}.weight"] = (D, D)
        s[f"{p}.self_attn.{n}.bias"] = (D,)
    s[f"{p}.self_attn_layer_norm.weight"] = (D,)
    s[f"{p}.self_attn_layer_norm.bias"] = (D,)
    s[f"{p}.fc1.weight"] = (F, D)
    s[f"{p}.fc1.bias"] = (F,)
    s[f"{p}.fc2.weight"] = (D, F)
    s[f"{p}.fc2.bias"] = (D,)
    s[f"{p}.final_layer_norm.weight"] = (D,)
    s[f"{p}.final_layer_norm.bias"] = (D,)


def _multires_shapes(cfg: EncoderConfig, s: Dict[str, tuple]) -> Dict[str, tuple]:
    """multires-HuBERT (hubert_model.py:337-530): encoders.{i} / middle_encoder / decoders.{i} TransformerEncoders (only
    encoders.0 keeps its positional conv, :399-403,434-449) and the conv adapters between them."""
    D, F, k = cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.conv_adapter_kernel
    R = len(cfg.rate_pairs) + 1
    names = [f"encoders.{i}" for i in range(R - 1)] + ["middle_encoder"] + [f"decoders.{i}" for i in range(R - 1)]
    for bi, name in enumerate(names):
        if bi == 0:
            s[f"{name}.pos_conv.0.bias"] = (D,)
            s[f"{name}.pos_conv.0.weight_g"] = (1, 1, cfg.conv_pos)
            s[f"{name}.pos_conv.0.weight_v"] = (D, D // cfg.conv_pos_groups, cfg.conv_pos)
        s[f"{name}.layer_norm.weight"] = (D,)
        s[f"{name}.layer_norm.bias"] = (D,)
        for l in range(cfg.block_layers[bi]):
            _layer_shapes(s, f"{name}.layers.{l}", D, F)
    for i in range(R - 1):
        for mod, convs in ((f"downsample_modules.{i}", ("downsample_conv",) if cfg.use_plain_updownsample
                            else ("upsample_conv", "downsample_conv")),
                           (f"upsample_modules.{i}", ("upsample_conv",) if cfg.use_plain_updownsample
                            else ("upsample_conv", "downsample_conv"))):
            for cv in convs:
                s[f"{mod}.{cv}.0.weight"] = (D, D, k)  # Conv1d (out, in, k) / ConvTranspose1d (in, out, k)
                s[f"{mod}.{cv}.2.weight"] = (D,)       # Fp32GroupNorm(1, D)
                s[f"{mod}.{cv}.2.bias"] = (D,)
    return s


PROFILES = ("synthetic", "pretrained_like")


def synth_weights(cfg: EncoderConfig, seed: int = 0, profile: str = "synthetic") -> Dict[str, np.ndarray]:
    """Seeded fp32 weights for every hot-path parameter.

    ``profile="synthetic"``: gaussian weights scaled so activations stay O(1) (every fixture of rounds 1-3).
    ``profile="pretrained_like"``: the same draw reshaped to the statistics released wav2vec 2.0 / HuBERT / WavLM checkpoints
    show and O(1) gaussians do not (``_pretrained_like``): outlier channels in the residual stream, LayerNorm gains of 2-4
    on a few channels, heavy-tailed (Student-t) matrices, key / query biases that shift every score of a row, a near-silent
    and a near-constant conv0 channel, a wide relative-position table.  What the reference's own regression test runs on
    (test/test_upstream.py:118-136 loads released checkpoints; there is no network here)."""
    if profile not in PROFILES:
        raise ValueError(f"profile must be one of {PROFILES}, got {profile!r}")
    out = _synthetic(cfg, seed)
    return _pretrained_like(cfg, out, seed) if profile == "pretrained_like" else out


def _pretrained_like(cfg: EncoderConfig, w: Dict[str, np.ndarray], seed: int) -> Dict[str, np.ndarray]:
    """Reshape an O(1) draw into released-checkpoint statistics.  Every choice is a function of (cfg, seed) only."""
    rng = np.random.default_rng([seed, 0x5EED])
    D, C = cfg.encoder_embed_dim, cfg.conv_dim
    hot = rng.choice(D, size=4, replace=False)      # the residual stream's outlier channels (the same in every layer)
    warm = rng.choice(D, size=8, replace=False)     # channels with large LayerNorm gains only
    pre_ln = bool(cfg.layer_norm_first)

    def student_t(shape, df=4.0):  # unit variance, tails ~ |x|^-5: a released matrix has entries at 8-15 sigma
        return rng.standard_t(df, size=shape) * np.sqrt((df - 2.0) / df)

    out = {}
    for name, v in w.items():
        v = v.astype(np.float64)
        leaf = name.rsplit(".", 1)[-1]
        is_linear = leaf == "weight" and v.ndim == 2 and "relative_attention_bias" not in name and "grep" not in name
        if is_linear:
            v = student_t(v.shape) * v.std()
            if name.endswith((".fc2.weight", ".out_proj.weight")):
                # writers of the residual stream: the outlier channels receive 30-100x the typical update (pre-LN models
                # accumulate them over 24 layers to the "massive activations" of released large checkpoints) ...
                v[hot] *= rng.uniform(30.0, 100.0, size=(len(hot), 1)) if pre_ln else rng.uniform(8.0, 20.0, size=(len(hot), 1))
            if name.endswith((".fc1.weight", "q_proj.weight", "k_proj.weight", "v_proj.weight")):
                # ... and its readers have learned small weights on them and on the high-gain channels (otherwise every
                # logit saturates and fp32 itself is 1e-3 away from an fp64 evaluation: the "ref vs fp64" column of
                # tools/parity_table.py / profiles/r04_parity.md is the check that a profile stays well-conditioned)
                v[:, hot] *= 0.05 if pre_ln else 0.1
                v[:, warm] *= 0.3
        elif "conv_layers" in name and leaf == "weight" and v.ndim == 3:
            if name.startswith("feature_extractor.conv_layers.0."):
                v[1] *= 1e-4                                       # a near-silent filter: variance far below GroupNorm's eps
                v[2] = 0.3 * np.abs(v[2]).mean() + 1e-3 * v[2]     # a near-constant one (local average): mean >> deviation
                v[3:8] *= rng.uniform(5.0, 30.0, size=(5, 1, 1))   # and a few loud ones
            else:
                v = student_t(v.shape) * v.std()
        elif leaf == "weight" and v.ndim == 1:  # LayerNorm / GroupNorm gains
            if v.shape[0] == D:
                v[warm] *= rng.uniform(2.0, 4.0, size=len(warm))
                v[hot] *= (0.1 if pre_ln else 3.0)  # pre-LN: the outlier is squashed on read; post-LN: written by the gain
            else:
                idx = rng.choice(v.shape[0], size=6, replace=False)
                v[idx] *= rng.uniform(3.0, 8.0, size=6)
        elif leaf == "bias":
            if name.endswith(("k_proj.bias", "q_proj.bias")):
                v = v * 20.0  # std 1: q . b_k shifts all scores of a query by tens (softmax-invariant, unbounded in training)
            elif name.endswith((".fc2.bias", ".out_proj.bias")):
                v[hot] += rng.choice([-1.0, 1.0], size=len(hot)) * (rng.uniform(2.0, 6.0, size=len(hot)) if pre_ln else 1.0)
            elif v.shape[0] == D and "layer_norm" in name:
                v[hot] += rng.choice([-1.0, 1.0], size=len(hot)) * rng.uniform(1.0, 3.0, size=len(hot))
        elif name.endswith("relative_attention_bias.weight"):
            v = student_t(v.shape, 3.0) * 2.0  # released tables span roughly +-10
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def outlier_channels(cfg: EncoderConfig, seed: int) -> np.ndarray:
    """The residual-stream outlier channels ``_pretrained_like`` picks for (cfg, seed) (its first draw)."""
    return np.random.default_rng([seed, 0x5EED]).choice(cfg.encoder_embed_dim, size=4, replace=False)


def scale_outlier_writers(cfg: EncoderConfig, weights: Dict[str, np.ndarray], seed: int, factor: float) -> Dict[str, np.ndarray]:
    """A copy of pretrained-like ``weights`` whose residual-stream writers (rows of fc2 / out_proj and their biases on the
    outlier channels) are ``factor`` times louder: the knob of tools/fp16_cliff.py, which looks for the activation scale at
    which each 16-bit mode leaves its tolerance or its number range."""
    hot = outlier_channels(cfg, seed)
    out = dict(weights)
    for name, v in weights.items():
        if name.endswith((".fc2.weight", ".out_proj.weight", ".fc2.bias", ".out_proj.bias")):
            v = np.array(v, dtype=np.float32, copy=True)
            v[hot] *= np.float32(factor)
            out[name] = v
    return out


def _synthetic(cfg: EncoderConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("relative_attention_bias.weight"):
            w = rng.standard_normal(shape) * 0.5
        elif name.endswith("grep_a"):
            w = 1.0 + 0.3 * rng.standard_normal(shape)
        elif name.endswith("weight_g"):
            # weight_norm gain: w[:,:,k] = g[k] * v[:,:,k] / ||v[:,:,k]||  → RMS(w) = g/sqrt(numel per tap)
            d_out, d_in, _ = param_shapes(cfg)[name[:-1] + "v"]
            w = (1.0 + 0.2 * rng.standard_normal(shape)) * np.sqrt(d_out / shape[-1]) * 0.5
        elif name == "output_layer.2.weight":  # SplitLinear (N, Din, Dout)
            w = rng.standard_normal(shape) * np.sqrt(1.0 / shape[1])
        elif leaf == "weight" and len(shape) == 1:  # norm gains
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif leaf == "bias":
            w = 0.05 * rng.standard_normal(shape)
        elif "pos_conv" in name and leaf == "weight":  # data2vec conv block (out, in/g, k)
            w = rng.standard_normal(shape) * np.sqrt(2.0 / (shape[1] * shape[2]))
        elif "sample_conv.0" in name:  # conv adapters (D, D, k): unit-variance output before the GroupNorm
            w = rng.standard_normal(shape) * np.sqrt(1.0 / (shape[1] * shape[2]))
        elif "conv_layers" in name:  # (out, in, k): keep unit variance through GELU (gain ~ sqrt(2.5))
            fan_in = shape[1] * shape[2]
            w = rng.standard_normal(shape) * np.sqrt(2.5 / fan_in)
        elif leaf == "weight_v":
            w = rng.standard_normal(shape)
        else:  # linear (out, in)
            gain = 2.0 if ".fc2." in name else 1.0
            w = rng.standard_normal(shape) * np.sqrt(gain / shape[-1])
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def synth_wavs(lengths: List[int], seed: int = 1234, dc: float = 0.0, scale: float = 1.0) -> List[np.ndarray]:
    """Seeded gaussian 'waveforms' (the reference's own convention: ``torch.randn``,
    s3prl/util/pseudo_data.py:70). numpy-generated so both sides can rebuild them offline."""
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal(n) * scale + dc).astype(np.float32) for n in lengths]


def pseudo_lengths(n: int = 2, min_secs: float = 1.0, max_secs: float = 3.0, seed: int = 0,
                   sample_rate: int = 16000) -> List[int]:
    """Lengths in the style of ``get_pseudo_wavs`` (s3prl/util/pseudo_data.py:52-77): n utterances
    between min_secs and max_secs."""
    rng = np.random.default_rng(seed)
    lo, hi = int(min_secs * sample_rate), int(max_secs * sample_rate)
    return [int(rng.integers(lo, hi + 1)) for _ in range(n)]


# ------------------------------------------------------------------------------------------
# named configurations
# ------------------------------------------------------------------------------------------

def named_config(name: str) -> EncoderConfig:
    """Architectures named by BASELINE.json plus tiny variants of each for fast tests.

    Large-model hyper-parameters are not in the reference tree (they live in the released
    checkpoints); the values follow SURVEY §8c.
    """
    tiny_conv = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2
    tiny = dict(conv_layers=tiny_conv, encoder_layers=3, encoder_embed_dim=128,
                encoder_ffn_embed_dim=256, encoder_attention_heads=2, conv_pos=16, conv_pos_groups=4)
    large = dict(extractor_mode="layer_norm", encoder_layers=24, encoder_embed_dim=1024,
                 encoder_ffn_embed_dim=4096, encoder_attention_heads=16, layer_norm_first=True,
                 normalize=True)
    table = {
        "hubert_base": dict(family="hubert"),
        "wav2vec2_base": dict(family="wav2vec2"),
        "wavlm_base": dict(family="wavlm"),
        "wavlm_base_plus": dict(family="wavlm", relative_position_embedding=True, num_buckets=320,
                                max_distance=800, gru_rel_pos=True),
        "hubert_large": dict(family="hubert", conv_bias=False, **large),
        "wav2vec2_large": dict(family="wav2vec2", conv_bias=True, **large),
        "wavlm_large": dict(family="wavlm", conv_bias=False, relative_position_embedding=True,
                            num_buckets=320, max_distance=800, gru_rel_pos=True, **large),
        "distilhubert": dict(family="distiller", encoder_layers=2, feature_layer_norm=False, pred_heads=3),
        "tiny_distiller": dict(family="distiller", feature_layer_norm=False, pred_heads=3, **{**tiny, "encoder_layers": 2}),
        "tiny_wavlm_norel": dict(family="wavlm", **tiny),
        "data2vec_base": dict(family="wav2vec2", extractor_mode="layer_norm", conv_pos=95, pos_conv_depth=5, normalize=True),
        "tiny_data2vec": dict(family="wav2vec2", **{**tiny, "extractor_mode": "layer_norm", "conv_pos": 15,
                                                   "pos_conv_depth": 3, "normalize": True}),
        "tiny_hubert": dict(family="hubert", **tiny),
        # multires-HuBERT (mrhubert_mono_base: 20 ms -> 40 ms -> 20 ms, 4 layers each; the hyper-parameters are those of
        # the released checkpoint's config as far as the reference tree shows them: hubert_model.py:97-330 defaults)
        "multires_hubert_base": dict(family="multires_hubert", label_rate_ratios=[1, 2], block_layers=[4, 4, 4]),
        "tiny_multires": dict(family="multires_hubert", label_rate_ratios=[1, 2], block_layers=[2, 1, 2],
                              **{**tiny, "encoder_layers": 5}),
        "tiny_multires_large": dict(family="multires_hubert", label_rate_ratios=[1, 2], block_layers=[1, 2, 1],
                                    **{**tiny, "encoder_layers": 4, "extractor_mode": "layer_norm",
                                       "layer_norm_first": True, "normalize": True}),
        "tiny_multires3": dict(family="multires_hubert", label_rate_ratios=[1, 2, 1, 2], block_layers=[1, 1, 2, 1, 1],
                               **{**tiny, "encoder_layers": 6}),
        "tiny_multires_plain": dict(family="multires_hubert", label_rate_ratios=[1, 2], block_layers=[1, 1, 1],
                                    use_plain_updownsample=True, **{**tiny, "encoder_layers": 3}),
        "tiny_wav2vec2": dict(family="wav2vec2", **tiny),
        "tiny_hubert_large": dict(family="hubert", **{**tiny, "extractor_mode": "layer_norm",
                                                      "layer_norm_first": True, "normalize": True,
                                                      "conv_bias": True}),
        "tiny_wav2vec2_large": dict(family="wav2vec2", **{**tiny, "extractor_mode": "layer_norm",
                                                          "layer_norm_first": True, "normalize": True,
                                                          "conv_bias": True}),
        "tiny_wavlm": dict(family="wavlm", relative_position_embedding=True, num_buckets=32,
                           max_distance=64, gru_rel_pos=True, **tiny),
        "tiny_wavlm_large": dict(family="wavlm", relative_position_embedding=True, num_buckets=32,
                                 max_distance=64, gru_rel_pos=True,
                                 **{**tiny, "extractor_mode": "layer_norm", "layer_norm_first": True,
                                    "normalize": True}),
    }
    if name not in table:
        raise KeyError(f"unknown config {name!r}; have {sorted(table)}")
    cfg = EncoderConfig(**table[name])
    cfg.validate()
    return cfg
