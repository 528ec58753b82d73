"""CPU restatement (numpy) of the reference multi-resolution HuBERT upstream forward.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s checker legs may import this file; nothing under
``s3prl_amd/`` does, and the product path has no CPU fallback.

Parity status: **pinned** against outputs of the reference's own ``multires_hubert`` expert (PyTorch CPU, imported from
``/root/reference`` in the build container): ``tests/golden/make_golden.py`` wrote ``tests/golden/tiny_multires*.npz``,
``tests/test_oracle_golden.py`` checks this restatement against them.

Follows ``s3prl/upstream/multires_hubert/expert.py:30-126`` -> ``MultiresHubertModel.forward(features_only=True)``
(``hubert_model.py:738-852``) -> ``TransformerEncoder`` (``wav2vec2_model.py:3046-3121``) and the conv adapters
(``hubert_model.py:970-1266``).  The hook capture of the expert becomes the returned list: per block its layer inputs and
its output, each repeated to the finest frame rate and cut to the common length.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

from .encoder_oracle import EPS, encoder_layer, feature_extractor, gelu, layer_norm, pos_conv, wav_normalize

RESIDUAL_SCALE = math.sqrt(0.4)  # hubert_model.py:429,1036


def group_norm_1(x_btc: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``Fp32GroupNorm(1, C)`` (wav2vec_model.py:46-56): ONE group, statistics over all channels and all frames of an
    utterance (zero-padded and to-be-discarded tail frames included)."""
    mu = x_btc.mean(axis=(1, 2), keepdims=True)
    xc = x_btc - mu
    var = (xc * xc).mean(axis=(1, 2), keepdims=True)
    return (xc / np.sqrt(var + EPS) * w + b).astype(x_btc.dtype)


def conv1d_same(x_btc: np.ndarray, w_oik: np.ndarray, stride: int) -> np.ndarray:
    """``nn.Conv1d(C, C, k, stride, padding=(k-1)//2, bias=False)`` on channel-last data (hubert_model.py:990-1000)."""
    B, T, C = x_btc.shape
    k = w_oik.shape[2]
    p = (k - 1) // 2
    xp = np.zeros((B, T + 2 * p, C), dtype=x_btc.dtype)
    xp[:, p:p + T] = x_btc
    L = (T + 2 * p - k) // stride + 1
    out = np.zeros((B, L, w_oik.shape[0]), dtype=x_btc.dtype)
    for j in range(k):
        out += xp[:, j:j + (L - 1) * stride + 1:stride] @ w_oik[:, :, j].T
    return out


def conv_transpose1d(x_btc: np.ndarray, w_iok: np.ndarray, stride: int) -> np.ndarray:
    """``nn.ConvTranspose1d(C, C, k, stride, padding=0, output_padding=stride-1, bias=False)`` (hubert_model.py:1008-1018):
    out[t*stride + j] += x[t] @ w[:, :, j]; length (T-1)*stride + k + stride - 1."""
    B, T, C = x_btc.shape
    k = w_iok.shape[2]
    L = (T - 1) * stride + k + stride - 1
    out = np.zeros((B, L, w_iok.shape[1]), dtype=x_btc.dtype)
    for j in range(k):
        out[:, j:j + (T - 1) * stride + 1:stride] += x_btc @ w_iok[:, :, j]
    return out


def conv_adapter(W: Dict[str, np.ndarray], mod: str, x: np.ndarray, up: int, down: int, kind: str) -> np.ndarray:
    """``ConvAdapter.forward`` (kind "full", hubert_model.py:1038-1078), ``ConvDownsampler.forward`` ("down",
    :1146-1167), ``ConvUpsampler.forward`` ("up", :1232-1250) on channel-last (B, T, C)."""
    sc = x.dtype.type(RESIDUAL_SCALE)
    r_up = None
    if kind in ("full", "up"):
        p = f"{mod}.upsample_conv"
        y = gelu(group_norm_1(conv_transpose1d(x, W[f"{p}.0.weight"], up), W[f"{p}.2.weight"], W[f"{p}.2.bias"]))
        r_up = np.repeat(x, up, axis=1)
        n = min(y.shape[1], r_up.shape[1])
        x = (y[:, :n] + r_up[:, :n]) * sc
    if kind in ("full", "down"):
        p = f"{mod}.downsample_conv"
        y = gelu(group_norm_1(conv1d_same(x, W[f"{p}.0.weight"], down), W[f"{p}.2.weight"], W[f"{p}.2.bias"]))
        r = x[:, ::down]
        n = min(y.shape[1], r.shape[1])
        x = (y[:, :n] + r[:, :n]) * sc
        if kind == "full":  # highway
            r = r_up[:, ::down]
            n = min(x.shape[1], r.shape[1])
            x = (x[:, :n] + r[:, :n]) * sc
    return x.astype(sc.dtype)


def transformer_encoder(cfg, W, prefix: str, n_layers: int, x: np.ndarray, valid, with_pos_conv: bool, states: list,
                        factor: int) -> np.ndarray:
    """``TransformerEncoder.forward`` (wav2vec2_model.py:3046-3121) with ``skip_pos_conv`` / ``override_encoder_layer``
    (:2986-3040).  Zeroes the padded frames of ``x`` IN PLACE like ``index_put`` (:3061-3062).  Appends the hooked
    tensors — every layer's input, then the encoder output — to ``states`` as (tensor, factor)."""
    B, T, _ = x.shape
    for b in range(B):
        x[b, valid[b]:] = 0
    h = x
    if with_pos_conv:
        h = h + pos_conv(cfg, W, h, prefix=f"{prefix}.pos_conv")
    if not cfg.layer_norm_first:
        h = layer_norm(h, W[f"{prefix}.layer_norm.weight"], W[f"{prefix}.layer_norm.bias"])
    for l in range(n_layers):
        # (the reference pads T to a multiple of 2 with a masked zero frame, :3072-3082: it changes no kept value, and the
        # expert cuts every state to the common length)
        states.append((h, factor))
        h = encoder_layer(cfg, W, l, h, valid, None, prefix=prefix)
    if cfg.layer_norm_first:
        h = layer_norm(h, W[f"{prefix}.layer_norm.weight"], W[f"{prefix}.layer_norm.bias"])
    states.append((h, factor))
    return h


def forward(cfg, weights: Dict[str, np.ndarray], wavs: List[np.ndarray], dtype=np.float32,
            n_max: Optional[int] = None) -> List[np.ndarray]:
    """``UpstreamExpert.__call__(wavs)["hidden_states"]`` of ``upstream/multires_hubert``."""
    dt = np.dtype(dtype)
    W = {k: v.astype(dt) for k, v in weights.items()}
    lens = [int(len(w)) for w in wavs]
    if n_max is None:
        n_max = max(lens)
    B = len(wavs)
    padded = np.zeros((B, n_max), dtype=dt)
    for b, w in enumerate(wavs):
        w = w.astype(dt)
        if cfg.normalize:  # multires_hubert/expert.py:107-108
            w = wav_normalize(w)
        padded[b, :lens[b]] = w
    x = feature_extractor(cfg, W, padded)
    T0 = x.shape[1]
    x = layer_norm(x, W["layer_norm.weight"], W["layer_norm.bias"])  # hubert_model.py:753-754
    x = x @ W["post_extract_proj.weight"].T + W["post_extract_proj.bias"]  # :760-761
    valid = [cfg.valid_frames(n, n_max) for n in lens]  # forward_padding_mask, :726-736

    blocks, T_out = cfg.multires_plan(T0)
    R = len(cfg.rate_pairs) + 1
    states: list = []
    residuals = []

    def adapt(x, valid, ad):
        kind, up, down, mod = ad
        y = conv_adapter(W, mod, x, up, down, kind)
        # padding mask: repeat_interleave(up)[::down][:T'] (:1080-1084,1169-1172,1256-1258)
        v = [min(-(-(v * (up if kind != "down" else 1)) // (down if kind != "up" else 1)), y.shape[1]) for v in valid]
        return y, v

    for bi, blk in enumerate(blocks):
        if blk["adapter"] is not None:
            x, valid = adapt(x, valid, blk["adapter"])
        assert x.shape[1] == blk["T"], (x.shape, blk)
        x_in = x
        h = transformer_encoder(cfg, W, blk["prefix"], blk["layers"], x_in, valid, bi == 0, states, blk["factor"])
        if bi < R - 1:
            residuals.append(h)  # :795
            x = h
        elif bi == R - 1:
            x = x_in + h  # :801-802 (x_in: padded frames zeroed in place by the encoder)
            residuals.reverse()
        else:
            r = residuals[bi - R]  # align_size_sum, :777-783,816
            c = min(h.shape[1], r.shape[1])
            x = h[:, :c] + r[:, :c]
            valid = [min(v, c) for v in valid]
    out = []
    for h, f in states:  # multires_hubert/expert.py:26-27,93-101
        out.append(np.ascontiguousarray(np.repeat(h, f, axis=1)[:, :T_out]).astype(dt))
    return out
