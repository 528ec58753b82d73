"""CPU restatement (numpy) of the reference upstream-encoder forward.  TEST INFRASTRUCTURE ONLY.

This file is the *checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``s3prl_amd/`` imports it, and the
product path has no CPU fallback.

Parity status: **pinned** against outputs of the reference itself (PyTorch CPU, imported from
``/root/reference`` in the build container) — ``tests/golden/make_golden.py`` generated the committed
fixtures ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this restatement against them.
(The reference's own golden vectors are hosted remotely, test/test_upstream.py:25-66, and are not
reachable offline.)

Every function cites the reference lines it restates (paths relative to the reference root).
All arithmetic is done in ``dtype`` (float32 like the reference CPU run, or float64 as a tighter truth).
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
from scipy.special import erf as _erf

EPS = 1e-5  # every LayerNorm / GroupNorm on the path uses the torch default eps


# ------------------------------------------------------------------------------------------------
# elementwise / normalisation pieces
# ------------------------------------------------------------------------------------------------

def gelu(x: np.ndarray) -> np.ndarray:
    """erf-GELU: ``nn.GELU()`` wav2vec2_model.py:2896-2906, ``F.gelu(x.float())`` :1899-1900."""
    return (0.5 * x * (1.0 + _erf(x * (1.0 / math.sqrt(2.0))))).astype(x.dtype)


def layer_norm(x: np.ndarray, w: Optional[np.ndarray], b: Optional[np.ndarray]) -> np.ndarray:
    """``F.layer_norm`` over the last axis, biased variance, eps 1e-5 (wav2vec2_model.py:1822-1838)."""
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    y = xc / np.sqrt(var + EPS)
    if w is not None:
        y = y * w + b
    return y.astype(x.dtype)


def wav_normalize(wav: np.ndarray, eps: float = EPS) -> np.ndarray:
    """``F.layer_norm(wav, wav.shape)`` per utterance before padding (hubert/expert.py:57-58, wavlm/expert.py:72-73,
    wav2vec2/expert.py:68); ``eps`` 1e-7 = Hugging Face's ``zero_mean_unit_var_norm`` (hf_hubert/expert.py:30-37)."""
    mu = wav.mean()
    xc = wav - mu
    return (xc / np.sqrt((xc * xc).mean() + eps)).astype(wav.dtype)


def group_norm_per_channel(x_btc: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``Fp32GroupNorm(C, C)``: one group per channel, statistics over ALL T frames including the
    zero-padded tail (wav2vec2_model.py:2902,1841-1853).  ``x_btc`` is channel-last (B, T, C)."""
    mu = x_btc.mean(1, keepdims=True)
    xc = x_btc - mu
    var = (xc * xc).mean(1, keepdims=True)
    return (xc / np.sqrt(var + EPS) * w + b).astype(x_btc.dtype)


# ------------------------------------------------------------------------------------------------
# conv feature extractor
# ------------------------------------------------------------------------------------------------

def conv1d_channel_last(x_btc: np.ndarray, w_oik: np.ndarray, bias: Optional[np.ndarray], stride: int) -> np.ndarray:
    """``nn.Conv1d(Cin, Cout, k, stride)`` (wav2vec2_model.py:2879) on channel-last data:
    y[b,t,co] = sum_{j,ci} w[co,ci,j] * x[b, stride*t + j, ci]."""
    B, L, Cin = x_btc.shape
    Cout, _, k = w_oik.shape
    Lout = (L - k) // stride + 1
    x_btc = np.ascontiguousarray(x_btc)
    it = x_btc.itemsize  # (explicit strides: numpy reports arbitrary strides for length-1 axes)
    win = np.lib.stride_tricks.as_strided(x_btc, shape=(B, Lout, k * Cin),
                                          strides=(L * Cin * it, stride * Cin * it, it))
    wmat = np.ascontiguousarray(w_oik.transpose(2, 1, 0).reshape(k * Cin, Cout))  # [j*Cin+ci, co]
    y = win.reshape(B * Lout, k * Cin) @ wmat
    y = y.reshape(B, Lout, Cout)
    if bias is not None:
        y = y + bias
    return y.astype(x_btc.dtype)


def feature_extractor(cfg, W: Dict[str, np.ndarray], padded: np.ndarray, taps: Optional[dict] = None) -> np.ndarray:
    """``ConvFeatureExtractionModel.forward`` (wav2vec2_model.py:2857-2934; WavLM.py:408-529).
    Input (B, n_max); output channel-last (B, T, C) — the reference's (B, C, T) transposed."""
    x = padded[:, :, None]
    for i, (_, k, s) in enumerate(cfg.conv_layers):
        p = f"feature_extractor.conv_layers.{i}"
        x = conv1d_channel_last(x, W[f"{p}.0.weight"], W.get(f"{p}.0.bias"), s)
        if cfg.extractor_mode == "layer_norm":
            x = layer_norm(x, W[f"{p}.2.1.weight"], W[f"{p}.2.1.bias"])  # :2887-2897
        elif i == 0:
            x = group_norm_per_channel(x, W[f"{p}.2.weight"], W[f"{p}.2.bias"])  # :2898-2904
        x = gelu(x)
        if taps is not None:
            taps[f"conv{i}"] = x
    return x


# ------------------------------------------------------------------------------------------------
# positional conv
# ------------------------------------------------------------------------------------------------

def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """``nn.utils.weight_norm(conv, dim=2)`` (wav2vec2_model.py:2950; WavLM.py:548):
    w[:,:,k] = g[k] * v[:,:,k] / ||v[:,:,k]||_F."""
    norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True))
    return (g.astype(np.float64) * v.astype(np.float64) / norm).astype(v.dtype)


def grouped_conv_same(x_btd: np.ndarray, w: np.ndarray, bias: np.ndarray, G: int) -> np.ndarray:
    """Conv1d(D, D, K, padding=K//2, groups=G) + SamePad (drop the last frame when K is even) on channel-last data
    (wav2vec2_model.py:2941-2951,1797-1808)."""
    B, T, D = x_btd.shape
    K = w.shape[2]
    Dg = D // G
    pad = K // 2
    xp = np.zeros((B, T + 2 * pad, D), dtype=x_btd.dtype)
    xp[:, pad:pad + T] = x_btd
    Tout = T + 2 * pad - K + 1
    out = np.empty((B, Tout, D), dtype=x_btd.dtype)
    it = xp.itemsize
    Tp = T + 2 * pad
    for g in range(G):
        xg = np.ascontiguousarray(xp[:, :, g * Dg:(g + 1) * Dg])
        win = np.lib.stride_tricks.as_strided(xg, shape=(B, Tout, K * Dg), strides=(Tp * Dg * it, Dg * it, it))
        wg = w[g * Dg:(g + 1) * Dg]  # (Dg_out, Dg_in, K)
        wmat = np.ascontiguousarray(wg.transpose(2, 1, 0).reshape(K * Dg, Dg))
        out[:, :, g * Dg:(g + 1) * Dg] = (win.reshape(B * Tout, K * Dg) @ wmat).reshape(B, Tout, Dg)
    out = out + bias
    if K % 2 == 0:
        out = out[:, :-1]  # SamePad
    return out


def pos_conv(cfg, W: Dict[str, np.ndarray], x_btd: np.ndarray, prefix: str = "encoder.pos_conv") -> np.ndarray:
    """``make_conv_pos`` + ``SamePad`` + GELU (wav2vec2_model.py:2937-2953,1797-1808): weight-normed grouped
    Conv1d(D, D, k, padding=k//2, groups=g), drop the last frame when k is even, GELU.  data2vec (``pos_conv_depth`` > 1,
    :2995-3023): a stack of {Conv1d(k = max(3, conv_pos // depth)) -> SamePad -> LayerNorm(no affine) -> GELU}."""
    G = cfg.conv_pos_groups
    if getattr(cfg, "pos_conv_depth", 1) > 1:
        y = x_btd
        for i in range(cfg.pos_conv_depth):
            y = grouped_conv_same(y, W[f"{prefix}.{i}.0.weight"].astype(x_btd.dtype),
                                  W[f"{prefix}.{i}.0.bias"].astype(x_btd.dtype), G)
            y = gelu(layer_norm(y, None, None))
        return y
    w = fold_weight_norm(W[f"{prefix}.0.weight_g"], W[f"{prefix}.0.weight_v"]).astype(x_btd.dtype)
    return gelu(grouped_conv_same(x_btd, w, W[f"{prefix}.0.bias"].astype(x_btd.dtype), G))


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------

def relative_position_bucket(rel: np.ndarray, num_buckets: int, max_distance: int) -> np.ndarray:
    """``_relative_positions_bucket(bidirectional=True)`` (wavlm/modules.py:418-446)."""
    nb = num_buckets // 2
    ret = (rel > 0).astype(np.int64) * nb
    a = np.abs(rel)
    max_exact = nb // 2
    is_small = a < max_exact
    # float32 log like torch's ``relative_positions.float()``; guard a == 0 (unused branch there)
    af = np.maximum(a, 1).astype(np.float32)
    large = max_exact + (np.log(af / np.float32(max_exact)) / np.float32(math.log(max_distance / max_exact))
                         * np.float32(nb - max_exact)).astype(np.int64)
    large = np.minimum(large, nb - 1)
    return ret + np.where(is_small, a, large)


def rel_pos_bias(cfg, W: Dict[str, np.ndarray], T: int, dtype) -> np.ndarray:
    """``compute_bias`` (wavlm/modules.py:448-462): (H, T, T) with bias[h,i,j] = E[bucket(j-i), h]."""
    ctx = np.arange(T)[:, None]
    mem = np.arange(T)[None, :]
    bucket = relative_position_bucket(mem - ctx, cfg.num_buckets, cfg.max_distance)
    E = W["encoder.layers.0.self_attn.relative_attention_bias.weight"].astype(dtype)  # (buckets, H)
    return np.ascontiguousarray(E[bucket].transpose(2, 0, 1))


def multihead_attention(cfg, W, prefix: str, x: np.ndarray, valid: Sequence[int],
                        pos_bias: Optional[np.ndarray]) -> np.ndarray:
    """Self-attention as executed by ``F.multi_head_attention_forward`` from
    wav2vec2_model.py:1146-1168 and wavlm/modules.py:556-579: separate q/k/v weights, q scaled by
    head_dim**-0.5, additive float mask, key-padding mask as -inf, fp32 softmax, out_proj.
    WavLM gate: wavlm/modules.py:535-551 (computed from the layer *input* split into heads)."""
    B, T, D = x.shape
    H = cfg.encoder_attention_heads
    dh = D // H
    dt = x.dtype
    q = x @ W[f"{prefix}.q_proj.weight"].T.astype(dt) + W[f"{prefix}.q_proj.bias"].astype(dt)
    k = x @ W[f"{prefix}.k_proj.weight"].T.astype(dt) + W[f"{prefix}.k_proj.bias"].astype(dt)
    v = x @ W[f"{prefix}.v_proj.weight"].T.astype(dt) + W[f"{prefix}.v_proj.bias"].astype(dt)
    q = q.reshape(B, T, H, dh).transpose(0, 2, 1, 3) * dt.type(dh ** -0.5)
    k = k.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    v = v.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    s = q @ k.transpose(0, 1, 3, 2)  # (B,H,T,T)
    if pos_bias is not None:
        bias = pos_bias[None]  # (1,H,T,T)
        if cfg.gru_rel_pos:
            xh = x.reshape(B, T, H, dh).transpose(0, 2, 1, 3)  # layer input split into heads
            gl = xh @ W[f"{prefix}.grep_linear.weight"].T.astype(dt) + W[f"{prefix}.grep_linear.bias"].astype(dt)
            gl = gl.reshape(B, H, T, 2, 4).sum(-1)
            gate = 1.0 / (1.0 + np.exp(-gl))
            ga, gb = gate[..., 0:1], gate[..., 1:2]
            grep_a = W[f"{prefix}.grep_a"].astype(dt).reshape(1, H, 1, 1)
            gate_a_1 = ga * (gb * grep_a - 1.0) + 2.0  # (B,H,T,1)
            bias = gate_a_1 * bias
        s = s + bias.astype(dt)
    for b in range(B):
        if valid[b] < T:
            s[b, :, :, valid[b]:] = -np.inf
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, D)
    return (o @ W[f"{prefix}.out_proj.weight"].T.astype(dt) + W[f"{prefix}.out_proj.bias"].astype(dt)).astype(dt)


# ------------------------------------------------------------------------------------------------
# transformer layer + encoder
# ------------------------------------------------------------------------------------------------

def encoder_layer(cfg, W, l: int, x: np.ndarray, valid, pos_bias, ffn_out: Optional[list] = None,
                  prefix: str = "encoder") -> np.ndarray:
    """``TransformerSentenceEncoderLayer.forward`` (wav2vec2_model.py:3260-3322; WavLM.py:709-774;
    distiller/module.py:191-243).  ``ffn_out`` collects ``layer_result`` = fc2 output before the residual (:3296,3317)."""
    p = f"{prefix}.layers.{l}"
    dt = x.dtype
    ln1 = (W[f"{p}.self_attn_layer_norm.weight"].astype(dt), W[f"{p}.self_attn_layer_norm.bias"].astype(dt))
    ln2 = (W[f"{p}.final_layer_norm.weight"].astype(dt), W[f"{p}.final_layer_norm.bias"].astype(dt))

    def ffn(h):
        h = gelu(h @ W[f"{p}.fc1.weight"].T.astype(dt) + W[f"{p}.fc1.bias"].astype(dt))
        h = h @ W[f"{p}.fc2.weight"].T.astype(dt) + W[f"{p}.fc2.bias"].astype(dt)
        if ffn_out is not None:
            ffn_out.append(h.astype(dt))
        return h

    if cfg.layer_norm_first:
        x = x + multihead_attention(cfg, W, f"{p}.self_attn", layer_norm(x, *ln1), valid, pos_bias)
        x = x + ffn(layer_norm(x, *ln2))
    else:
        x = layer_norm(x + multihead_attention(cfg, W, f"{p}.self_attn", x, valid, pos_bias), *ln1)
        x = layer_norm(x + ffn(x), *ln2)
    return x.astype(dt)


def forward(cfg, weights: Dict[str, np.ndarray], wavs: List[np.ndarray], dtype=np.float32,
            n_max: Optional[int] = None, taps: Optional[dict] = None, selection: Optional[str] = None) -> List[np.ndarray]:
    """``UpstreamExpert.__call__(wavs)["hidden_states"]`` for hubert / wav2vec2 / wavlm / distiller.

    Follows hubert/expert.py:56-72 → HubertModel.forward (hubert_model.py:466-513) →
    TransformerEncoder (wav2vec2_model.py:3046-3121); the hook capture of upstream/interfaces.py:90-131
    becomes the returned list: [input of layer 0 .. input of layer NL-1, encoder output].
    ``n_max`` > max(len) reproduces a data-parallel shard padded to the global batch maximum (§8e).
    ``selection`` (wav2vec2/expert.py:81-93): "fairseq_layers" = every layer's output (``layer_results[i][0]``),
    "fairseq_layers_before_residual" = every layer's fc2 output before the residual (``layer_results[i][2]``).
    family "distiller" (distiller/expert.py:43-52, model.py:178-268, module.py:302-334):
    [feat_final, layer outputs ..., prediction heads ...].
    """
    if cfg.family == "multires_hubert":  # the U-net of encoders: oracle/multires_oracle.py
        from . import multires_oracle

        assert selection is None and taps is None
        return multires_oracle.forward(cfg, weights, wavs, dtype=dtype, n_max=n_max)
    dt = np.dtype(dtype)
    W = {k: v.astype(dt) for k, v in weights.items()}
    lens = [int(len(w)) for w in wavs]
    if n_max is None:
        n_max = max(lens)
    B = len(wavs)
    padded = np.zeros((B, n_max), dtype=dt)
    for b, w in enumerate(wavs):
        w = w.astype(dt)
        if cfg.normalize:
            w = wav_normalize(w, getattr(cfg, "wav_norm_eps", EPS))
        padded[b, :lens[b]] = w

    feats = feature_extractor(cfg, W, padded, taps)  # (B,T,C)
    T = feats.shape[1]
    assert T == cfg.num_frames(n_max)
    valid = [cfg.valid_frames(n, n_max) for n in lens]

    x = feats
    if cfg.feature_layer_norm:
        x = layer_norm(x, W["layer_norm.weight"], W["layer_norm.bias"])  # hubert_model.py:482-483
    x = x @ W["post_extract_proj.weight"].T + W["post_extract_proj.bias"]  # :489-490
    for b in range(B):
        x[b, valid[b]:] = 0  # index_put(x, padding_mask, 0) wav2vec2_model.py:3061-3062; distiller/module.py:303-304
    if taps is not None:
        taps["proj"] = x.copy()
    feat_final = x  # distiller: the in-place zeroing above is visible in hidden_states[0] (a view of feat_final)
    x = x + pos_conv(cfg, W, x)  # :3064-3067
    if not cfg.layer_norm_first:
        x = layer_norm(x, W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"])  # :3069-3070

    pos_bias = None
    if cfg.family == "wavlm" and cfg.relative_position_embedding:
        pos_bias = rel_pos_bias(cfg, W, T, dt)  # layer 0 computes it, later layers reuse it (WavLM.py:622-632)

    hidden, layer_out, ffn_out = [], [], []
    for l in range(cfg.encoder_layers):
        hidden.append(x)
        x = encoder_layer(cfg, W, l, x, valid, pos_bias, ffn_out)
        layer_out.append(x)
    if cfg.layer_norm_first:
        x = layer_norm(x, W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"])  # :3049-3050
    hidden.append(x)
    if selection == "fairseq_layers":
        return layer_out
    if selection == "fairseq_layers_before_residual":
        return ffn_out
    assert selection is None, selection
    if cfg.family == "distiller":
        # output_layer = Linear(D, N*D) -> GELU -> SplitLinear(D, N, D) on the encoder output (model.py:155-161,245;
        # module.py:77-90); pred.reshape(B, T, N, D).permute(0, 2, 1, 3) -> N tensors (expert.py:47-50)
        N, D = cfg.pred_heads, cfg.encoder_embed_dim
        h = gelu(x @ W["output_layer.0.weight"].T + W["output_layer.0.bias"]).reshape(B, T, N, D)
        w2, b2 = W["output_layer.2.weight"], W["output_layer.2.bias"].reshape(N, D)
        preds = [(h[:, :, k] @ w2[k] + b2[k]).astype(dt) for k in range(N)]
        return [feat_final] + layer_out + preds
    return hidden


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """Parity metric of SURVEY §8d: ||a-b||_F / ||b||_F."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
