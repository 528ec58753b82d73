"""CPU restatement of the reference upstream-encoder forward on the SAME ATen call sites the reference uses
(``F.conv1d``, ``F.group_norm``, ``F.layer_norm``, ``F.gelu``, ``F.linear``, ``F.multi_head_attention_forward``).
TEST INFRASTRUCTURE ONLY — like ``encoder_oracle.py`` it may only be imported by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Why a second restatement: ``/root/reference`` does not exist on the GPU box, so the reference's PyTorch-CPU path
cannot be timed there.  This file runs the same third-party kernels (oneDNN / MKL through ATen, all host threads)
in the same order and layouts as the reference ((B,C,T) convolutions, (T,B,C) transformer, hook-style capture), so
its wall time is the "reference PyTorch-CPU path" of BASELINE.json up to Python glue; the numpy oracle is the
independent arithmetic check.  Parity status: **pinned** — ``tests/test_oracle_golden.py`` checks it against the
reference-generated fixtures in ``tests/golden``.

Every function cites the reference lines it restates (paths relative to the reference root).
"""

from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _t(W, name, dtype):
    return torch.from_numpy(np.ascontiguousarray(W[name])).to(dtype)


def prepare(cfg, weights: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """state_dict -> torch tensors; folds ``weight_norm(dim=2)`` of the positional conv
    (wav2vec2_model.py:2950; WavLM.py:548) once, like the parametrisation does on every forward."""
    W = {k: _t(weights, k, dtype) for k in weights}
    for enc in ("encoder", "encoders.0"):  # "encoders.0": multires-HuBERT's only positional conv
        if f"{enc}.pos_conv.0.weight_g" in W:
            g, v = W[f"{enc}.pos_conv.0.weight_g"].double(), W[f"{enc}.pos_conv.0.weight_v"].double()
            W[f"{enc}.pos_conv.0.weight"] = (g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()).to(dtype)
    return W


def feature_extractor(cfg, W, x: torch.Tensor) -> torch.Tensor:
    """``ConvFeatureExtractionModel.forward`` (wav2vec2_model.py:2857-2934): (B, n) -> (B, C, T)."""
    x = x.unsqueeze(1)
    for i, (_, k, s) in enumerate(cfg.conv_layers):
        p = f"feature_extractor.conv_layers.{i}"
        x = F.conv1d(x, W[f"{p}.0.weight"], W.get(f"{p}.0.bias"), stride=s)  # :2879
        if cfg.extractor_mode == "layer_norm":  # TransposeLast, Fp32LayerNorm, TransposeLast :2887-2897
            x = F.layer_norm(x.transpose(-2, -1), (x.shape[1],), W[f"{p}.2.1.weight"], W[f"{p}.2.1.bias"], 1e-5)
            x = x.transpose(-2, -1)
        elif i == 0:  # Fp32GroupNorm(dim, dim) :2898-2904
            x = F.group_norm(x, x.shape[1], W[f"{p}.2.weight"], W[f"{p}.2.bias"], 1e-5)
        x = F.gelu(x)  # nn.GELU() :2896-2906
    return x


def rel_pos_bias(cfg, W, T: int) -> torch.Tensor:
    """``compute_bias`` / ``_relative_positions_bucket`` (wavlm/modules.py:418-462): (H, T, T)."""
    import math

    ctx = torch.arange(T)[:, None]
    mem = torch.arange(T)[None, :]
    rel = mem - ctx
    nb = cfg.num_buckets // 2
    buckets = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(cfg.max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    buckets = buckets + torch.where(a < max_exact, a, large)
    E = W["encoder.layers.0.self_attn.relative_attention_bias.weight"]
    return F.embedding(buckets, E).permute(2, 0, 1).contiguous()


def self_attention(cfg, W, p: str, x_tbc: torch.Tensor, key_padding_mask: Optional[torch.Tensor],
                   pos_bias: Optional[torch.Tensor]) -> torch.Tensor:
    """``MultiheadAttention.forward`` fast path -> ``F.multi_head_attention_forward`` with separate projection
    weights (wav2vec2_model.py:1124-1168); WavLM adds the gated relative-position bias as a float ``attn_mask``
    (wavlm/modules.py:535-579)."""
    T, B, D = x_tbc.shape
    H = cfg.encoder_attention_heads
    attn_mask = None
    if pos_bias is not None:
        bias = pos_bias.unsqueeze(0).repeat(B, 1, 1, 1).view(B * H, T, T)
        if cfg.gru_rel_pos:
            q = x_tbc.transpose(0, 1).reshape(B, T, H, -1).permute(0, 2, 1, 3)
            g = F.linear(q, W[f"{p}.grep_linear.weight"], W[f"{p}.grep_linear.bias"])
            ga, gb = torch.sigmoid(g.view(B, H, T, 2, 4).sum(-1)).chunk(2, dim=-1)
            gate = ga * (gb * W[f"{p}.grep_a"].view(1, H, 1, 1) - 1.0) + 2.0
            bias = gate.view(B * H, T, 1) * bias
        attn_mask = bias
    out, _ = F.multi_head_attention_forward(
        x_tbc, x_tbc, x_tbc, D, H, torch.empty([0]),
        torch.cat((W[f"{p}.q_proj.bias"], W[f"{p}.k_proj.bias"], W[f"{p}.v_proj.bias"])),
        None, None, False, 0.0, W[f"{p}.out_proj.weight"], W[f"{p}.out_proj.bias"],
        training=False, key_padding_mask=key_padding_mask, need_weights=False, attn_mask=attn_mask,
        use_separate_proj_weight=True, q_proj_weight=W[f"{p}.q_proj.weight"], k_proj_weight=W[f"{p}.k_proj.weight"],
        v_proj_weight=W[f"{p}.v_proj.weight"])
    return out


def encoder_layer(cfg, W, l: int, x: torch.Tensor, kpm, pos_bias, prefix: str = "encoder") -> torch.Tensor:
    """``TransformerSentenceEncoderLayer.forward`` (wav2vec2_model.py:3260-3322; WavLM.py:709-774), (T,B,C)."""
    p = f"{prefix}.layers.{l}"
    D = x.shape[-1]
    ln1 = (W[f"{p}.self_attn_layer_norm.weight"], W[f"{p}.self_attn_layer_norm.bias"])
    ln2 = (W[f"{p}.final_layer_norm.weight"], W[f"{p}.final_layer_norm.bias"])

    def ffn(h):
        h = F.gelu(F.linear(h, W[f"{p}.fc1.weight"], W[f"{p}.fc1.bias"]).float()).type_as(h)
        return F.linear(h, W[f"{p}.fc2.weight"], W[f"{p}.fc2.bias"])

    if cfg.layer_norm_first:
        x = x + self_attention(cfg, W, f"{p}.self_attn", F.layer_norm(x, (D,), *ln1, 1e-5), kpm, pos_bias)
        x = x + ffn(F.layer_norm(x, (D,), *ln2, 1e-5))
    else:
        x = F.layer_norm(x + self_attention(cfg, W, f"{p}.self_attn", x, kpm, pos_bias), (D,), *ln1, 1e-5)
        x = F.layer_norm(x + ffn(x), (D,), *ln2, 1e-5)
    return x


@torch.no_grad()
def forward(cfg, W: Dict[str, torch.Tensor], wavs: List[torch.Tensor], n_max: Optional[int] = None) -> List[torch.Tensor]:
    """``UpstreamExpert.__call__(wavs)["hidden_states"]`` (hubert/expert.py:56-72 -> hubert_model.py:466-513 ->
    wav2vec2_model.py:3046-3121 with the hook capture of upstream/interfaces.py:90-131).  ``W = prepare(...)``."""
    if cfg.family == "multires_hubert":
        return multires_forward(cfg, W, wavs, n_max)
    dt = W["layer_norm.weight"].dtype
    lens = [int(w.numel()) for w in wavs]
    n_max = n_max or max(lens)
    B = len(wavs)
    if cfg.normalize:  # hubert/expert.py:57-58 (eps 1e-5); Hugging Face's feature extractor: 1e-7
        wavs = [F.layer_norm(w.to(dt), w.shape, eps=getattr(cfg, "wav_norm_eps", 1e-5)) for w in wavs]
    padded = torch.zeros(B, n_max, dtype=dt)
    for b, w in enumerate(wavs):
        padded[b, : lens[b]] = w.to(dt)

    feats = feature_extractor(cfg, W, padded).transpose(1, 2)  # (B,T,C) hubert_model.py:480
    T = feats.shape[1]
    valid = [cfg.valid_frames(n, n_max) for n in lens]
    kpm = torch.zeros(B, T, dtype=torch.bool)
    for b in range(B):
        kpm[b, valid[b]:] = True
    use_mask = cfg.family != "wav2vec2" or any(n < n_max for n in lens)  # wav2vec2_model.py:2652-2671

    x = F.layer_norm(feats, (feats.shape[-1],), W["layer_norm.weight"], W["layer_norm.bias"], 1e-5)
    x = F.linear(x, W["post_extract_proj.weight"], W["post_extract_proj.bias"])
    if use_mask:
        x = x.masked_fill(kpm.unsqueeze(-1), 0.0)  # index_put(x, padding_mask, 0) :3061-3062
    xc = x.transpose(1, 2)
    depth = getattr(cfg, "pos_conv_depth", 1)
    for i in range(max(1, depth)):  # data2vec (depth > 1): conv -> SamePad -> LayerNorm -> GELU blocks :2995-3023
        name = f"encoder.pos_conv.{i}.0" if depth > 1 else "encoder.pos_conv.0"
        K = W[f"{name}.weight"].shape[-1]
        xc = F.conv1d(xc, W[f"{name}.weight"], W[f"{name}.bias"], padding=K // 2, groups=cfg.conv_pos_groups)
        if K % 2 == 0:
            xc = xc[:, :, :-1]  # SamePad :1797-1808
        if depth > 1:
            xc = F.layer_norm(xc.transpose(1, 2), (xc.shape[1],)).transpose(1, 2)
        xc = F.gelu(xc)
    x = x + xc.transpose(1, 2)
    if not cfg.layer_norm_first:
        x = F.layer_norm(x, (x.shape[-1],), W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"], 1e-5)
    pos_bias = rel_pos_bias(cfg, W, T).to(dt) if (cfg.family == "wavlm" and cfg.relative_position_embedding) else None

    x = x.transpose(0, 1)  # B x T x C -> T x B x C :3084-3085
    hidden = []
    for l in range(cfg.encoder_layers):
        hidden.append(x.transpose(0, 1))
        x = encoder_layer(cfg, W, l, x, kpm if use_mask else None, pos_bias)
    x = x.transpose(0, 1)
    if cfg.layer_norm_first:
        x = F.layer_norm(x, (x.shape[-1],), W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"], 1e-5)
    hidden.append(x)
    return hidden


# ---- multi-resolution HuBERT (upstream/multires_hubert) -----------------------------------------------------------------

def conv_adapter(W, mod: str, x_bct: torch.Tensor, up: int, down: int, kind: str) -> torch.Tensor:
    """``ConvAdapter`` / ``ConvDownsampler`` / ``ConvUpsampler`` forward (multires_hubert/hubert_model.py:1038-1078,
    1146-1167,1232-1250) on (B, C, T): ConvTranspose1d / Conv1d -> Fp32GroupNorm(1, C) -> GELU, skip connections scaled
    by sqrt(0.4), the highway branch for the two-conv adapter."""
    sc = 0.4 ** 0.5
    C = x_bct.shape[1]
    r_up = None
    x = x_bct
    if kind in ("full", "up"):
        p = f"{mod}.upsample_conv"
        y = F.conv_transpose1d(x, W[f"{p}.0.weight"], None, stride=up, padding=0, output_padding=up - 1)
        y = F.gelu(F.group_norm(y.float(), 1, W[f"{p}.2.weight"], W[f"{p}.2.bias"], 1e-5).type_as(y))
        r_up = torch.repeat_interleave(x, up, dim=2)
        n = min(y.shape[2], r_up.shape[2])
        x = (y[..., :n] + r_up[..., :n]) * sc
    if kind in ("full", "down"):
        p = f"{mod}.downsample_conv"
        k = W[f"{p}.0.weight"].shape[-1]
        y = F.conv1d(x, W[f"{p}.0.weight"], None, stride=down, padding=(k - 1) // 2)
        y = F.gelu(F.group_norm(y.float(), 1, W[f"{p}.2.weight"], W[f"{p}.2.bias"], 1e-5).type_as(y))
        r = x[..., ::down]
        n = min(y.shape[2], r.shape[2])
        x = (y[..., :n] + r[..., :n]) * sc
        if kind == "full":
            r = r_up[..., ::down]
            n = min(x.shape[2], r.shape[2])
            x = (x[..., :n] + r[..., :n]) * sc
    return x


@torch.no_grad()
def multires_forward(cfg, W, wavs: List[torch.Tensor], n_max: Optional[int] = None) -> List[torch.Tensor]:
    """``UpstreamExpert.__call__(wavs)["hidden_states"]`` of upstream/multires_hubert (expert.py:30-126 ->
    hubert_model.py:738-852 -> wav2vec2_model.py:3046-3121) on the reference's ATen call sites."""
    dt = W["layer_norm.weight"].dtype
    lens = [int(w.numel()) for w in wavs]
    n_max = n_max or max(lens)
    B = len(wavs)
    if cfg.normalize:
        wavs = [F.layer_norm(w.to(dt), w.shape) for w in wavs]
    padded = torch.zeros(B, n_max, dtype=dt)
    for b, w in enumerate(wavs):
        padded[b, : lens[b]] = w.to(dt)
    x = feature_extractor(cfg, W, padded).transpose(1, 2)
    T0 = x.shape[1]
    x = F.layer_norm(x, (x.shape[-1],), W["layer_norm.weight"], W["layer_norm.bias"], 1e-5)
    x = F.linear(x, W["post_extract_proj.weight"], W["post_extract_proj.bias"])
    valid = [cfg.valid_frames(n, n_max) for n in lens]
    blocks, T_out = cfg.multires_plan(T0)
    R = len(cfg.rate_pairs) + 1
    states, residuals = [], []
    for bi, blk in enumerate(blocks):
        if blk["adapter"] is not None:
            kind, up, down, mod = blk["adapter"]
            x = conv_adapter(W, mod, x.transpose(1, 2), up, down, kind).transpose(1, 2)
            ue, de = (up if kind != "down" else 1), (down if kind != "up" else 1)
            valid = [min(-(-(v * ue) // de), x.shape[1]) for v in valid]
        T = x.shape[1]
        kpm = torch.zeros(B, T, dtype=torch.bool)
        for b in range(B):
            kpm[b, valid[b]:] = True
        x_in = x.masked_fill(kpm.unsqueeze(-1), 0.0)  # index_put (in place in the reference: visible to x + residual)
        h = x_in
        p = blk["prefix"]
        if bi == 0:
            K = W[f"{p}.pos_conv.0.weight"].shape[-1]
            xc = F.conv1d(h.transpose(1, 2), W[f"{p}.pos_conv.0.weight"], W[f"{p}.pos_conv.0.bias"], padding=K // 2,
                          groups=cfg.conv_pos_groups)
            if K % 2 == 0:
                xc = xc[:, :, :-1]
            h = h + F.gelu(xc).transpose(1, 2)
        if not cfg.layer_norm_first:
            h = F.layer_norm(h, (h.shape[-1],), W[f"{p}.layer_norm.weight"], W[f"{p}.layer_norm.bias"], 1e-5)
        h = h.transpose(0, 1)
        for l in range(blk["layers"]):
            states.append((h.transpose(0, 1), blk["factor"]))
            h = encoder_layer(cfg, W, l, h, kpm, None, prefix=p)
        h = h.transpose(0, 1)
        if cfg.layer_norm_first:
            h = F.layer_norm(h, (h.shape[-1],), W[f"{p}.layer_norm.weight"], W[f"{p}.layer_norm.bias"], 1e-5)
        states.append((h, blk["factor"]))
        if bi < R - 1:
            residuals.append(h)
            x = h
        elif bi == R - 1:
            x = x_in + h
            residuals.reverse()
        else:
            r = residuals[bi - R]
            c = min(h.shape[1], r.shape[1])
            x = h[:, :c] + r[:, :c]
            valid = [min(v, c) for v in valid]
    return [torch.repeat_interleave(h, f, dim=1)[:, :T_out].contiguous() for h, f in states]
