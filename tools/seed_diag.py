#!/usr/bin/env python3
"""Where does a fixture's fp16x2 error enter?  (GPU box; test infrastructure — imports oracle/.)  Per-layer relative error of the
HIP encoder against the reference golden, and the stage taps (conv stack, LayerNorm(C), post_extract_proj, positional conv)
against the float64 numpy oracle, for the modes / tuning keys given.
usage: seed_diag.py <fixture> [mode[:key=value,...]] ...   e.g.  seed_diag.py hubert_base_s1_pl fp16x2 fp16x2:gemm16_mx=0 fp32x3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_golden
from oracle import encoder_oracle as O
from s3prl_amd import _lib
from s3prl_amd.encoder import HipEncoder

name = sys.argv[1]
specs = sys.argv[2:] or ["fp32", "fp16x2"]
meta, cfg, weights, wavs, golden, _ = load_golden(name)
dev = [torch.from_numpy(w).cuda() for w in wavs]
ts, cs = meta["t_stride"], meta["c_stride"]
taps = {}
O.forward(cfg, weights, wavs, dtype=np.float64, taps=taps)
h64 = O.forward(cfg, weights, wavs, dtype=np.float64)
lib = _lib.load()


def layer0_exact():
    """q | k | v and the attention output (before out_proj) of encoder layer 0 in float64, from the exact layer input."""
    W = {k: v.astype(np.float64) for k, v in weights.items()}
    x = h64[0]
    B, T, D = x.shape
    H = cfg.encoder_attention_heads
    dh = D // H
    p = "encoder.layers.0"
    a = O.layer_norm(x, W[f"{p}.self_attn_layer_norm.weight"], W[f"{p}.self_attn_layer_norm.bias"]) if cfg.layer_norm_first else x
    q = a @ W[f"{p}.self_attn.q_proj.weight"].T + W[f"{p}.self_attn.q_proj.bias"]
    k = a @ W[f"{p}.self_attn.k_proj.weight"].T + W[f"{p}.self_attn.k_proj.bias"]
    v = a @ W[f"{p}.self_attn.v_proj.weight"].T + W[f"{p}.self_attn.v_proj.bias"]
    n_max = max(len(w) for w in wavs)
    valid = [cfg.valid_frames(len(w), n_max) for w in wavs]
    sp = lambda t: t.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
    s_ = (sp(q) * dh ** -0.5) @ sp(k).transpose(0, 1, 3, 2)
    if getattr(cfg, "relative_position_embedding", False):
        return q, k, v, None, valid  # (the bias path: compare q | k | v only)
    for b in range(B):
        s_[b, :, :, valid[b]:] = -np.inf
    s_ = s_ - s_.max(-1, keepdims=True)
    e = np.exp(s_)
    o = ((e / e.sum(-1, keepdims=True)) @ sp(v)).transpose(0, 2, 1, 3).reshape(B, T, D)
    return q, k, v, o, valid


q64, k64, v64, o64, valid0 = layer0_exact()
print(f"# {name}: per-layer ||h - h_ref|| / ||h_ref|| vs the reference golden; taps vs the float64 oracle\n")
for spec in specs:
    mode, _, kv = spec.partition(":")
    keys = dict(x.split("=") for x in kv.split(",")) if kv else {}
    for k, v in keys.items():
        _lib.check(lib.s3enc_set_tuning(k.encode(), int(v)))
    enc = HipEncoder(cfg, weights, dtype=mode)
    hs = enc.forward(dev).cpu().numpy()
    errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
    tp = []
    for t in [f"conv{i}" for i in range(len(cfg.conv_layers) - 3, len(cfg.conv_layers))] + ["feat_ln", "proj", "posconv"]:
        if t in taps:
            try:
                got = enc.debug_tap(t).reshape(taps[t].shape)
                tp.append(f"{t} {O.rel_err(got, taps[t]):.1e}")
            except Exception as e:  # a tap the mode does not keep
                tp.append(f"{t} n/a")
    print(f"{spec}: max {max(errs):.2e}  layers " + " ".join(f"{e:.1e}" for e in errs))
    if len(h64) == len(golden):
        # the same per-layer error over the FULL tensors (the fixture stores every t_stride-th frame and c_stride-th channel: on
        # pretrained-like statistics a few outlier channels carry most of a state's norm, and a subsample that misses them weighs the
        # ordinary channels' error against a much smaller norm), against the float64 oracle
        full = [O.rel_err(hs[l], h64[l]) for l in range(len(h64))]
        print(f"    full tensors vs float64: max {max(full):.2e}  layers " + " ".join(f"{e:.1e}" for e in full))
    print(f"    taps: " + ", ".join(tp))
    try:  # layer 0's q | k | v (q carries head_dim^-0.5, and log2(e) in the 16-bit modes) and attention output, valid frames only.
        # The tap buffers are re-used by every layer: a ONE-layer copy of the model leaves layer 0's contents in them.
        import dataclasses

        enc1 = HipEncoder(dataclasses.replace(cfg, encoder_layers=1), weights, dtype=mode)
        h1 = enc1.forward(dev).cpu().numpy()
        print(f"    one-layer model: state 0 {O.rel_err(h1[0], h64[0]):.1e}, state 1 {O.rel_err(h1[1], h64[1]):.1e} (vs float64)")
        B, T, D = q64.shape
        qkv = enc1.debug_tap("qkv0").reshape(B, T, 3 * D).astype(np.float64)
        qs = (D // cfg.encoder_attention_heads) ** -0.5 * (1.0 if mode in ("fp32", "fp32x3") else 1.4426950408889634)
        rows = np.zeros((B, T), bool)
        for b in range(B):
            rows[b, :valid0[b]] = True
        parts = [("q", qkv[..., :D] / qs, q64), ("k", qkv[..., D:2 * D], k64), ("v", qkv[..., 2 * D:], v64)]
        if o64 is not None:
            parts.append(("attn", enc1.debug_tap("attn0").reshape(B, T, D).astype(np.float64), o64))
        print("    layer 0: " + ", ".join(f"{n} {O.rel_err(g[rows], e[rows]):.1e}" for n, g, e in parts))
        enc1.close()
    except Exception as ex:
        print("    layer 0: n/a", repr(ex)[:100])
    enc.close()
    for k in keys:  # back to the defaults this tool knows
        _lib.check(lib.s3enc_set_tuning(k.encode(), {"gemm16_mx": 14}.get(k, 0)))
