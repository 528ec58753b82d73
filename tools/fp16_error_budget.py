#!/usr/bin/env python3
"""Where does the fp16x2 mode's error come from?  (CPU, numpy fp64; test / design infrastructure — imports oracle/.)

S3ENC_F16X2 keeps every GEMM weight as two fp16 terms, so only the ACTIVATIONS' fp16 rounding is left.  This script re-runs the
transformer stack of a fixture in float64 with fp16 rounding injected at the sites where the HIP path stores a 16-bit operand
(LayerNorm output -> q|k|v / fc1, q / k / v, the softmax probabilities, the attention output -> out_proj, GELU(fc1) -> fc2, the
conv stack's activations) and reports the max-over-layers relative error with ALL sites rounded, with each site alone exact
(what a two-term / fp32 operand THERE would buy) and with each site alone rounded.

usage: tools/fp16_error_budget.py [golden fixture names ...]      (default: the post-LN pretrained-like base fixtures)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_golden
from oracle import encoder_oracle as O

SITES = ["conv", "feat", "ln_out", "q", "k", "v", "p", "attn_out", "fc1_out"]


def r16(x):
    return x.astype(np.float16).astype(np.float64)


# ---- MX-fp4 emulation (round 5: the second weight term of S3ENC_F16X2 on the scaled-MFMA pipe, gemm16.hip MXW) -------------------
E2M1 = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def mx4(x):
    """x (..., K) -> its MX-fp4 image, dequantised: per 32 elements along K an E8M0 scale 2^e (the tightest with max / 2^e <= 6) and
    e2m1 values rounded to nearest (ties to the even code) — what `pack_mx4_lo` writes for W_lo and the kernel builds for A."""
    K = x.shape[-1]
    b = x.reshape(x.shape[:-1] + (K // 32, 32))
    amax = np.abs(b).max(-1, keepdims=True)
    with np.errstate(divide="ignore"):
        e = np.where(amax > 0, np.ceil(np.log2(np.maximum(amax, 1e-300) / 6.0)), -127.0)
    e = np.maximum(e, -127.0)
    s = np.exp2(e)
    a = np.abs(b) / s
    idx = np.clip(np.searchsorted(E2M1, a, side="left"), 1, 7)          # E2M1[idx - 1] < a <= E2M1[idx] (a > 0)
    lo_v, hi_v = E2M1[idx - 1], E2M1[idx]
    up = (a - lo_v > hi_v - a) | ((a - lo_v == hi_v - a) & (idx % 2 == 0))  # ties: the even code
    q = np.where(a <= 0, 0.0, np.where(up, hi_v, lo_v))
    return (np.sign(b) * q * s).reshape(x.shape)


def mm(site, rounded, a, Wm):
    """a @ Wm.T as the fp16x2 GEMM `site` computes it: exact weights (two fp16 terms) — or, with "mx_<site>" in `rounded`,
    a @ fp16(W).T + mx4(a) @ mx4(W - fp16(W)).T (the lo term as MX-fp4 images of both operands)."""
    if "mx_" + site not in rounded:
        return a @ Wm.T
    w_hi = r16(Wm)
    return a @ w_hi.T + mx4(a) @ mx4(Wm - w_hi).T


def run(cfg, W, wavs, rounded):
    rd = lambda site, x: r16(x) if site in rounded else x
    dt = np.float64
    lens = [len(w) for w in wavs]
    n_max = max(lens)
    B = len(wavs)
    padded = np.zeros((B, n_max))
    for b, w in enumerate(wavs):
        w = w.astype(dt)
        if cfg.normalize:
            w = O.wav_normalize(w, getattr(cfg, "wav_norm_eps", O.EPS))
        padded[b, :lens[b]] = w
    # conv stack: activations between the conv layers are 16-bit operands of the next conv GEMM
    # "conv" = the outputs of conv0 .. conv(n-2) (the last conv feeds the fp32 LayerNorm / is rounded under "feat");
    # "conv<=i" = only the outputs of conv0 .. conv i
    upto = len(cfg.conv_layers) - 2 if "conv" in rounded else max([int(s[6:]) for s in rounded if s.startswith("conv<=")], default=-1)
    feats = feature_extractor_rounded(cfg, W, padded, upto, "mx_conv1" in rounded) if (upto >= 0 or "mx_conv1" in rounded) else O.feature_extractor(cfg, W, padded, None)
    T = feats.shape[1]
    valid = [cfg.valid_frames(n, n_max) for n in lens]
    x = feats
    if cfg.feature_layer_norm:
        x = O.layer_norm(x, W["layer_norm.weight"], W["layer_norm.bias"])
    x = rd("feat", x) @ W["post_extract_proj.weight"].T + W["post_extract_proj.bias"]
    for b in range(B):
        x[b, valid[b]:] = 0
    x = x + O.pos_conv(cfg, W, x)   # (the positional conv reads the fp32 stream; its operand rounding is counted under "feat")
    if not cfg.layer_norm_first:
        x = O.layer_norm(x, W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"])
    H = cfg.encoder_attention_heads
    D = x.shape[-1]
    dh = D // H
    hidden = []
    # WavLM: the bucketed relative-position bias (layer 0's table, reused by every layer) and the gate computed from the
    # attention's input — both fp32 in the HIP path (the gate from the LayerNorm kernel's registers), so never rounded here
    pos_bias = O.rel_pos_bias(cfg, W, T, dt) if getattr(cfg, "relative_position_embedding", False) else None
    for l in range(cfg.encoder_layers):
        hidden.append(x)
        p = f"encoder.layers.{l}"
        ln1 = (W[f"{p}.self_attn_layer_norm.weight"], W[f"{p}.self_attn_layer_norm.bias"])
        ln2 = (W[f"{p}.final_layer_norm.weight"], W[f"{p}.final_layer_norm.bias"])

        def attn(a):
            a_exact = a
            a = rd("ln_out", a)
            q = (mm("qkv", rounded, a, W[f"{p}.self_attn.q_proj.weight"]) + W[f"{p}.self_attn.q_proj.bias"]) * dh ** -0.5
            k = mm("qkv", rounded, a, W[f"{p}.self_attn.k_proj.weight"]) + W[f"{p}.self_attn.k_proj.bias"]
            v = mm("qkv", rounded, a, W[f"{p}.self_attn.v_proj.weight"]) + W[f"{p}.self_attn.v_proj.bias"]
            sp = lambda t: t.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
            q, k, v = sp(rd("q", q)), sp(rd("k", k)), sp(rd("v", v))
            s = q @ k.transpose(0, 1, 3, 2)
            if pos_bias is not None:
                bias = pos_bias[None]
                if cfg.gru_rel_pos:
                    xh = a_exact.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
                    gl = xh @ W[f"{p}.self_attn.grep_linear.weight"].T + W[f"{p}.self_attn.grep_linear.bias"]
                    gl = gl.reshape(B, H, T, 2, 4).sum(-1)
                    gate = 1.0 / (1.0 + np.exp(-gl))
                    bias = (gate[..., 0:1] * (gate[..., 1:2] * W[f"{p}.self_attn.grep_a"].reshape(1, H, 1, 1) - 1.0) + 2.0) * bias
                s = s + bias
            for b in range(B):
                s[b, :, :, valid[b]:] = -np.inf
            s = s - s.max(-1, keepdims=True)
            e = np.exp(s)
            den = e.sum(-1, keepdims=True)           # the row sum is accumulated in fp32 from the unrounded probabilities
            o = (rd("p", e) @ v) / den
            o = o.transpose(0, 2, 1, 3).reshape(B, T, D)
            return rd("attn_out", o) @ W[f"{p}.self_attn.out_proj.weight"].T + W[f"{p}.self_attn.out_proj.bias"]

        def ffn(a):
            h = O.gelu(mm("fc1", rounded, rd("ln_out", a), W[f"{p}.fc1.weight"]) + W[f"{p}.fc1.bias"])
            return mm("fc2", rounded, rd("fc1_out", h), W[f"{p}.fc2.weight"]) + W[f"{p}.fc2.bias"]

        if cfg.layer_norm_first:
            x = x + attn(O.layer_norm(x, *ln1))
            x = x + ffn(O.layer_norm(x, *ln2))
        else:
            x = O.layer_norm(x + attn(x), *ln1)
            x = O.layer_norm(x + ffn(x), *ln2)
    if cfg.layer_norm_first:
        x = O.layer_norm(x, W["encoder.layer_norm.weight"], W["encoder.layer_norm.bias"])
    hidden.append(x)
    return hidden


def feature_extractor_rounded(cfg, W, padded, upto, mx_conv1=False):
    """oracle feature_extractor with the OUTPUT of conv layers 0 .. upto rounded to fp16 (the next conv GEMM's operand); mx_conv1:
    conv1's lo weight term as MX-fp4 images (k axis tap-major, j * Cin + ci: a 32-block is 32 consecutive channels of one tap)."""
    taps = {}
    orig, orig_conv = O.gelu, O.conv1d_channel_last
    calls, convs = [0], [0]

    def gelu_r(x):
        calls[0] += 1
        return r16(orig(x)) if calls[0] - 1 <= upto else orig(x)

    def conv_mx(x, w, bias, stride):
        convs[0] += 1
        if not (mx_conv1 and convs[0] == 2):
            return orig_conv(x, w, bias, stride)
        B, L, Cin = x.shape
        Cout, _, k = w.shape
        Lout = (L - k) // stride + 1
        x = np.ascontiguousarray(x)
        it = x.itemsize
        win = np.lib.stride_tricks.as_strided(x, shape=(B, Lout, k * Cin), strides=(L * Cin * it, stride * Cin * it, it)).reshape(B * Lout, k * Cin)
        wm = np.ascontiguousarray(w.transpose(0, 2, 1).reshape(Cout, k * Cin))  # [co, j*Cin+ci]
        w_hi = r16(wm)
        y = win @ w_hi.T + mx4(np.ascontiguousarray(win)) @ mx4(wm - w_hi).T
        y = y.reshape(B, Lout, Cout)
        return y + bias if bias is not None else y

    O.gelu, O.conv1d_channel_last = gelu_r, conv_mx
    try:
        return O.feature_extractor(cfg, W, padded, taps)
    finally:
        O.gelu, O.conv1d_channel_last = orig, orig_conv


def main_mx(names):
    """`fp16_error_budget.py mx [fixtures]`: the fp16x2 mode's error with the MX second term on each GEMM (the CPU twin of
    tools/mx_mask_sweep.py's GPU table; float64 sums, the mode's activation-rounding sites as the engine has them)."""
    masks = [((), "two fp16 terms"), (("mx_conv1",), "conv1"), (("mx_qkv",), "q|k|v"), (("mx_fc1",), "fc1"), (("mx_fc2",), "fc2"),
             (("mx_qkv", "mx_fc1", "mx_fc2"), "q|k|v + fc1 + fc2"), (("mx_conv1", "mx_qkv", "mx_fc1", "mx_fc2"), "all four")]
    print("| fixture | activation sites rounded | " + " | ".join(d for _, d in masks) + " |")
    print("|---|---|" + "---:|" * len(masks))
    for name in names:
        meta, cfg, weights, wavs, golden, _ = load_golden(name)
        W = {k: v.astype(np.float64) for k, v in weights.items()}
        exact = run(cfg, W, wavs, set())
        err = lambda hs: max(O.rel_err(h, e) for h, e in zip(hs, exact))
        # what S3ENC_F16X2 rounds today: conv0's output (conv1's operand), the LayerNorm outputs, q / k / v / P, GELU(fc1)
        act = {"conv<=0", "ln_out", "q", "k", "v", "p", "fc1_out"}
        for label, base in (("none (the MX term alone)", set()), ("the mode's", act)):
            cells = ["%.2e" % err(run(cfg, W, wavs, base | set(m))) if (m or base) else "0" for m, _ in masks]
            print(f"| `{name}` | {label} | " + " | ".join(cells) + " |")
            sys.stdout.flush()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "mx":
        return main_mx(sys.argv[2:] or ["wav2vec2_base_pl", "hubert_base_pl"])
    names = sys.argv[1:] or ["wav2vec2_base_pl", "hubert_base_pl", "hubert_base_pseudo"]
    for name in names:
        meta, cfg, weights, wavs, golden, _ = load_golden(name)
        W = {k: v.astype(np.float64) for k, v in weights.items()}
        exact = run(cfg, W, wavs, set())
        err = lambda hs: max(O.rel_err(h, e) for h, e in zip(hs, exact))
        print(f"\n## {name} ({meta['config']}): max over hidden states of the relative error vs the exact fp64 evaluation\n")
        print(f"| rounded to fp16 | error |\n|---|---:|")
        allr = err(run(cfg, W, wavs, set(SITES)))
        print(f"| every site | {allr:.2e} |")
        others = [s for s in SITES if s != "conv"]
        for i in range(len(cfg.conv_layers) - 1):
            e1 = err(run(cfg, W, wavs, set(others) | {f"conv<={i}"}))
            print(f"| every other site + the outputs of conv0..conv{i} (conv{i + 2}.. on fp32 activations) | {e1:.2e} |")
            sys.stdout.flush()
        for s in SITES:
            e1 = err(run(cfg, W, wavs, set(SITES) - {s}))
            e2 = err(run(cfg, W, wavs, {s}))
            print(f"| every site but `{s}` | {e1:.2e} |")
            print(f"| only `{s}` | {e2:.2e} |")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
