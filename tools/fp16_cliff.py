#!/usr/bin/env python3
"""Where does each 16-bit mode leave its tolerance — or its number range?  (GPU box)

Pretrained-like weights (synth_weights(profile="pretrained_like")) with the residual stream's outlier WRITERS scaled x1, x3,
x10, x30, x100 (s3prl_amd.synth.scale_outlier_writers): the massive activations of released large checkpoints grow with
exactly these rows.  Per model, scale and compute mode: the largest |state| the exact fp32 run sees, the mode's max per-layer
relative error against that fp32 run (the fp32 HIP path is pinned to the reference at <= 4e-6 on the `*_pl` goldens — the
reference itself is not on the GPU box), whether every state is finite, and what the library's own flag says
(s3enc_forward_status, ABI 6: a row LayerNorm met a non-finite statistic).  The last table is the cliff: per mode the first
scale with a state > 1e-3 away / a non-finite state.

usage: fp16_cliff.py [config names ...] > profiles/rNN_fp16_cliff.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import named_config, scale_outlier_writers, synth_wavs, synth_weights

SCALES = (1.0, 3.0, 10.0, 30.0, 100.0)
FFN_SCALES = (1.0, 10.0, 100.0, 1000.0, 10000.0)  # second sweep: fc1 (weight and bias) x s — GELU(fc1), fc2's fp16 OPERAND, grows with s
MODES = ("fp32x3", "fp16x2", "fp16", "bf16")
SEED = 1


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def scale_ffn_hidden(weights, factor):
    out = dict(weights)
    for name, v in weights.items():
        if name.endswith((".fc1.weight", ".fc1.bias")):
            out[name] = (np.asarray(v, dtype=np.float32) * np.float32(factor)).astype(np.float32)
    return out


def sweep(name, cfg, wavs, variants, label, first_bad, key):
    print(f"| {label} | max abs state (fp32) | " + " | ".join(f"{m}: err / finite / flag" for m in MODES) + " |")
    print("|---:|---:|" + "---|" * len(MODES))
    for sc, w in variants:
        enc = HipEncoder(cfg, w, dtype="fp32", check="off")
        ref = enc.forward(wavs).cpu().numpy()
        flag32 = enc.status()
        enc.close()
        assert np.isfinite(ref).all() and flag32 == 0, "the fp32 run itself is not finite"
        cells = []
        for m in MODES:
            enc = HipEncoder(cfg, w, dtype=m, check="off")
            hs = enc.forward(wavs).cpu().numpy()
            flag = enc.status()
            enc.close()
            finite = bool(np.isfinite(hs).all())
            err = max(rel(np.nan_to_num(hs[l], nan=0.0, posinf=0.0, neginf=0.0), ref[l]) for l in range(len(ref)))
            # the library's flag must agree with a scan of the states: nothing non-finite goes unreported
            agree = "" if finite == (flag == 0) else " **FLAG DISAGREES WITH THE SCAN**"
            cells.append(f"{err:.2e} / {'yes' if finite else '**NO**'} / {flag}{agree}")
            if (not finite) or err > 1e-3:
                first_bad.setdefault((name, key, m), (sc, "non-finite" if not finite else f"{err:.1e}"))
        print(f"| {sc:g} | {np.abs(ref).max():.3g} | " + " | ".join(cells) + " |")
        sys.stdout.flush()


def main():
    names = sys.argv[1:] or ["hubert_large", "wavlm_large", "hubert_base"]
    print("# The fp16 cliff: outlier writers scaled until a 16-bit mode breaks")
    print()
    print(__doc__.split("usage:")[0].strip())
    first_bad = {}
    for name in names:
        cfg = named_config(name)
        base = synth_weights(cfg, SEED, "pretrained_like")
        pcm = dict(dc=60.0, scale=3000.0) if not cfg.normalize else {}
        wavs = [torch.from_numpy(w).cuda() for w in synth_wavs([32000, 23456], 77, **pcm)]
        print()
        print(f"## {name} (2 utterances: 2.0 s / 1.47 s, weight seed {SEED})")
        print()
        sweep(name, cfg, wavs, [(sc, scale_outlier_writers(cfg, base, SEED, sc)) for sc in SCALES], "writers x", first_bad, "writers")
        print()
        print("fc1 (weight and bias) scaled: the FFN's hidden activation — an fp16 operand — grows in proportion")
        print()
        sweep(name, cfg, wavs, [(sc, scale_ffn_hidden(base, sc)) for sc in FFN_SCALES], "fc1 x", first_bad, "fc1")
    print()
    print("## The cliff: first scale with a state more than 1e-3 from fp32, or non-finite")
    print()
    print("| model | sweep | " + " | ".join(MODES) + " |")
    print("|---|---|" + "---|" * len(MODES))
    for name in names:
        for key, top in (("writers", SCALES[-1]), ("fc1", FFN_SCALES[-1])):
            row = []
            for m in MODES:
                fb = first_bad.get((name, key, m))
                row.append(f"x{fb[0]:g} ({fb[1]})" if fb else f"none up to x{top:g}")
            print(f"| {name} | {key} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
