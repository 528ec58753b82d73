#!/usr/bin/env python3
"""Per-layer parity of the HIP encoder against the reference-generated goldens (tests/golden/*.npz), every operand
mode, as a markdown table (GPU box).  usage: parity_table.py [golden names ...] > profiles/rNN_parity.md"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_golden
from oracle import encoder_oracle as O
from s3prl_amd.encoder import HipEncoder

names = sys.argv[1:] or ["hubert_base_pseudo", "wav2vec2_base_pseudo", "wavlm_base_plus_pseudo", "distilhubert_pseudo",
                         "unispeech_sat_base_pseudo", "data2vec_base_pseudo", "multires_hubert_base_pseudo", "hubert_large_10s",
                         "wavlm_large_15s_pad", "hubert_base_pl", "wav2vec2_base_pl", "hubert_large_pl", "wavlm_large_pl",
                         "hubert_base_10s_pl", "hubert_large_10s_pl", "wavlm_large_15s_pl"]
print("# Parity of the HIP encoder vs outputs of the reference itself (tests/golden, PyTorch CPU fp32), per operand mode")
print()
print("max / mean over the hidden states of the per-layer relative error ||h - h_ref||_F / ||h_ref||_F (SURVEY §8d); target 1e-3.")
print()
print("`ref vs fp64`: the reference's own distance from an fp64 evaluation of the same network (oracle/encoder_oracle.py in float64) —")
print("the conditioning floor of the fixture: no fp32 implementation can be expected closer to the reference than about this.")
print("`*_pl` fixtures: synth_weights(profile=\"pretrained_like\") (outlier channels, large LayerNorm gains, Student-t matrices, ...).")
print()
print("| fixture | shape | ref vs fp64 | fp32 | fp32x3 | fp16x2 | fp16 | bf16 |")
print("|---|---|---:|---:|---:|---:|---:|---:|")
for name in names:
    meta, cfg, weights, wavs, golden, _ = load_golden(name)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    ts, cs = meta["t_stride"], meta["c_stride"]
    cells = []
    if meta.get("selection") or meta["config"].startswith("multires") or max(meta["lengths"]) > 40000:
        cells.append("—")  # (the numpy fp64 evaluation is for the short plain-encoder fixtures)
    else:
        h64 = O.forward(cfg, weights, wavs, dtype=np.float64)
        e64 = [O.rel_err(golden[l], h64[l][:, ::ts, ::cs]) for l in range(len(golden))]
        cells.append(f"{max(e64):.2e}")
    for mode in ("fp32", "fp32x3", "fp16x2", "fp16", "bf16"):
        enc = HipEncoder(cfg, weights, dtype=mode)
        hs = enc.forward(dev, selection=meta.get("selection")).cpu().numpy()
        errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
        cells.append(f"{max(errs):.2e} / {np.mean(errs):.2e}")
        enc.close()
    print(f"| `{name}` ({meta['config']}, lengths {meta['lengths']}) | {len(golden)} x {tuple(meta['shape'])} | " + " | ".join(cells) + " |")
