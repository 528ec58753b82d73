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
                         "wavlm_large_15s_pad"]
print("# Parity of the HIP encoder vs outputs of the reference itself (tests/golden, PyTorch CPU fp32), per operand mode")
print()
print("max / mean over the hidden states of the per-layer relative error ||h - h_ref||_F / ||h_ref||_F (SURVEY §8d); target 1e-3.")
print()
print("| fixture | shape | fp32 | fp32x3 | fp16x2 | fp16 | bf16 |")
print("|---|---|---:|---:|---:|---:|---:|")
for name in names:
    meta, cfg, weights, wavs, golden, _ = load_golden(name)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    ts, cs = meta["t_stride"], meta["c_stride"]
    cells = []
    for mode in ("fp32", "fp32x3", "fp16x2", "fp16", "bf16"):
        enc = HipEncoder(cfg, weights, dtype=mode)
        hs = enc.forward(dev, selection=meta.get("selection")).cpu().numpy()
        errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
        cells.append(f"{max(errs):.2e} / {np.mean(errs):.2e}")
        enc.close()
    print(f"| `{name}` ({meta['config']}, lengths {meta['lengths']}) | {len(golden)} x {tuple(meta['shape'])} | " + " | ".join(cells) + " |")
