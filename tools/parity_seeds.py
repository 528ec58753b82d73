#!/usr/bin/env python3
"""The fp16x2 mode's "inside 1e-3" claim on a DISTRIBUTION of weight seeds (GPU box).  Every `*_s<seed>_pl` fixture of
tests/golden (reference-generated: tests/golden/make_golden.py, SEED_MODELS — weight seeds 1-6 x five pretrained-like models on
1-2 s ragged inputs, WavLM-large seeds 2-3 at the 15 s ragged shape) plus the seed-0 / seed-1 `*_pl` fixtures of earlier rounds, in
fp32, fp32x3 and fp16x2: per fixture the max per-layer relative error, then max / median per model.
usage: parity_seeds.py [modes ...] > profiles/rNN_parity_seeds.md"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import golden_names, load_golden
from oracle import encoder_oracle as O
from s3prl_amd.encoder import HipEncoder

modes = sys.argv[1:] or ["fp32", "fp32x3", "fp16x2"]
names = [n for n in golden_names() if n.endswith("_pl") and not n.startswith("tiny_")]
rows, by_model = [], {}
for name in names:
    meta, cfg, weights, wavs, golden, _ = load_golden(name)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    ts, cs = meta["t_stride"], meta["c_stride"]
    errs = {}
    for mode in modes:
        enc = HipEncoder(cfg, weights, dtype=mode)
        hs = enc.forward(dev).cpu().numpy()
        assert np.isfinite(hs).all(), (name, mode)
        errs[mode] = max(O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden)))
        enc.close()
    rows.append((name, meta, errs))
    by_model.setdefault(meta["config"], []).append(errs)
print("# fp16x2 (and fp32 / fp32x3) parity over weight seeds — pretrained-like statistics, reference-generated goldens")
print()
print("Max over the hidden states of the per-layer relative error ||h - h_ref||_F / ||h_ref||_F; tolerance of the path: 1e-3.")
print("Fixtures: `tests/golden/*_pl.npz` at the models' full dimensions (weight seed in the table; `make_golden.py` runs the reference).")
print()
print("| model | fixtures | " + " | ".join(f"{m} max | {m} median" for m in modes) + " |")
print("|---|---:|" + "---:|" * (2 * len(modes)))
for model, es in sorted(by_model.items()):
    cells = []
    for m in modes:
        v = np.array([e[m] for e in es])
        cells += [f"{v.max():.2e}", f"{np.median(v):.2e}"]
    print(f"| `{model}` | {len(es)} | " + " | ".join(cells) + " |")
allv = {m: np.array([e[m] for _, _, e in rows]) for m in modes}
print(f"| **all** | {len(rows)} | " + " | ".join(f"**{allv[m].max():.2e}** | {np.median(allv[m]):.2e}" for m in modes) + " |")
print()
print("| fixture | weight seed | lengths | " + " | ".join(modes) + " |")
print("|---|---:|---|" + "---:|" * len(modes))
for name, meta, errs in rows:
    print(f"| `{name}` | {meta['weight_seed']} | {meta['lengths']} | " + " | ".join(f"{errs[m]:.2e}" for m in modes) + " |")
