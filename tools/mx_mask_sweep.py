#!/usr/bin/env python3
"""fp16x2 parity per GEMM that takes its second weight term on the MX pipe (tuning key gemm16_mx: 1 conv1, 2 q|k|v, 4 fc1, 8 fc2).
(GPU box.)  usage: mx_mask_sweep.py [golden names ...] >> profiles/rNN_mx_second_term.md"""
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

MASKS = [(0, "two fp16 terms"), (1, "conv1"), (2, "q\\|k\\|v"), (4, "fc1"), (8, "fc2"), (14, "q\\|k\\|v + fc1 + fc2"), (15, "all four")]
names = sys.argv[1:] or ["hubert_base_pseudo", "distilhubert_pseudo", "data2vec_base_pseudo", "hubert_base_pl", "wav2vec2_base_pl",
                         "hubert_large_pl", "wavlm_large_pl", "hubert_base_10s_pl", "hubert_large_10s_pl", "wavlm_large_15s_pl"]
lib = _lib.load()
print("max over the hidden states of the relative error against the reference's golden, compute dtype fp16x2, per MX mask")
print()
print("| fixture | " + " | ".join(f"{m}: {d}" for m, d in MASKS) + " |")
print("|---|" + "---:|" * len(MASKS))
for name in names:
    meta, cfg, weights, wavs, golden, _ = load_golden(name)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    ts, cs = meta["t_stride"], meta["c_stride"]
    enc = HipEncoder(cfg, weights, dtype="fp16x2")
    cells = []
    for m, _d in MASKS:
        _lib.check(lib.s3enc_set_tuning(b"gemm16_mx", m))
        hs = enc.forward(dev, selection=meta.get("selection")).cpu().numpy()
        cells.append("%.2e" % max(O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))))
    enc.close()
    print(f"| `{name}` | " + " | ".join(cells) + " |")
    sys.stdout.flush()
_lib.check(lib.s3enc_set_tuning(b"gemm16_mx", 14))
