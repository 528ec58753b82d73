#!/usr/bin/env python3
"""Forwards of ONE short utterance (the latency-bound serving shape), for `rocprofv3 --kernel-trace --stats`:
usage: small_batch_profile.py [model] [dtype] [batch] [secs] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import named_config, synth_weights

model = sys.argv[1] if len(sys.argv) > 1 else "hubert_base"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 50
cfg = named_config(model)
enc = HipEncoder(cfg, synth_weights(cfg, 0), dtype=dtype)
wavs = [torch.randn(int(secs * 16000), device="cuda") for _ in range(B)]
out = enc.forward(wavs)
for _ in range(5):
    enc.forward(wavs, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    enc.forward(wavs, out=out)
torch.cuda.synchronize()
print(f"{model} {dtype} {B} x {secs:g} s: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per forward", flush=True)
enc.close()
