#!/usr/bin/env python3
"""Latency of small-batch forwards: eager launches vs hipGraph replay (s3enc_set_graph_replay), as a markdown table.
usage (GPU box): python tools/graph_latency.py > profiles/rNN_graph_replay.md"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import named_config, synth_weights

SHAPES = [(1, 2.0), (1, 10.0), (4, 5.0), (8, 10.0), (32, 10.0)]


def timed(enc, wavs, out, iters, sync_each):
    for _ in range(4):
        enc.forward(wavs, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        enc.forward(wavs, out=out)
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


print("# hipGraph replay vs eager launches (s3enc_set_graph_replay), one MI355X")
print()
print("ms per forward: `latency` = one forward at a time (synchronise after each: what an online feature server sees), "
      "`throughput` = back-to-back forwards (the queue never drains).  The graph is captured on the second forward of a "
      "shape and replayed from the third; outputs are bit-identical to eager (tests/test_graph_gpu.py).")
print()
print("| model | dtype | batch | eager latency | graph latency | eager throughput | graph throughput |")
print("|---|---|---|---:|---:|---:|---:|")
for model, dtype in (("hubert_base", "bf16"), ("hubert_base", "fp32"), ("wavlm_large", "bf16"), ("multires_hubert_base", "bf16")):
    cfg = named_config(model)
    enc = HipEncoder(cfg, synth_weights(cfg, 0), dtype=dtype)
    for B, secs in SHAPES:
        if dtype == "fp32" and B > 8:
            continue
        wavs = [torch.randn(int(secs * 16000), device="cuda") for _ in range(B)]
        out = enc.forward(wavs)
        iters = 200 if B * secs <= 20 else 40
        enc.graph_replay(False)
        e_lat, e_thr = timed(enc, wavs, out, iters, True), timed(enc, wavs, out, iters, False)
        enc.graph_replay(True)
        g_lat, g_thr = timed(enc, wavs, out, iters, True), timed(enc, wavs, out, iters, False)
        st = enc.graph_stats()
        print(f"| {model} | {dtype} | {B} x {secs:g} s | {e_lat:.3f} | {g_lat:.3f} | {e_thr:.3f} | {g_thr:.3f} |", flush=True)
        enc.graph_replay(False)
    enc.close()
