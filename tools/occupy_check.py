#!/usr/bin/env python3
"""Sanity of the CU-contention proxy (GPU box): does s3enc_debug_occupy_cus hold its workgroups for the time asked, and what does a
bf16 forward cost while K of them are resident?  usage: occupy_check.py >> profiles/rNN_cu_contention.md"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s3prl_amd import _lib
from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import named_config, synth_wavs, synth_weights

lib = _lib.load()
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("\n## sanity of the proxy (`tools/occupy_check.py`)\n")
for k, threads in ((32, 256), (256, 256), (256, 1024)):
    with torch.cuda.stream(side):
        e0.record(side)
        _lib.check(lib.s3enc_debug_occupy_cus(k, threads, 20.0, side.cuda_stream))
        e1.record(side)
    torch.cuda.synchronize()
    print(f"* {k} workgroups x {threads} threads asked to stay 20.0 ms: the launch took {e0.elapsed_time(e1):.2f} ms")
cfg = named_config("hubert_base")
enc = HipEncoder(cfg, synth_weights(cfg, 0), dtype="bf16")
wavs = [torch.from_numpy(w).cuda() for w in synth_wavs([160000] * 32, 5)]
for _ in range(3):
    enc.forward(wavs)
torch.cuda.synchronize()


def forward_ms(k, threads):
    if k:
        _lib.check(lib.s3enc_debug_occupy_cus(k, threads, 200.0, side.cuda_stream))
        time.sleep(0.02)  # the idle workgroups are resident before the forward's first kernel arrives
    cur = torch.cuda.current_stream(dev)
    e0.record(cur)
    for _ in range(5):
        enc.forward(wavs)
    e1.record(cur)
    cur.synchronize()
    ms = e0.elapsed_time(e1) / 5
    torch.cuda.synchronize()
    return ms


print()
print("| idle workgroups resident BEFORE the forward starts | bf16 forward, ms |")
print("|---|---:|")
for k, threads in ((0, 0), (8, 256), (32, 256), (64, 256), (256, 256), (256, 1024), (1024, 1024)):
    print(f"| {k} x {threads} threads | {forward_ms(k, threads):.3f} |")
