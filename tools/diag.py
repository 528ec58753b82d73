#!/usr/bin/env python3
"""GPU-side diagnostic: per-stage error of the HIP encoder vs the oracle, and a first timing."""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import encoder_oracle as O
from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import named_config, synth_weights, synth_wavs
from conftest import load_golden


def stage_report(name, dtype="fp32"):
    meta, cfg, weights, wavs, golden, norms = load_golden(name)
    enc = HipEncoder(cfg, weights, dtype=dtype)
    hs = enc.forward([torch.from_numpy(w).cuda() for w in wavs])
    torch.cuda.synchronize()
    hs = hs.cpu().numpy()
    taps = {}
    ref = O.forward(cfg, weights, wavs, dtype=np.float64, taps=taps)
    print(f"== {name} [{dtype}] finite={np.isfinite(hs).all()}")
    for i in range(len(cfg.conv_layers) - 3, len(cfg.conv_layers)):
        got = enc.debug_tap(f"conv{i}").reshape(taps[f"conv{i}"].shape)
        print(f"   conv{i}: {O.rel_err(got, taps[f'conv{i}']):.3e}")
    got = enc.debug_tap("proj").reshape(taps["proj"].shape)
    print(f"   proj : {O.rel_err(got, taps['proj']):.3e}")
    print("   hs vs oracle64:", " ".join(f"{O.rel_err(hs[l], r):.2e}" for l, r in enumerate(ref)))
    ts, cs = meta["t_stride"], meta["c_stride"]
    print("   hs vs golden  :", " ".join(f"{O.rel_err(hs[l][:, ::ts, ::cs], g):.2e}" for l, g in enumerate(golden)))
    enc.close()


def timing(cfg_name, B, secs, dtype, iters=3):
    cfg = named_config(cfg_name)
    weights = synth_weights(cfg, 0)
    enc = HipEncoder(cfg, weights, dtype=dtype)
    g = torch.Generator(device="cuda").manual_seed(1234)
    wavs = [torch.randn(int(secs * 16000), device="cuda", generator=g) for _ in range(B)]
    out = enc.forward(wavs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        enc.forward(wavs, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    T = out.shape[2]
    print(f"== timing {cfg_name} {dtype} B={B} {secs}s: {dt*1e3:.2f} ms/batch, {B*T/dt:.0f} frames/s, finite={torch.isfinite(out).all().item()}")
    enc.profile_enable(True)
    enc.forward(wavs, out=out)
    prof = enc.profile_read()
    enc.profile_enable(False)
    tot = sum(p["ms"] for p in prof)
    for p in sorted(prof, key=lambda p: -p["ms"]):
        tf = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0
        gb = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0
        print(f"   {p['name']:18s} n={p['launches']:3d} {p['ms']:8.3f} ms {100*p['ms']/tot:5.1f}%  {tf:7.1f} TF/s {gb:8.1f} GB/s")
    enc.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["stages", "time"]
    if "stages" in what:
        for n in ["tiny_hubert_pad", "tiny_hubert_large_pad", "tiny_wavlm_large_pad", "hubert_base_pseudo"]:
            try:
                stage_report(n)
            except Exception as ex:
                print(f"!! {n}: {type(ex).__name__}: {ex}")
    if "stages16" in what:
        for n in ["tiny_hubert_pad", "hubert_base_pseudo"]:
            for d in ("bf16", "fp16"):
                try:
                    stage_report(n, d)
                except Exception as ex:
                    print(f"!! {n}/{d}: {type(ex).__name__}: {ex}")
    if "stagesx3" in what:
        for n in ["tiny_hubert_pad", "hubert_base_pseudo", "hubert_large_pseudo", "wavlm_large_pseudo"]:
            try:
                stage_report(n, "fp32x3")
            except Exception as ex:
                print(f"!! {n}/fp32x3: {type(ex).__name__}: {ex}")
    if "time" in what:
        for d in ("fp32", "bf16", "fp16"):
            try:
                timing("hubert_base", 32, 10.0, d)
            except Exception as ex:
                print(f"!! timing {d}: {type(ex).__name__}: {ex}")
