#!/usr/bin/env python3
"""DESIGN.md §6's measurement table, printed from the committed bench lines (profiles/<tag>_bench_*.json), so that the
table can be re-made after a lease instead of being typed.  usage: tools/design_table.py r06"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [  # (file stem, workload label, mode label)
    ("fp32", "configs[1]* HuBERT-base 32x10 s (the metric)", "fp32"),
    ("cfg1_wav2vec2_base_fp32", "configs[1] wav2vec2-base 32x10 s", "fp32"),
    ("fp32x3", "HuBERT-base 32x10 s", "fp32x3"),
    ("fp16x2", "HuBERT-base 32x10 s", "**fp16x2**"),
    ("fp16", "HuBERT-base 32x10 s", "fp16"),
    ("bf16", "HuBERT-base 32x10 s", "bf16"),
    ("cfg2_hubert_base_b64_fp16x2", "configs[2] HuBERT-base 64x10 s", "**fp16x2**"),
    ("cfg2_hubert_base_b64_bf16", "configs[2]", "bf16"),
    ("cfg3_hubert_large_fp16x2", "configs[3] HuBERT-large 32x10 s", "**fp16x2**"),
    ("cfg3_hubert_large_bf16", "configs[3]", "bf16"),
    ("cfg3_hubert_large_fp32x3", "configs[3]", "fp32x3"),
    ("cfg3_hubert_large_fp32", "configs[3]", "fp32"),
    ("cfg4_wavlm_large_mixed_fp16x2", "configs[4] WavLM-large 32x<=15 s mixed", "**fp16x2**"),
    ("cfg4_wavlm_large_mixed_bf16", "configs[4]", "bf16"),
    ("cfg4_wavlm_large_mixed_fp32x3", "configs[4]", "fp32x3"),
    ("cfg4_wavlm_large_mixed_fp32", "configs[4]", "fp32"),
    ("multires_hubert_base_fp32", "multires-HuBERT-base 32x10 s", "fp32"),
]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    print("| workload | mode | ms / batch | frames/s | path TFLOP/s | dominant GEMM: TFLOP/s, frac of peak | HBM traffic / algorithmic bytes "
          "per launch | parity (max rel err) | clock GHz |")
    print("|---|---|---:|---:|---:|---|---|---:|---:|")
    for stem, work, mode in ROWS:
        path = os.path.join(ROOT, "profiles", f"{tag}_bench_{stem}.json")
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        r = d.get("roofline") or {}
        par = d.get("parity") or {}
        err = par.get("max_layer_rel_err_vs_torch_oracle") or par.get("max_layer_rel_err_vs_numpy_oracle")
        traffic = "—"
        if r.get("traffic") and r.get("algorithmic_bytes"):
            traffic = f"{r['traffic'] / 1e6:.0f} / {r['algorithmic_bytes'] / 1e6:.0f} MB = {r['traffic'] / r['algorithmic_bytes']:.2f}"
        print(f"| {work} | {mode} | {d['ms_per_step']:.2f} | {d['value'] / 1e3:,.1f} k | {d.get('path_tflops', 0):.0f} | "
              f"{r.get('achieved', 0):.0f}, {r.get('frac', 0):.3f} of {r.get('peak', 0):.0f} | {traffic} | "
              f"{err:.2e} | {d.get('clock_ghz') or float('nan'):.2f} |" if err is not None else
              f"| {work} | {mode} | {d['ms_per_step']:.2f} | {d['value'] / 1e3:,.1f} k | {d.get('path_tflops', 0):.0f} | "
              f"{r.get('achieved', 0):.0f}, {r.get('frac', 0):.3f} of {r.get('peak', 0):.0f} | {traffic} | — | "
              f"{d.get('clock_ghz') or float('nan'):.2f} |")


if __name__ == "__main__":
    main()
