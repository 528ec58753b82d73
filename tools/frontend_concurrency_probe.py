#!/usr/bin/env python3
"""The front end (waveform statistics -> GroupNorm lag sums -> gn_final -> conv0 + GroupNorm + GELU, s3enc_op_conv0) from S host threads
at once, each on its own stream and buffers, compared bit for bit with the quiet result.  (The op synchronises its stream and frees its
scratch on return, so the overlap comes from the host threads.)

usage (GPU box): python tools/frontend_concurrency_probe.py [--dtype bf16] [--threads 4] [--rounds 10]"""
import argparse
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--layer-norm", action="store_true", help="the layer_norm extractor's conv0 (per-frame LayerNorm) instead of GroupNorm")
    args = ap.parse_args()
    import torch

    from s3prl_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    S, B, n, Cc, stride = args.threads, args.batch, 160000, 512, 5
    L0 = (n - 10) // stride + 1
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    jobs = []
    for s in range(S):
        g = torch.Generator(device=dev).manual_seed(5 + s)
        wavs = [torch.randn(n, device=dev, generator=g) for _ in range(B)]
        w0 = torch.randn((Cc, 10), device=dev, generator=g) * 0.3
        gam = 1 + 0.1 * torch.randn(Cc, device=dev, generator=g)
        bet = 0.05 * torch.randn(Cc, device=dev, generator=g)
        out = torch.zeros((B, L0, Cc), device=dev, dtype=tdt)
        jobs.append({"wavs": wavs, "w0": w0, "g": gam, "b": bet, "out": out, "stream": torch.cuda.Stream(device=dev),
                     "ptrs": (C.c_void_p * B)(*[w.data_ptr() for w in wavs]), "lens": (C.c_int64 * B)(*[n] * B)})
    torch.cuda.synchronize()

    def call(j):
        gn = (ptr(j["g"]), ptr(j["b"]), None, None) if not args.layer_norm else (None, None, ptr(j["g"]), ptr(j["b"]))
        rc = lib.s3enc_op_conv0(_lib.DTYPES[args.dtype], j["ptrs"], j["lens"], B, 0, int(args.layer_norm), ptr(j["w0"]), None, gn[0], gn[1], gn[2],
                                gn[3], Cc, stride, ptr(j["out"]), C.c_void_p(j["stream"].cuda_stream))
        _lib.check(rc, "s3enc_op_conv0")

    quiet = []
    for j in jobs:
        call(j)
        torch.cuda.synchronize()
        quiet.append(j["out"].clone())
        torch.cuda.synchronize()
    bad = [0] * S
    for _ in range(args.rounds):
        def work(j):
            for _ in range(3):
                call(j)
        ths = [threading.Thread(target=work, args=(j,)) for j in jobs]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        for s, j in enumerate(jobs):
            bad[s] += int(not torch.equal(j["out"], quiet[s]))
    print(json.dumps({"op": "conv0 + " + ("LayerNorm" if args.layer_norm else "GroupNorm") + " + GELU", "dtype": args.dtype, "threads": S,
                      "rounds": args.rounds, "rounds whose output differs from the quiet run, per thread": bad}), flush=True)


if __name__ == "__main__":
    main()
