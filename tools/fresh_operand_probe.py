#!/usr/bin/env python3
"""Per stream: a producer kernel REWRITES the GEMM's A operand (a 65-260 MB activation, like conv0 -> conv1 in the encoder) and the 16-bit
GEMM of the library reads it right behind, S streams at once with private buffers.  Every result is compared with the quiet run.
tools/op_concurrency_probe.py multiplied STATIC operands only.

usage (GPU box): python tools/fresh_operand_probe.py [--dtype bf16] [--streams 4] [--rounds 8] [--producer copy|scale] [--tune KEY=INT]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"conv1": (8, 15999, 512, 1536, 1024, 1), "conv2": (8, 7999, 512, 1536, 1024, 1), "fc1": (1, 3992, 3072, 768, 768, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--shape", default="conv1", choices=list(SHAPES))
    ap.add_argument("--producer", default="copy", choices=["copy", "scale", "none"])
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT")
    args = ap.parse_args()
    import torch

    from s3prl_amd import _lib

    lib = _lib.load()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(lib.s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    dev = torch.device("cuda", 0)
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    DT = _lib.DTYPES[args.dtype]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    nb, M, N, K, lda, act = SHAPES[args.shape]
    span = (M - 1) * lda + K
    S = args.streams
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    jobs = []
    for s in range(S):
        g = torch.Generator(device=dev).manual_seed(11 + s)
        src = torch.randn((nb, span), device=dev, generator=g).to(tdt)
        junk = torch.randn((nb, span), device=dev, generator=g).to(tdt)   # what A holds before the producer runs
        W = (torch.randn((N, K), device=dev, generator=g) / K ** 0.5).to(tdt)
        jobs.append({"src": src, "junk": junk, "A": torch.empty_like(src), "W": W,
                     "out": torch.empty((nb, M, N), device=dev, dtype=tdt if args.dtype != "fp32" else torch.float32)})
    torch.cuda.synchronize()

    def step(s, st):
        j = jobs[s]
        with torch.cuda.stream(st):
            j["A"].copy_(j["junk"])                 # the buffer's previous content (another layer's activation)
            if args.producer == "copy":
                j["A"].copy_(j["src"])
            elif args.producer == "scale":
                torch.mul(j["src"], 1.0, out=j["A"])
            o32 = ptr(j["out"]) if args.dtype == "fp32" else None
            o16 = None if args.dtype == "fp32" else ptr(j["out"])
            rc = lib.s3enc_op_gemm(DT, ptr(j["A"] if args.producer != "none" else j["src"]), lda, span, ptr(j["W"]), None, M, N, K, nb, act, None, None,
                                   o32, o16, N, M * N, C.c_void_p(st.cuda_stream))
            _lib.check(rc, "s3enc_op_gemm")

    quiet = []
    for s in range(S):
        step(s, streams[0])
        torch.cuda.synchronize()
        quiet.append(jobs[s]["out"].clone())
        torch.cuda.synchronize()
    bad, rows_bad = [0] * S, 0
    for _ in range(args.rounds):
        for rep in range(3):
            for s in range(S):
                step(s, streams[s])
        torch.cuda.synchronize()
        for s in range(S):
            if not torch.equal(jobs[s]["out"], quiet[s]):
                bad[s] += 1
                rows_bad += int(((jobs[s]["out"].float() - quiet[s].float()).abs().amax(dim=2) > 0).sum())
    print(json.dumps({"gemm": args.shape, "dtype": args.dtype, "tune": args.tune, "producer": args.producer, "streams": S, "rounds": args.rounds,
                      "rounds whose result differs from the quiet run, per stream": bad, "differing rows in all": rows_bad}), flush=True)


if __name__ == "__main__":
    main()
