#!/usr/bin/env python3
"""Op-level twin of tools/two_stream_probe.py: the SAME GEMM (private buffers per stream) on S streams at once, each result compared
with the quiet result of that stream's operands — names the kernel whose bits depend on what else the GPU is doing.

usage (GPU box): python tools/op_concurrency_probe.py [--dtype bf16] [--streams 4] [--rounds 12] [--tune KEY=INT ...]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name: (batches, M, N, K, row stride of A in elements (0 = K), act, residual, fp32 out)
SHAPES = {
    "conv2 (implicit GEMM, overlapping rows, GELU, 16-bit out)": (8, 7999, 512, 1536, 1024, 1, False, False),
    "conv6 (k = 2)": (8, 499, 512, 1024, 1024, 1, False, False),
    "q|k|v (plain, 16-bit out)": (1, 3992, 2304, 768, 0, 0, False, False),
    "fc1 (GELU, 16-bit out)": (1, 3992, 3072, 768, 0, 1, False, False),
    "fc2 (fp32 out + residual)": (1, 3992, 768, 3072, 0, 0, True, True),
    "proj (fp32 out)": (1, 3992, 768, 512, 0, 0, False, True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT")
    ap.add_argument("--mixed", action="store_true", help="a DIFFERENT GEMM shape (kernel variant) on every stream, all at once")
    args = ap.parse_args()
    import torch

    from s3prl_amd import _lib

    lib = _lib.load()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(lib.s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    dev = torch.device("cuda", 0)
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    S = args.streams
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if args.mixed:
        names = list(SHAPES)
        jobs = []
        for s in range(S):
            nb, M, N, K, lda, act, use_res, f32out = SHAPES[names[s % len(names)]]
            lda = lda or K
            span = (M - 1) * lda + K
            g = torch.Generator(device=dev).manual_seed(100 + s)
            A = torch.randn((nb, span), device=dev, generator=g).to(tdt)
            W = (torch.randn((N, K), device=dev, generator=g) / K ** 0.5).to(tdt)
            bias = torch.randn(N, device=dev, generator=g)
            res = torch.randn((nb, M, N), device=dev, generator=g) if use_res else None
            jobs.append((names[s % len(names)], nb, M, N, K, lda, span, act, f32out, A, W, bias, res))

        def go(s, st):
            name, nb, M, N, K, lda, span, act, f32out, A, W, bias, res = jobs[s]
            o32 = torch.empty((nb, M, N), device=dev) if (f32out or args.dtype == "fp32") else None
            o16 = torch.empty((nb, M, N), device=dev, dtype=tdt) if o32 is None else None
            rc = lib.s3enc_op_gemm(_lib.DTYPES[args.dtype], ptr(A), lda, span, ptr(W), ptr(bias), M, N, K, nb, act, ptr(res), None,
                                   ptr(o32), ptr(o16), N, M * N, C.c_void_p(st.cuda_stream))
            _lib.check(rc, "s3enc_op_gemm")
            return o32 if o32 is not None else o16

        quiet = []
        for s in range(S):
            quiet.append(go(s, streams[0]))
            torch.cuda.synchronize()
        bad = [0] * S
        for _ in range(args.rounds):
            for rep in range(4):
                outs = [go(s, streams[s]) for s in range(S)]
            torch.cuda.synchronize()
            for s in range(S):
                bad[s] += int(not torch.equal(outs[s], quiet[s]))
        print(json.dumps({"dtype": args.dtype, "tune": args.tune, "mixed": [j[0][:12] for j in jobs], "rounds": args.rounds,
                          "rounds whose result differs from the quiet run, per stream": bad}), flush=True)
        return
    for name, (nb, M, N, K, lda, act, use_res, f32out) in SHAPES.items():
        lda = lda or K
        span = (M - 1) * lda + K
        sets = []
        for s in range(S):
            g = torch.Generator(device=dev).manual_seed(100 + s)
            A = torch.randn((nb, span), device=dev, generator=g).to(tdt)
            W = (torch.randn((N, K), device=dev, generator=g) / K ** 0.5).to(tdt)
            bias = torch.randn(N, device=dev, generator=g)
            res = torch.randn((nb, M, N), device=dev, generator=g) if use_res else None
            sets.append((A, W, bias, res))

        def launch(s, st):
            A, W, bias, res = sets[s]
            o32 = torch.empty((nb, M, N), device=dev) if (f32out or args.dtype == "fp32") else None
            o16 = torch.empty((nb, M, N), device=dev, dtype=tdt) if o32 is None else None
            rc = lib.s3enc_op_gemm(_lib.DTYPES[args.dtype], ptr(A), lda, span, ptr(W), ptr(bias), M, N, K, nb, act, ptr(res), None,
                                   ptr(o32), ptr(o16), N, M * N, C.c_void_p(st.cuda_stream))
            _lib.check(rc, "s3enc_op_gemm")
            return o32 if o32 is not None else o16

        quiet = []
        for s in range(S):
            with torch.cuda.stream(streams[0]):
                quiet.append(launch(s, streams[0]))
            torch.cuda.synchronize()
        bad, worst = 0, 0.0
        for _ in range(args.rounds):
            outs = []
            for rep in range(3):  # a few launches per stream so that the streams really overlap
                outs = []
                for s in range(S):
                    with torch.cuda.stream(streams[s]):
                        outs.append(launch(s, streams[s]))
            torch.cuda.synchronize()
            for s in range(S):
                if not torch.equal(outs[s], quiet[s]):
                    bad += 1
                    worst = max(worst, float((outs[s].float() - quiet[s].float()).abs().max()))
        print(json.dumps({"dtype": args.dtype, "tune": args.tune, "gemm": name, "streams": S, "results compared": args.rounds * S,
                          "results that differ from the quiet run": bad, "max abs diff": worst}), flush=True)


if __name__ == "__main__":
    main()
