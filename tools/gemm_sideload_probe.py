#!/usr/bin/env python3
"""A conv-stack GEMM of the 16-bit modes (implicit GEMM over a 262 MB activation that streams from HBM) on one stream while OTHER streams
run HBM-bound kernels — torch elementwise passes, the library's row LayerNorm, its attention — instead of more GEMMs: does the GEMM keep
its bits when its LDS-DMA pieces come back late?  (tools/op_concurrency_probe.py only ever ran GEMMs beside GEMMs.)

usage (GPU box): python tools/gemm_sideload_probe.py [--dtype bf16] [--side eltwise ln attention] [--rounds 10] [--tune KEY=INT]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {  # batches, M, N, K, lda, act
    "conv1": (8, 15999, 512, 1536, 1024, 1),
    "conv2": (8, 7999, 512, 1536, 1024, 1),
    "conv4": (8, 1999, 512, 1536, 1024, 1),
    "fc1": (1, 3992, 3072, 768, 768, 1),
    "qkv": (1, 3992, 2304, 768, 768, 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--side", nargs="+", default=["none", "eltwise", "ln", "attention", "all"])
    ap.add_argument("--shapes", nargs="+", default=list(SHAPES))
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT")
    args = ap.parse_args()
    import torch

    from s3prl_amd import _lib

    lib = _lib.load()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(lib.s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    dev = torch.device("cuda", 0)
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    DT = _lib.DTYPES[args.dtype]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    main_st = torch.cuda.Stream(device=dev)
    side_st = [torch.cuda.Stream(device=dev) for _ in range(3)]
    g = torch.Generator(device=dev).manual_seed(3)
    big = [torch.randn(96 << 20, device=dev, generator=g) for _ in range(3)]
    rows = 262144
    lx = torch.randn((rows, 768), device=dev, generator=g)
    lg, lb = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    l16 = torch.empty((rows, 768), device=dev, dtype=tdt)
    Bq, Tq, Hq = 32, 499, 12
    qkv = torch.randn((Bq * Tq, 3 * 768), device=dev, generator=g).to(tdt)
    att = torch.empty((Bq * Tq, 768), device=dev, dtype=tdt)
    valid = torch.full((Bq,), Tq, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def side(kind, n):
        kinds = ["eltwise", "ln", "attention"] if kind == "all" else [kind]
        for i, st in enumerate(side_st):
            k = kinds[i % len(kinds)]
            with torch.cuda.stream(st):
                for _ in range(n):
                    if k == "eltwise":
                        big[i].mul_(1.0000001)
                    elif k == "ln":
                        _lib.check(lib.s3enc_op_layernorm(DT, ptr(lx), ptr(lg), ptr(lb), rows, 768, 0, None, ptr(l16), C.c_void_p(st.cuda_stream)), "ln")
                    elif k == "attention":
                        _lib.check(lib.s3enc_op_attention(DT, ptr(qkv), ptr(att), ptr(valid), Bq, Tq, Hq, None, 0, None, C.c_void_p(st.cuda_stream)),
                                   "attention")

    for name in args.shapes:
        nb, M, N, K, lda, act = SHAPES[name]
        span = (M - 1) * lda + K
        A = torch.randn((nb, span), device=dev, generator=g).to(tdt)
        W = (torch.randn((N, K), device=dev, generator=g) / K ** 0.5).to(tdt)
        out = torch.empty((nb, M, N), device=dev, dtype=tdt)
        torch.cuda.synchronize()

        def gemm():
            rc = lib.s3enc_op_gemm(DT, ptr(A), lda, span, ptr(W), None, M, N, K, nb, act, None, None, None, ptr(out), N, M * N,
                                   C.c_void_p(main_st.cuda_stream))
            _lib.check(rc, "s3enc_op_gemm")

        gemm()
        torch.cuda.synchronize()
        ref = out.clone()
        torch.cuda.synchronize()
        for kind in args.side:
            bad, rows_bad = 0, 0
            for _ in range(args.rounds):
                if kind != "none":
                    side(kind, 12)
                out.zero_()
                main_st.wait_stream(torch.cuda.current_stream(dev))  # (zero_ ran on the default stream)
                gemm()
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad += 1
                    rows_bad += int(((out.float() - ref.float()).abs().amax(dim=2) > 0).sum())
            print(json.dumps({"gemm": name, "dtype": args.dtype, "tune": args.tune, "beside": kind, "rounds": args.rounds,
                              "rounds whose result differs from the quiet run": bad, "differing rows in all": rows_bad}), flush=True)


if __name__ == "__main__":
    main()
