#!/usr/bin/env python3
"""A producer -> consumer CHAIN of 16-bit GEMMs per stream (ping-pong buffers, like the conv stack of the encoder), S streams at once:
is every link's input complete and visible when the next kernel of the same stream reads it, whatever the other streams do?
Each stream's final buffer is compared with the quiet (one stream at a time) result.  `--torch` runs the same experiment with
torch.mm + gelu (rocBLAS / hipBLASLt kernels) instead of the library's kernels.

usage (GPU box): python tools/chain_concurrency_probe.py [--streams 4] [--links 6] [--rounds 10] [--tune KEY=INT] [--torch]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--links", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--rows", type=int, default=16000)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--torch", action="store_true")
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
    S, M, W_ = args.streams, args.rows, args.width
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    jobs = []
    for s in range(S):
        g = torch.Generator(device=dev).manual_seed(7 + s)
        x0 = torch.randn((M, W_), device=dev, generator=g).to(tdt)
        ws = [(torch.randn((W_, W_), device=dev, generator=g) * (1.7 / W_ ** 0.5)).to(tdt) for _ in range(args.links)]
        bufs = [torch.empty((M, W_), device=dev, dtype=tdt) for _ in range(2)]
        jobs.append((x0, ws, bufs))

    def chain(s, st):
        x0, ws, bufs = jobs[s]
        cur = x0
        with torch.cuda.stream(st):
            for l, w in enumerate(ws):
                dst = bufs[l & 1]
                if args.torch:
                    y = torch.mm(cur, w.t())
                    dst.copy_(torch.nn.functional.gelu(y) if args.act else y)
                else:
                    rc = lib.s3enc_op_gemm(_lib.DTYPES[args.dtype], ptr(cur), W_, M * W_, ptr(w), None, M, W_, W_, 1, args.act, None, None,
                                           None, ptr(dst), W_, M * W_, C.c_void_p(st.cuda_stream))
                    _lib.check(rc, "s3enc_op_gemm")
                cur = dst
        return cur

    quiet = []
    for s in range(S):
        torch.cuda.synchronize()  # (the operands were made on the default stream)
        last = chain(s, streams[0])
        torch.cuda.synchronize()  # (... and clone() runs there too)
        quiet.append(last.clone())
        torch.cuda.synchronize()
    bad = [0] * S
    rows_bad = 0
    for _ in range(args.rounds):
        for rep in range(3):
            outs = [chain(s, streams[s]) for s in range(S)]
        torch.cuda.synchronize()
        for s in range(S):
            if not torch.equal(outs[s], quiet[s]):
                bad[s] += 1
                rows_bad += int(((outs[s].float() - quiet[s].float()).abs().amax(dim=1) > 0).sum())
    print(json.dumps({"kernels": "torch.mm + gelu" if args.torch else "libs3enc gemm", "dtype": args.dtype, "tune": args.tune, "streams": S,
                      "links": args.links, "act": args.act, "rounds": args.rounds, "rounds whose final buffer differs, per stream": bad,
                      "differing rows in all": rows_bad}), flush=True)


if __name__ == "__main__":
    main()
