#!/usr/bin/env python3
"""One transformer layer's kernels as a per-stream CHAIN at op level (LayerNorm -> q|k|v -> attention -> out_proj + residual -> LayerNorm
-> fc1 GELU -> fc2 + residual), L layers deep, on S streams at once (private buffers per stream); every stream's final residual stream
is compared bit for bit with its quiet (one stream at a time) result.  `--skip` leaves kernels out of the chain (their output buffer
keeps its quiet content) to find the one whose bits depend on what else the GPU is doing.

usage (GPU box): python tools/layer_concurrency_probe.py [--dtype bf16] [--streams 4] [--layers 4] [--skip attention ln]"""
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
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=499)
    ap.add_argument("--skip", nargs="*", default=[], choices=["ln", "qkv", "attention", "out_proj", "fc1", "fc2"])
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
    S, B, T, D, F, H = args.streams, args.batch, args.frames, 768, 3072, 12
    M = B * T
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    jobs = []
    for s in range(S):
        g = torch.Generator(device=dev).manual_seed(31 + s)
        rn = lambda *sh, sc=1.0: torch.randn(sh, device=dev, generator=g) * sc
        j = {"x0": rn(M, D), "valid": torch.full((B,), T, dtype=torch.int32, device=dev), "layers": []}
        for _ in range(args.layers):
            j["layers"].append({"g1": 1 + rn(D, sc=0.1), "b1": rn(D, sc=0.1), "g2": 1 + rn(D, sc=0.1), "b2": rn(D, sc=0.1),
                                "wqkv": rn(3 * D, D, sc=D ** -0.5).to(tdt), "bqkv": rn(3 * D, sc=0.1),
                                "wo": rn(D, D, sc=0.5 * D ** -0.5).to(tdt), "bo": rn(D, sc=0.1),
                                "w1": rn(F, D, sc=D ** -0.5).to(tdt), "bf1": rn(F, sc=0.1),
                                "w2": rn(D, F, sc=0.5 * F ** -0.5).to(tdt), "bf2": rn(D, sc=0.1)})
        z16 = lambda n: torch.zeros((M, n), device=dev, dtype=tdt)
        j.update(xa=torch.zeros((M, D), device=dev), xb=torch.zeros((M, D), device=dev), xT=z16(D), qkv=z16(3 * D), att=z16(D), h=z16(F))
        jobs.append(j)
    torch.cuda.synchronize()

    def chain(s, st):
        j = jobs[s]
        sp = C.c_void_p(st.cuda_stream)
        ck = lambda rc, what: _lib.check(rc, what)
        cur = j["x0"]
        for L in j["layers"]:
            if "ln" not in args.skip:
                ck(lib.s3enc_op_layernorm(DT, ptr(cur), ptr(L["g1"]), ptr(L["b1"]), M, D, 0, None, ptr(j["xT"]), sp), "ln1")
            if "qkv" not in args.skip:
                ck(lib.s3enc_op_gemm(DT, ptr(j["xT"]), D, M * D, ptr(L["wqkv"]), ptr(L["bqkv"]), M, 3 * D, D, 1, 0, None, None, None,
                                     ptr(j["qkv"]), 3 * D, M * 3 * D, sp), "qkv")
            if "attention" not in args.skip:
                ck(lib.s3enc_op_attention(DT, ptr(j["qkv"]), ptr(j["att"]), ptr(j["valid"]), B, T, H, None, 0, None, sp), "attention")
            if "out_proj" not in args.skip:
                ck(lib.s3enc_op_gemm(DT, ptr(j["att"]), D, M * D, ptr(L["wo"]), ptr(L["bo"]), M, D, D, 1, 0, ptr(cur), None, ptr(j["xa"]),
                                     None, D, M * D, sp), "out_proj")
            if "ln" not in args.skip:
                ck(lib.s3enc_op_layernorm(DT, ptr(j["xa"]), ptr(L["g2"]), ptr(L["b2"]), M, D, 0, None, ptr(j["xT"]), sp), "ln2")
            if "fc1" not in args.skip:
                ck(lib.s3enc_op_gemm(DT, ptr(j["xT"]), D, M * D, ptr(L["w1"]), ptr(L["bf1"]), M, F, D, 1, 1, None, None, None, ptr(j["h"]),
                                     F, M * F, sp), "fc1")
            if "fc2" not in args.skip:
                ck(lib.s3enc_op_gemm(DT, ptr(j["h"]), F, M * F, ptr(L["w2"]), ptr(L["bf2"]), M, D, F, 1, 0, ptr(j["xa"]), None, ptr(j["xb"]),
                                     None, D, M * D, sp), "fc2")
            cur = j["xb"]
        return j

    names = ["xT", "qkv", "att", "xa", "h", "xb"]
    quiet = []
    for s in range(S):
        j = chain(s, streams[0])
        torch.cuda.synchronize()
        quiet.append({n: j[n].clone() for n in names})
        torch.cuda.synchronize()
    bad = {n: [0] * S for n in names}
    for _ in range(args.rounds):
        for rep in range(2):
            for s in range(S):
                chain(s, streams[s])
        torch.cuda.synchronize()
        for s in range(S):
            for n in names:
                bad[n][s] += int(not torch.equal(jobs[s][n], quiet[s][n]))
    print(json.dumps({"dtype": args.dtype, "tune": args.tune, "streams": S, "layers": args.layers, "skip": args.skip, "rounds": args.rounds,
                      "rounds in which a buffer's final content differs from the quiet run, per stream": bad}), flush=True)


if __name__ == "__main__":
    main()
