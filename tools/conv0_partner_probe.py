#!/usr/bin/env python3
"""Which kernel, running beside conv0, makes conv0 compute frames wrong?  (profiles/r06d_concurrent_forwards_exclusions.md)

One encoder handle runs forwards that END behind conv0 (S3ENC_DEBUG_STOP=1: waveform statistics, GroupNorm statistics, conv0) in bursts on
its stream; P other streams run ONE kind of the library's kernels over private buffers in a loop (op level, as tools/layer_concurrency_probe.py
does).  conv0's output after every burst is compared bit for bit with the quiet run.

usage (GPU box): python tools/conv0_partner_probe.py [--dtype bf16] [--partners none ln attention qkv fc1 fc2 all]"""
import argparse
import ctypes as C
import json
import os
import sys

os.environ["S3ENC_DEBUG_STOP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--partners", nargs="+", default=["none", "ln", "attention", "qkv", "fc1", "fc2", "all"])
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--burst", type=int, default=8)
    ap.add_argument("--trials", type=int, default=4)
    ap.add_argument("--reps", type=int, default=6, help="partner chains enqueued per stream and trial")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT")
    args = ap.parse_args()
    import numpy as np
    import torch

    from s3prl_amd import _lib
    from s3prl_amd.synth import named_config, synth_weights
    from s3prl_amd.upstream.base import HipUpstreamExpert

    lib = _lib.load()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(lib.s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    _lib.check(lib.s3enc_set_tuning(b"forward_chain", 0), "s3enc_set_tuning")
    dev = torch.device("cuda", 0)
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    DT = _lib.DTYPES[args.dtype]
    cfg = named_config("hubert_base")

    class Expert(HipUpstreamExpert):
        family = cfg.family

    ex = Expert.from_weights(cfg, synth_weights(cfg, 0), dtype=args.dtype).eval()
    enc = ex._encoder_for(dev)
    n, BA = 160000, 4
    gen = torch.Generator(device=dev).manual_seed(1234)
    wavs = [torch.randn(n, device=dev, generator=gen) for _ in range(BA)]
    main_s = torch.cuda.Stream(device=dev)

    def tap():
        buf = np.empty(64 << 20, dtype=np.float32)
        ne = C.c_int64()
        _lib.check(lib.s3enc_debug_tap(enc._h, b"conv0", buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, C.byref(ne)), "tap")
        return buf[:ne.value].copy().reshape(-1, 512)

    with torch.cuda.stream(main_s):
        enc.forward(wavs)
    torch.cuda.synchronize()
    ref = tap()

    # partner workload: one transformer layer's kernels at op level (HuBERT-base shapes, 8 x 499 frames), private buffers per stream
    S, B, T, D, F, H = args.streams, 8, 499, 768, 3072, 12
    M = B * T
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    jobs = []
    for s in range(S):
        g = torch.Generator(device=dev).manual_seed(31 + s)
        rn = lambda *sh, sc=1.0: torch.randn(sh, device=dev, generator=g) * sc
        z16 = lambda k: torch.zeros((M, k), device=dev, dtype=tdt)
        jobs.append({"x0": rn(M, D), "valid": torch.full((B,), T, dtype=torch.int32, device=dev), "g1": 1 + rn(D, sc=0.1), "b1": rn(D, sc=0.1),
                     "wqkv": rn(3 * D, D, sc=D ** -0.5).to(tdt), "bqkv": rn(3 * D, sc=0.1), "w1": rn(F, D, sc=D ** -0.5).to(tdt), "bf1": rn(F, sc=0.1),
                     "w2": rn(D, F, sc=0.5 * F ** -0.5).to(tdt), "bf2": rn(D, sc=0.1), "xa": rn(M, D), "xb": torch.zeros((M, D), device=dev),
                     "xT": rn(M, D).to(tdt), "qkv": rn(M, 3 * D).to(tdt), "att": z16(D), "h": rn(M, F, sc=0.3).to(tdt)})

    tm = {k: [(torch.randn(sz, sz, device=dev).to(tdt), torch.randn(sz, sz, device=dev).to(tdt)) for _ in range(S)] for k, sz in (("torch_mm16_4096", 4096), ("torch_mm16_1024", 1024))}
    tm32 = [(torch.randn(2048, 2048, device=dev), torch.randn(2048, 2048, device=dev)) for _ in range(S)]

    def partner(kind, s):
        j, sp = jobs[s], C.c_void_p(streams[s].cuda_stream)
        ck = _lib.check
        if kind in tm:  # a caller's own 16-bit matrix kernels (rocBLAS / hipBLASLt through torch) on another stream
            with torch.cuda.stream(streams[s]):
                torch.mm(*tm[kind][s])
            return
        if kind == "torch_mm32":
            with torch.cuda.stream(streams[s]):
                torch.mm(*tm32[s])
            return
        if kind in ("ln", "all"):
            ck(lib.s3enc_op_layernorm(DT, ptr(j["x0"]), ptr(j["g1"]), ptr(j["b1"]), M, D, 0, None, ptr(j["xT"]), sp), "ln")
        if kind in ("qkv", "all"):
            ck(lib.s3enc_op_gemm(DT, ptr(j["xT"]), D, M * D, ptr(j["wqkv"]), ptr(j["bqkv"]), M, 3 * D, D, 1, 0, None, None, None, ptr(j["qkv"]), 3 * D, M * 3 * D, sp), "qkv")
        if kind in ("attention", "all"):
            ck(lib.s3enc_op_attention(DT, ptr(j["qkv"]), ptr(j["att"]), ptr(j["valid"]), B, T, H, None, 0, None, sp), "attention")
        if kind in ("fc1", "all"):
            ck(lib.s3enc_op_gemm(DT, ptr(j["xT"]), D, M * D, ptr(j["w1"]), ptr(j["bf1"]), M, F, D, 1, 1, None, None, None, ptr(j["h"]), F, M * F, sp), "fc1")
        if kind in ("fc2", "all"):
            ck(lib.s3enc_op_gemm(DT, ptr(j["h"]), F, M * F, ptr(j["w2"]), ptr(j["bf2"]), M, D, F, 1, 0, ptr(j["xa"]), None, ptr(j["xb"]), None, D, M * D, sp), "fc2")

    for kind in args.partners:
        bad_trials, bad_rows, spans = 0, 0, []
        for _ in range(args.trials):
            for rep in range(args.reps):
                if kind != "none":
                    for s in range(S):
                        for _k in range(12 if kind != "all" else 3):
                            partner(kind, s)
                with torch.cuda.stream(main_s):
                    for _b in range(max(1, args.burst // args.reps)):
                        enc.forward(wavs)
            torch.cuda.synchronize()
            got = tap()
            dm = got != ref
            rb = dm.any(axis=1)
            if rb.any():
                bad_trials += 1
                bad_rows += int(rb.sum())
                spans += [(int(r), int(dm[r].sum()), int(np.nonzero(dm[r])[0][0]), int(np.nonzero(dm[r])[0][-1])) for r in np.nonzero(rb)[0][:3]]
        print(json.dumps({"dtype": args.dtype, "tune": args.tune, "beside conv0": kind, "partner streams": S, "trials": args.trials,
                          "trials whose last conv0 output differs": bad_trials, "bad rows": bad_rows, "examples (row, bad columns, first, last)": spans[:6]}), flush=True)


if __name__ == "__main__":
    main()
