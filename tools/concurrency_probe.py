#!/usr/bin/env python3
"""Does a forward keep its bits when OTHER work runs on the GPU beside it (another HIP stream)?  One encoder handle, one batch; the
quiet forward is the reference, then the same forward is repeated while a side stream runs (a) idle workgroups holding CU slots
(s3enc_debug_occupy_cus), (b) torch elementwise kernels over a large tensor, (c) torch matmuls (rocBLAS / hipBLASLt kernels with LDS),
(d) a second handle's forwards of the same / another dtype.  Every repeat is compared with the quiet result bit for bit.

usage (GPU box): python tools/concurrency_probe.py --dtype bf16 [--repeats 6] [--tune KEY=INT ...]"""
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
    ap.add_argument("--model", default="hubert_base")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--secs", type=float, default=10.0)
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--side", nargs="+", default=["none", "occupy", "eltwise", "matmul", "same", "fp32"])
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT")
    args = ap.parse_args()

    import torch

    from s3prl_amd import _lib
    from s3prl_amd.synth import named_config, synth_weights
    from s3prl_amd.upstream.base import HipUpstreamExpert

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.load()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(lib.s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    cfg = named_config(args.model)
    weights = synth_weights(cfg, 0)

    class Expert(HipUpstreamExpert):
        family = cfg.family

    def make(dtype):
        ex = Expert.from_weights(cfg, weights, dtype=dtype).eval()
        return ex, ex._encoder_for(dev)

    main_ex, enc = make(args.dtype)
    n = int(args.secs * 16000)
    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1234)
    wavs = [torch.randn(n, device=dev, generator=gen) for _ in range(B)]
    side_stream = torch.cuda.Stream(device=dev)
    main_stream = torch.cuda.Stream(device=dev)

    with torch.cuda.stream(main_stream):
        ref = enc.forward(wavs).clone()
    torch.cuda.synchronize()
    others = {}
    big = torch.randn(64 << 20, device=dev)
    ma, mb = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)

    def side_work(kind):
        """enqueue ~one forward's worth of foreign work on the side stream"""
        if kind == "none":
            return
        with torch.cuda.stream(side_stream):
            if kind == "occupy":
                _lib.check(lib.s3enc_debug_occupy_cus(64, 256, 8.0, C.c_void_p(side_stream.cuda_stream)), "occupy")
            elif kind == "eltwise":
                for _ in range(40):
                    big.mul_(1.0000001)
            elif kind == "matmul":
                for _ in range(30):
                    torch.mm(ma, mb)
            else:
                dt = args.dtype if kind == "same" else kind
                if dt not in others:
                    others[dt] = make(dt)
                others[dt][1].forward(wavs[:B // 2])

    for kind in args.side:
        side_work(kind)  # (creates the second handle / warms the side kernels outside the compared repeats)
        torch.cuda.synchronize()
        bad, worst = 0, 0.0
        for _ in range(args.repeats):
            side_work(kind)
            with torch.cuda.stream(main_stream):
                out = enc.forward(wavs)
            side_work(kind)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                worst = max(worst, float((out - ref).abs().max()))
        print(json.dumps({"dtype": args.dtype, "tune": args.tune, "beside": kind, "repeats": args.repeats, "repeats that differ from the quiet forward": bad,
                          "max abs diff": worst}), flush=True)


if __name__ == "__main__":
    main()
