#!/usr/bin/env python3
"""Probe: does a forward keep its bits when the workgroups of its kernels get a NON-ZERO LDS base address?

A side stream first fills the chip with idle workgroups that each hold `--lds` bytes of LDS (tools/micro/liblds_occupy.so); the
forward launched behind them on another stream shares its CUs — and their LDS — with them.  Alone on the chip every workgroup of the
path starts at LDS address 0 (one workgroup per CU for the 112 / 128 KiB tiles of the 16-bit GEMMs); beside other handles' forwards
it does not.  (Written while looking for the cause of profiles/r06c_concurrent_forwards.md.)

usage (GPU box): python tools/lds_base_probe.py [--dtype bf16] [--lds 0 4096 16384 28672] [--workgroups 256 1024]"""
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--secs", type=float, default=10.0)
    ap.add_argument("--lds", type=int, nargs="+", default=[0, 4096, 16384, 28672, 45056])
    ap.add_argument("--workgroups", type=int, nargs="+", default=[256, 1024])
    ap.add_argument("--threads", type=int, default=64)
    ap.add_argument("--ms", type=float, default=40.0)
    ap.add_argument("--mode", type=int, default=0, help="what the holders do: 0 sleep, 1 stream LDS reads / writes, 2 stream global loads")
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT")
    args = ap.parse_args()

    import torch

    from s3prl_amd import _lib
    from s3prl_amd.synth import named_config, synth_weights
    from s3prl_amd.upstream.base import HipUpstreamExpert

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.load()
    occ = C.CDLL(os.path.join(ROOT, "tools", "micro", "liblds_occupy.so"))
    occ.lds_occupy2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    gbuf = torch.zeros(64 << 20, dtype=torch.float32, device=dev)
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(lib.s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    cfg = named_config(args.model)
    weights = synth_weights(cfg, 0)

    class Expert(HipUpstreamExpert):
        family = cfg.family

    ex = Expert.from_weights(cfg, weights, dtype=args.dtype).eval()
    enc = ex._encoder_for(dev)
    n = int(args.secs * 16000)
    gen = torch.Generator(device=dev).manual_seed(1234)
    wavs = [torch.randn(n, device=dev, generator=gen) for _ in range(args.batch)]
    side, main_s = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    with torch.cuda.stream(main_s):
        ref = enc.forward(wavs).clone()
        again = enc.forward(wavs)
    torch.cuda.synchronize()
    assert torch.equal(ref, again), "the quiet forward is not reproducible"
    for wg in args.workgroups:
        for lds in args.lds:
            bad, worst, pairs = 0, 0.0, 0
            for _ in range(args.repeats):
                if lds > 0:
                    rc = occ.lds_occupy2(wg, args.threads, lds, args.ms, C.c_void_p(side.cuda_stream), args.mode, C.c_void_p(gbuf.data_ptr()), gbuf.numel() * 4)
                    assert rc == 0, rc
                with torch.cuda.stream(main_s):
                    out = enc.forward(wavs)
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad += 1
                    worst = max(worst, float((out - ref).abs().max()))
                    pairs += sum(1 for l in range(ref.shape[0]) for b in range(ref.shape[1]) if not torch.equal(out[l, b], ref[l, b]))
            print(json.dumps({"dtype": args.dtype, "model": args.model, "batch": args.batch, "holders": wg, "holder_mode": args.mode, "threads": args.threads, "lds_bytes_each": lds,
                              "repeats": args.repeats, "repeats that differ from the quiet forward": bad, "differing (state, utterance) pairs": pairs,
                              "max abs diff": worst}), flush=True)


if __name__ == "__main__":
    main()
