#!/usr/bin/env python3
"""Probe: does running the batch as S independent sub-batches on S HIP streams (one encoder handle each) hide the ramp / tail of
the path's ~100 kernel launches?  The rows of a forward do not depend on the batch they sit in, so the S results
written side by side into one (NS, B, T, D) tensor must equal the one-shot forward bit for bit — checked here.

usage (GPU box): python tools/two_stream_probe.py [--dtype fp32] [--splits 1 2 4] [--steps 30]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--model", default="hubert_base")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--secs", type=float, default=10.0)
    ap.add_argument("--splits", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT")
    ap.add_argument("--private-out", action="store_true", help="every handle writes its own (NS, B / S, T, D) tensor instead of a slice of one")
    ap.add_argument("--prewarm", action="store_true",
                    help="every handle's first forward (its workspace allocation) alone on the default stream, synchronised, before any concurrent run")
    ap.add_argument("--taps-burst", type=int, default=1, help="forwards per handle enqueued back to back in a --taps trial (the streams' phases mix)")
    ap.add_argument("--taps", action="store_true", help="which intermediate of a concurrent run differs first from the same handle's serial run")
    ap.add_argument("--diagnose", action="store_true",
                    help="where a split run differs from the one-shot forward, and whether concurrency or the sub-batch size does it")
    args = ap.parse_args()

    import torch

    from s3prl_amd import _lib
    from s3prl_amd.synth import named_config, synth_weights
    from s3prl_amd.upstream.base import HipUpstreamExpert

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(_lib.load().s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    cfg = named_config(args.model)
    weights = synth_weights(cfg, 0)

    class Expert(HipUpstreamExpert):
        family = cfg.family

    smax = max(args.splits)
    encs = []
    for _ in range(smax):
        ex = Expert.from_weights(cfg, weights, dtype=args.dtype).eval()
        encs.append((ex, ex._encoder_for(dev)))
    enc0 = encs[0][1]
    n = int(args.secs * 16000)
    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1234)
    wavs = [torch.randn(n, device=dev, generator=gen) for _ in range(B)]
    T = enc0.num_output_frames(n)
    NS, D = enc0.num_states(), enc0.embed_dim
    lib = _lib.load()
    streams = [torch.cuda.Stream(device=dev) for _ in range(smax)]

    priv = {}

    def gather_private(S, out):
        per = B // S
        for s in range(S):
            out[:, s * per:(s + 1) * per] = priv[(S, s)]

    def run(S, out, serial=False, one_handle=False):
        """one step: S sub-batches, each on its own stream and handle, into `out` (serial: all on stream 0; one_handle: handle 0)"""
        per = B // S
        done = []
        for s in range(S):
            enc = encs[0 if one_handle else s][1]
            sub = wavs[s * per:(s + 1) * per]
            ptrs = (C.c_void_p * per)(*[w.data_ptr() for w in sub])
            lens = (C.c_int64 * per)(*[n] * per)
            opts = _lib.S3ForwardOpts(_lib.SELECTIONS[None], _lib.F32, 0, 0, None)
            st = streams[0 if serial else s]
            base = out.data_ptr() + s * per * T * D * 4
            stride = B * T * D
            if args.private_out:
                if (S, s) not in priv:
                    priv[(S, s)] = torch.empty((NS, per, T, D), dtype=torch.float32, device=dev)
                base, stride = priv[(S, s)].data_ptr(), per * T * D
            rc = lib.s3enc_forward_ex(enc._h, ptrs, lens, per, n, C.byref(opts), C.c_void_p(base), stride, C.c_void_p(st.cuda_stream))
            _lib.check(rc, "s3enc_forward_ex")
            done.append((ptrs, lens))
        return done

    if args.prewarm:
        for ex, enc in encs:
            enc.forward(wavs)
            torch.cuda.synchronize()

    ref = None
    for S in args.splits:
        assert B % S == 0
        out = torch.empty((NS, B, T, D), dtype=torch.float32, device=dev)
        keep = []
        for _ in range(args.warmup):
            keep.append(run(S, out))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            keep.append(run(S, out))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        if args.private_out:
            gather_private(S, out)
            torch.cuda.synchronize()
        same = None
        if ref is None:
            ref = out.clone()
        else:
            same = bool(torch.equal(ref, out))
        print(json.dumps({"dtype": args.dtype, "model": args.model, "splits": S, "ms_per_step": round(ms, 3),
                          "frames_per_s": round(B * T / ms * 1e3, 1), "bit_identical_to_one_shot": same}), flush=True)
        if args.diagnose and S > 1:
            diagnose(run, ref, S, B, torch, gather_private if args.private_out else None)
        if args.taps and S > 1:
            import numpy as np
            names = ["conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "feat_ln", "proj", "posconv", "qkv0", "attn0"]

            def read_taps():
                got = []
                for s_ in range(S):
                    d = {}
                    for nm in names:
                        buf = np.empty(64 << 20, dtype=np.float32)
                        ne = C.c_int64()
                        rc = lib.s3enc_debug_tap(encs[s_][1]._h, nm.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, C.byref(ne))
                        if rc == 0:
                            d[nm] = buf[:ne.value].copy()
                    got.append(d)
                return got

            for trial in range(3):
                o1 = torch.empty_like(ref)
                k1 = [run(S, o1) for _ in range(args.taps_burst)]
                torch.cuda.synchronize()
                conc = read_taps()
                o2 = torch.empty_like(ref)
                k2 = run(S, o2, serial=True)
                torch.cuda.synchronize()
                ser = read_taps()
                rep = {}
                for s_ in range(S):
                    for nm in names:
                        if nm not in conc[s_]:
                            continue
                        a, b_ = conc[s_][nm], ser[s_][nm]
                        nbad = int((a != b_).sum())
                        if nbad:
                            idx = np.nonzero(a != b_)[0]
                            width = {"qkv0": 3 * D, "proj": D, "posconv": D, "attn0": D}.get(nm, cfg.conv_layers[-1][0] if hasattr(cfg, "conv_layers") else 512)
                            dm = (a != b_).reshape(-1, width)
                            rb, cb = dm.any(axis=1), dm.any(axis=0)
                            rel = np.abs(a - b_) / (np.abs(b_) + 1e-6)
                            if nm == "conv0":  # what IS the wrong value?  (good, bad, the same channel's good value 2 / 1 frames before and after)
                                W_ = width
                                A2, B2 = a.reshape(-1, W_), b_.reshape(-1, W_)
                                ex = []
                                for r_ in np.nonzero(rb)[0][:6]:
                                    for c_ in np.nonzero(dm[r_])[0][:3]:
                                        nb = {d_: float(B2[r_ + d_, c_]) for d_ in (-4, -2, -1, 1, 2, 4) if 0 <= r_ + d_ < B2.shape[0]}
                                        ex.append({"row": int(r_), "col": int(c_), "good": float(B2[r_, c_]), "bad": float(A2[r_, c_]), "neighbours (good)": nb,
                                                   "cols c+1..c+3 good": [float(B2[r_, c_ + k]) for k in (1, 2, 3) if c_ + k < W_]})
                                print(json.dumps({"conv0 examples": ex}), flush=True)
                            rep.setdefault(nm, []).append({"handle": s_, "bad": nbad, "n": int(a.size), "max abs": float(np.abs(a - b_).max()),
                                                           "median rel of bad": float(np.median(rel[a != b_])),
                                                           "rows bad": int(rb.sum()), "rows": int(rb.size), "cols bad": int(cb.sum()), "cols": int(cb.size),
                                                           "bad per bad row (median)": float(np.median(dm[rb].sum(axis=1))),
                                                           "first bad rows": [int(x) for x in np.nonzero(rb)[0][:12]],
                                                           # (row, bad columns, first, last): an ORIGIN row of a GEMM is bad inside one column tile
                                                           "column span of bad rows": [(int(r), int(dm[r].sum()), int(np.nonzero(dm[r])[0][0]), int(np.nonzero(dm[r])[0][-1]))
                                                                                       for r in np.nonzero(rb)[0][:16]]})
                print(json.dumps({"splits": S, "trial": trial, "taps that differ (handle, n_bad, n, first, last, max abs)": rep,
                                  "output differs": not bool(torch.equal(o1, o2))}), flush=True)


def diagnose(run, ref, S, B, torch, gather=None):
    per = B // S
    for label, kw in (("concurrent streams, one handle each", {}), ("same, second run", {}),
                      ("ONE stream, one handle each", {"serial": True}), ("ONE stream, ONE handle", {"serial": True, "one_handle": True})):
        out = torch.empty_like(ref)
        keep = run(S, out, **kw)
        torch.cuda.synchronize()
        if gather is not None:
            gather(S, out)
            torch.cuda.synchronize()
        bad = []
        for l in range(ref.shape[0]):
            for b in range(B):
                if not torch.equal(ref[l, b], out[l, b]):
                    d = (ref[l, b] - out[l, b]).abs()
                    rows = (d.amax(dim=1) > 0).nonzero().flatten()
                    bad.append((l, b, float(d.max()), int(rows.numel()), int(rows[0]), int(rows[-1])))
        nan_pairs = int(torch.isnan(out).flatten(2).any(dim=2).sum().item())  # (S3ENC_DEBUG_POISON=1: a read that overtook its producer)
        print(json.dumps({"splits": S, "how": label, "differing (layer, utterance) pairs": len(bad), "pairs with NaN": nan_pairs,
                          "first": bad[:6], "layers": sorted({x[0] for x in bad})[:14], "utterances": sorted({x[1] for x in bad})}), flush=True)


if __name__ == "__main__":
    main()
