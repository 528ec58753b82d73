#!/usr/bin/env python3
"""Headline benchmark: encoder-frames/sec (20 ms stride), HuBERT-base, 32 x 10 s @16 kHz per GPU.

    python bench.py --gpus 1 --steps K --warmup W [--dtype fp32|bf16|fp16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one batch: raw waveforms already resident in HBM ->
all NL+1 hidden_states (fp32, (B,T,D) each) in HBM; with N > 1 every rank encodes its own 32 utterances
(weak scaling) and the batch's hidden states are re-assembled on every rank by per-layer RCCL all-gathers
(inside the timed region).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     — the dominant kernel (the MFMA GEMM that serves conv1-6 and every linear layer): algorithmic
                 FLOPs / summed HIP-event time of its launches inside the timed region vs the dense MFMA peak
                 of the compute dtype (MI355X_MICROARCH.md: 157.3 TF fp32-in, 2500 TF bf16/f16).
  cpu_baseline — oracle/torch_oracle.py (the reference forward restated on the reference's own ATen call sites:
                 F.conv1d / F.group_norm / F.multi_head_attention_forward ..., PyTorch CPU fp32, all host threads —
                 /root/reference itself does not exist on the GPU box) timed on this box's host cores on a bounded
                 sample of the same workload (N=1, rank 0 only).
  parity       — max per-layer relative error of the HIP path on a sample, against the independent numpy oracle
                 (oracle/encoder_oracle.py) and against the torch restatement.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# fp32x3: three bf16 MFMAs per product -> the matrix-pipe ceiling for ALGORITHMIC flops is 2500 / 3
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0, "fp32x3": 2500.0 / 3}
MODEL_NAMES = {"hubert_base": "HuBERT-base", "hubert_large": "HuBERT-large", "wav2vec2_base": "wav2vec2-base",
               "wav2vec2_large": "wav2vec2-large", "wavlm_base_plus": "WavLM-base+", "wavlm_large": "WavLM-large"}


def pmc_traffic(path, model, dtype, batch, secs):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC passes of this same workload
    (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md §HBM); None when no matching record is committed."""
    try:
        for rec in json.load(open(path)):
            if (rec["model"], rec["dtype"], rec["batch"], rec["secs"]) == (model, dtype, batch, secs):
                return rec
    except (OSError, ValueError, KeyError):
        pass
    return None


def flops_per_utt(cfg, n):
    """Algorithmic FLOPs of one n-sample utterance (SURVEY §8d formula)."""
    L = cfg.conv_lengths(n)
    C, D, F, NL = cfg.conv_dim, cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.encoder_layers
    cin, f = 1, 0.0
    for (_, k, _), l in zip(cfg.conv_layers, L):
        f += 2.0 * cin * C * k * l
        cin = C
    T = L[-1]
    Tp = T + (T % 2) if cfg.family != "wavlm" else T
    f += 2.0 * T * C * D + 2.0 * T * D * (D // cfg.conv_pos_groups) * cfg.conv_pos
    f += NL * (2.0 * Tp * (4 * D * D + 2 * D * F) + 4.0 * Tp * Tp * D)
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default=os.environ.get("S3ENC_BENCH_DTYPE", "fp32"), choices=["fp32", "bf16", "fp16", "fp32x3"])
    ap.add_argument("--model", default="hubert_base")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--secs", type=float, default=10.0)
    ap.add_argument("--mixed", action="store_true",
                    help="mixed-length batch (BASELINE configs[4] recipe): utterance 0 has --secs, the rest "
                         "randint(1 s, --secs), seed 1234; frames are counted per utterance (sum of T_i)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="s3enc_set_tuning knob for A/B runs (e.g. gemm16_big=4); results are unchanged")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the N-rank path on a "
                         "box with fewer GPUs than ranks: ranks then share devices)")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip the short side measurement of the other operand modes (fp32x3, bf16) in the default run")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region (A/B of their cost)")
    ap.add_argument("--cpu-sample", type=int, default=32, help="utterances of the workload timed on the CPU oracle")
    ap.add_argument("--parity-sample", type=int, default=2, help="utterances checked against the numpy oracle")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-kernel HBM-side bytes from the committed rocprofv3 PMC passes (tools/pmc.sh)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())

    if args.tune:
        from s3prl_amd import _lib

        for kv in args.tune:
            k, v = kv.split("=")
            _lib.check(_lib.load().s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    cfg = named_config(args.model)
    weights = synth_weights(cfg, 0)  # random-init weights of the named architecture (no checkpoints offline)
    enc = HipEncoder(cfg, weights, dtype=args.dtype, device=dev.index)
    n = int(args.secs * 16000)
    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    lens = [n] * B
    if args.mixed:
        rng = np.random.default_rng(1234 + rank)
        lens = [n] + [int(x) for x in rng.integers(16000, n, size=B - 1)]
    wavs = [torch.randn(m, device=dev, generator=gen) for m in lens]
    T = enc.num_frames(n)
    frames_per_batch = sum(enc.num_frames(m) for m in lens)
    NL, D = cfg.encoder_layers, cfg.encoder_embed_dim
    out = torch.empty((NL + 1, B, T, D), dtype=torch.float32, device=dev)
    gathered = torch.empty((NL + 1, world * B, T, D), dtype=torch.float32, device=dev) if world > 1 else None

    events = None
    if world > 1:
        from s3prl_amd.parallel import gather_layers

        events = enc.layer_events()

    def step():
        enc.forward(wavs, out=out)
        if world > 1:
            # one all-gather per layer (hidden_states[l] stays a contiguous (B_global, T, D) block), issued on a
            # side stream as soon as layer l is final so it overlaps the remaining layers' compute
            gather_layers(out, overlap_events=events, out=gathered)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    enc.profile_reset()
    # timed region: HIP events only around the launches of the dominant kernel (the GEMM), and only on every 4th step —
    # an event pair costs a ~5 us bubble on the stream: around all ~290 launches of a forward that is 0.8 ms per batch
    # (2 % fp32, 10 % bf16), around the 55 GEMM launches 0.6 ms.  The full per-kernel breakdown is measured in extra,
    # untimed steps after the timed region.
    PROF_EVERY = 4
    prof_steps = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = (not args.no_profile) and i % PROF_EVERY == 0
        enc.profile_enable(2 if on else 0)
        prof_steps += int(on)
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    enc.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = enc.profile_read()
    breakdown, bd_steps = [], 3
    if not args.no_profile:
        enc.profile_reset()
        enc.profile_enable(1)
        for _ in range(bd_steps):
            enc.forward(wavs, out=out)
        torch.cuda.synchronize()
        enc.profile_enable(0)
        breakdown = enc.profile_read()

    if rank == 0:
        frames = world * frames_per_batch * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        value = frames / elapsed
        gem = [p for p in prof if p["name"].startswith("gemm:")]
        g_ms = sum(p["ms"] for p in gem)
        g_fl = sum(p["flops"] for p in gem)
        g_n = sum(p["launches"] for p in gem)
        g_by = sum(p["bytes"] for p in gem)
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        g_ms = g_ms or 1e-30
        peak = PEAK_TFLOPS[args.dtype]
        total_ms = sum(p["ms"] for p in breakdown) / bd_steps * args.steps if breakdown else 0.0
        line = {
            "metric": f"encoder-frames/sec (20 ms stride) {MODEL_NAMES.get(args.model, args.model)} {B}x{args.secs:g} s @16 kHz",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "fp32x3": "bf16x3 (split fp32, fp32 accumulate)"}[args.dtype],
            "data": "synthetic",
            "config": {
                "workload": f"{args.model} random-init, {B}x{args.secs:g} s @16 kHz per GPU, all {NL + 1} hidden_states "
                            f"(fp32) written; per-layer RCCL all-gather across {world} GPU(s)",
                "utterances_per_gpu": B, "samples": n, "frames_per_utt": T, "parallelism": f"dp{world}",
                "lengths": "mixed (utt 0 = max, rest randint(16000, max), padding-masked)" if args.mixed else "equal",
            },
            # the reference computes padded frames too, so the path's work is B x F_utt(n_max)
            "path_tflops": round(world * B * flops_per_utt(cfg, n) / (ms_per_step * 1e-3) / 1e12, 2),
            "roofline": {
                "kernel": ({"fp32": "gemm_kernel<float> (gemm.hip)", "fp32x3": "gemm_x3_kernel (gemm_x3.hip)"}.get(args.dtype, "gemm16_big_kernel (gemm16.hip)"))
                          + ": conv1-6 implicit GEMM + proj/qkv/out_proj/fc1/fc2",
                "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None, "algorithmic_bytes": round(g_by / max(g_n, 1)),
                "launches_per_step": g_n // max(prof_steps, 1), "avg_launch_ms": round(g_ms / max(g_n, 1), 4),
                "timed_steps_with_events": prof_steps,
                "share_of_kernel_time": round(g_ms / total_ms, 3) if total_ms else None,
            },
            # every kernel kind, from the untimed all-kernel profiling steps after the timed region
            "kernels_ms_per_step": {p["name"]: round(p["ms"] / bd_steps, 4) for p in sorted(breakdown, key=lambda p: -p["ms"])},
        }
        tr = pmc_traffic(args.traffic, args.model, args.dtype, B, args.secs)
        if tr is not None:
            line["roofline"]["traffic"] = tr["gemm_bytes_per_launch"]
            line["roofline"]["traffic_source"] = tr["source"]
        if world == 1 and not args.no_cpu_baseline:
            # checkers + CPU baseline only; never on the product path
            from oracle import encoder_oracle as O
            from oracle import torch_oracle as TO

            ns = max(1, min(args.cpu_sample, B))
            if args.mixed:
                raise SystemExit("--mixed: pass --no-cpu-baseline (the CPU sample assumes equal lengths)")
            Wt = TO.prepare(cfg, weights)
            sample = [w.cpu() for w in wavs[:ns]]
            # the reference's recipe is set_num_threads(os.cpu_count()); on a many-core host that oversubscribes oneDNN
            # (256 threads: 32 s per utterance on the GPU box), so give the CPU path its best thread count: sweep on
            # a 4-utterance batch, keep the fastest
            ncpu = os.cpu_count() or 1
            sweep = {}
            for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128)}):
                torch.set_num_threads(th)
                TO.forward(cfg, Wt, sample[:1])  # warm-up (thread pool, oneDNN primitive cache)
                t1 = time.perf_counter()
                TO.forward(cfg, Wt, sample[:4])
                sweep[th] = time.perf_counter() - t1
            best = min(sweep, key=sweep.get)
            torch.set_num_threads(best)
            t1 = time.perf_counter()
            ref_t = TO.forward(cfg, Wt, sample)
            cpu_s = time.perf_counter() - t1
            hs = enc.forward(wavs[:ns])
            torch.cuda.synchronize()
            err_t = max(O.rel_err(hs[l].cpu().numpy(), ref_t[l].numpy()) for l in range(NL + 1))
            npar = max(1, min(args.parity_sample, ns))
            ref_n = O.forward(cfg, weights, [w.numpy() for w in sample[:npar]], dtype=np.float32)
            hs_n = enc.forward(wavs[:npar])
            torch.cuda.synchronize()
            err_n = max(O.rel_err(hs_n[l].cpu().numpy(), ref_n[l]) for l in range(NL + 1))
            line["cpu_baseline"] = {
                "value": round(ns * T / cpu_s, 1), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{ns}x{args.secs:g} s of the same workload through oracle/torch_oracle.py (the reference's ATen "
                          f"call sites, PyTorch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads on "
                          f"{os.cpu_count()} host cores — the fastest of a 4-utterance sweep "
                          f"{ {k: round(v, 2) for k, v in sweep.items()} } s), {cpu_s:.1f} s wall",
            }
            line["parity"] = {"max_layer_rel_err_vs_numpy_oracle": float(f"{err_n:.3e}"), "numpy_sample": f"{npar} utterances",
                              "max_layer_rel_err_vs_torch_oracle": float(f"{err_t:.3e}"), "torch_sample": f"{ns} utterances",
                              "tolerance": 1e-3}
            if not args.no_other_modes and not args.mixed:
                # side measurement (same workload, same inputs, untimed for the headline): the opt-in operand modes,
                # each with its own parity against the torch restatement of the full batch
                other = {}
                for mode in [m for m in ("fp32x3", "bf16") if m != args.dtype]:
                    enc2 = HipEncoder(cfg, weights, dtype=mode, device=dev.index)
                    out2 = torch.empty_like(out)
                    for _ in range(3):
                        enc2.forward(wavs, out=out2)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(10):
                        enc2.forward(wavs, out=out2)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t1) / 10
                    err = max(O.rel_err(out2[l][:ns].cpu().numpy(), ref_t[l].numpy()) for l in range(NL + 1))
                    other[mode] = {"value": round(B * T / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                                   "max_layer_rel_err_vs_torch_oracle": float(f"{err:.3e}")}
                    enc2.close()
                    del out2
                line["other_modes"] = other
        print(json.dumps(line))
    enc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
