#!/usr/bin/env python3
"""Headline benchmark: encoder-frames/sec (20 ms stride), HuBERT-base, 32 x 10 s @16 kHz per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype fp32|fp32x3|fp16x2|bf16|fp16] [--model ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

``--gpus N`` without a launcher (no WORLD_SIZE in the environment) re-executes itself under ``torch.distributed.run``
with N ranks, one per GPU; ``n_gpus`` in the JSON line is the world size RCCL really saw.

A "step" is one full ``UpstreamExpert.forward``: raw waveforms already resident in HBM -> all NL+1
hidden_states in HBM.  With N > 1 every rank encodes its own shard and the batch's hidden states are re-assembled on
every rank (inside the timed region):
  --gather auto        (default) layers for fp32 / fp32x3, layers16 for the 16-bit compute dtypes
  --gather layers      one RCCL all-gather per layer, issued on a side stream as each layer becomes final
  --gather layers16    the same with 16-bit states (bf16 / fp16 compute modes): half the bytes
  --exchange-algo ring|direct|copy   one all-gather per state, its all-pairs send / receive form (one peer per xGMI link), or the
                                copy-engine form (S3ENC_EXCHANGE_COPY: IPC-mapped slabs + hipMemcpyAsync, no CU, no RCCL; per-state gathers only)
  --gather featurized  the Featurizer's weighted sum runs as the encoder's epilogue; ONE (B, T, D) all-gather
  --gather none        no exchange (what "exposed communication" is measured against)
--scaling weak (default): --batch utterances per GPU;  --scaling strong: --global-batch utterances split over the ranks.
Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     — the dominant kernel (the MFMA GEMM that serves conv1-6 and every linear layer): algorithmic FLOPs /
                 summed HIP-event time of its launches inside the timed region vs the dense MFMA peak of the compute
                 dtype (MI355X_MICROARCH.md: 157.3 TF fp32-in, 2500 TF bf16/f16).
  cpu_baseline — oracle/torch_oracle.py (the reference forward restated on the reference's own ATen call sites, PyTorch
                 CPU fp32 — /root/reference itself does not exist on the GPU box) timed on this box's host cores on a
                 bounded sample of the same workload (N = 1, rank 0 only).
  parity       — max per-layer relative error of the HIP path on a sample of THIS workload's shapes (for mixed-length
                 batches the longest + shortest utterance padded to the batch n_max), against the torch restatement and
                 the independent numpy oracle.  Every timed configuration carries one.
  comm         — (N > 1) bytes each GPU receives per step, the step time without the exchange and the exposed part.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# fp32x3: three bf16 MFMAs per product -> the matrix-pipe ceiling for ALGORITHMIC flops is 2500 / 3
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0, "fp32x3": 2500.0 / 3, "fp16x2": 2500.0 / 2}
MODEL_NAMES = {"hubert_base": "HuBERT-base", "hubert_large": "HuBERT-large", "wav2vec2_base": "wav2vec2-base",
               "wav2vec2_large": "wav2vec2-large", "wavlm_base_plus": "WavLM-base+", "wavlm_large": "WavLM-large",
               "wavlm_base": "WavLM-base", "distilhubert": "DistilHuBERT"}
DTYPE_NAMES = {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "fp32x3": "bf16x3 (split fp32, fp32 accumulate)",
               "fp16x2": "f16x2 (fp16 activations x two-term weights: fp16 + fp16, or fp16 + MX-fp4 on the scaled-MFMA pipe for q|k|v / fc1 / fc2; fp32 accumulate)"}
# default timed region >= 5 s of GPU work at the default workload (HuBERT-base 32 x 10 s): steps per dtype
DEFAULT_STEPS = {"fp32": 150, "fp32x3": 330, "bf16": 700, "fp16": 700, "fp16x2": 500}


def csrc_sha16():
    """Identity of the kernels a PMC record was measured on: sha256 over s3prl_amd/csrc/*.hip, *.h (sorted by name), 16 hex digits.
    tools/pmc_to_traffic.py stamps every record of profiles/traffic.json with it (the GPU box has no .git)."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "s3prl_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def device_code_md5():
    """md5 of the `.hip_fatbin` section of the built s3prl_amd/libs3enc.so — the gfx950 code objects themselves.  The second identity a
    PMC record carries (round 5): a change to HOST code under csrc/ (a weight packer, an argument check) moves `csrc_sha16` but not the
    kernels the counters were measured on.  None if the library is not built / not an ELF64 file."""
    import hashlib
    import struct

    try:
        with open(os.path.join(ROOT, "s3prl_amd", "libs3enc.so"), "rb") as f:
            data = f.read()
        if data[:4] != b"\x7fELF" or data[4] != 2:
            return None
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
        sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
        str_off = sec(shstrndx)[4]
        for i in range(shnum):
            name, _type, _flags, _addr, off, size = sec(i)[:6]
            end = data.index(b"\0", str_off + name)
            if data[str_off + name:end] == b".hip_fatbin":
                return hashlib.md5(data[off:off + size]).hexdigest()
    except (OSError, ValueError, struct.error):
        pass
    return None


def pmc_traffic(path, model, dtype, batch, secs):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC passes of this same workload
    (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md §HBM).  Returns (record or None, note or None): a record measured on other
    kernels than the ones in this tree (its `csrc_sha16` stamp differs and so does its `device_code_md5`, or it carries neither) is NOT
    quoted — the note says so."""
    try:
        for rec in json.load(open(path)):
            if (rec["model"], rec["dtype"], rec["batch"], rec["secs"]) == (model, dtype, batch, secs):
                have = csrc_sha16()
                if rec.get("csrc_sha16") == have:
                    return rec, None
                if rec.get("device_code_md5") and rec["device_code_md5"] == device_code_md5():
                    return rec, None  # the sources moved (host code), the code objects the counters were measured on did not
                return None, (f"profiles/traffic.json holds a record of this workload measured on kernels {rec.get('csrc_sha16', '(unstamped, round <= 3)')}"
                              f"{' at commit ' + rec['commit'] if rec.get('commit') else ''}; this tree's csrc is {have} — stale, not quoted "
                              "(tools/round_profiles.sh re-measures it)")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def flops_per_utt(cfg, n):
    """Algorithmic FLOPs of one n-sample utterance (SURVEY §8d formula)."""
    L = cfg.conv_lengths(n)
    C, D, F, NL = cfg.conv_dim, cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.encoder_layers
    cin, f = 1, 0.0
    for (_, k, _), l in zip(cfg.conv_layers, L):
        f += 2.0 * cin * C * k * l
        cin = C
    T = L[-1]
    Tp = T + (T % 2) if cfg.family in ("hubert", "wav2vec2") else T
    f += 2.0 * T * C * D + 2.0 * T * D * (D // cfg.conv_pos_groups) * cfg.conv_pos
    if cfg.family == "multires_hubert":
        # the same per-layer formula at every block's own frame count, plus the adapter convolutions as the reference
        # runs them (ConvTranspose1d: 2*D*D*k per INPUT frame, Conv1d: 2*D*D*k per OUTPUT frame; hubert_model.py:970-1266)
        k = cfg.conv_adapter_kernel
        for blk in cfg.multires_plan(T)[0]:
            tb = blk["T"] + (blk["T"] % 2)
            f += blk["layers"] * (2.0 * tb * (4 * D * D + 2 * D * F) + 4.0 * tb * tb * D)
        prev = None
        for blk in cfg.multires_plan(T)[0]:
            if blk["adapter"] is not None:
                kind, up, down, _ = blk["adapter"]
                t_in = prev["T"] if "T_sum" not in prev else prev["T_sum"]
                rows = t_in
                if kind in ("full", "up"):
                    f += 2.0 * rows * D * D * k
                    rows *= up
                if kind in ("full", "down"):
                    f += 2.0 * ((rows - 1) // down + 1) * D * D * k
            prev = blk
        return f
    f += NL * (2.0 * Tp * (4 * D * D + 2 * D * F) + 4.0 * Tp * Tp * D)
    return f


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: >= 5 s of GPU work at the default workload)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default=os.environ.get("S3ENC_BENCH_DTYPE", "fp32"), choices=["fp32", "bf16", "fp16", "fp32x3", "fp16x2"])
    ap.add_argument("--model", default="hubert_base")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--global-batch", type=int, default=256, help="utterances of the whole job (strong scaling, SURVEY §8d cfg4)")
    ap.add_argument("--secs", type=float, default=10.0)
    ap.add_argument("--mixed", action="store_true",
                    help="mixed-length batch (BASELINE configs[4] recipe): utterance 0 has --secs, the rest "
                         "randint(1 s, --secs), seed 1234; frames are counted per utterance (sum of T_i)")
    ap.add_argument("--gather", default="auto", choices=["auto", "layers", "layers16", "featurized", "none"],
                    help="auto (default): layers for fp32 / fp32x3, layers16 for the 16-bit compute dtypes (half the xGMI bytes: a "
                         "16-bit step is too short to hide fp32 slabs behind, DESIGN §7)")
    ap.add_argument("--exchange-algo", default="ring", choices=["ring", "direct", "copy"],
                    help="ring: one all-gather per state (RCCL picks its algorithm); direct: per state one group of all-pairs "
                         "send / receive — xGMI is point-to-point, every peer has its own link (S3ENC_EXCHANGE_DIRECT)")
    ap.add_argument("--exchange-via", default="torch", choices=["torch", "cabi"],
                    help="who issues the per-layer RCCL all-gathers: torch.distributed (default) or the library's own "
                         "s3enc_comm_* entry points (the path a non-Python binder uses; needs --backend nccl)")
    ap.add_argument("--steal-cus", type=int, default=0, metavar="K",
                    help="CU-contention proxy (one-GPU boxes cannot time a real exchange against the compute it overlaps): K idle workgroups "
                         "of 256 threads hold CU slots on a side stream for the whole timed region, like the channel kernels of a "
                         "collective would; combine with --tune reserve_cus=K (profiles/r05_cu_contention.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline leg (the parity leg still runs)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="s3enc_set_tuning knob for A/B runs (e.g. gemm16_big=4); results are unchanged")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the N-rank path on a "
                         "box with fewer GPUs than ranks: ranks then share devices)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group / exchange plumbing only, on CPU tensors (no GPU, no encoder): what the "
                         "CPU-box test of `--gpus N` runs")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip the short side measurement of the other operand modes (fp32x3, bf16) in the default run")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region (A/B of their cost)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="utterances of the workload timed on the CPU oracle")
    ap.add_argument("--parity-sample", type=int, default=2, help="utterances checked against the numpy oracle")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-kernel HBM-side bytes from the committed rocprofv3 PMC passes (tools/pmc.sh)")
    return ap.parse_args(argv)


def self_launch(args):
    """`bench.py --gpus N` without a launcher: run N ranks of this same command under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """No GPU: the rendezvous, the shard bookkeeping and the per-layer exchange of the N-rank path on CPU tensors."""
    import torch
    import torch.distributed as dist

    from s3prl_amd.parallel import gather_layers, shard_bounds

    if world > 1:
        dist.init_process_group(args.backend if args.backend != "nccl" else "gloo")
    B = args.batch if args.scaling == "weak" else -(-args.global_batch // world)
    hs = torch.full((3, B, 5, 8), float(rank))
    # (the copy-engine form needs device memory: the dry run checks the same layout contract through the collective)
    got = gather_layers(hs, algo=args.exchange_algo if args.exchange_algo != "copy" else "ring") if world > 1 else hs
    ok = all(bool((got[:, r * B:(r + 1) * B] == r).all()) for r in range(world))
    if world > 1:
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item() == 1.0)
    if rank == 0:
        print(json.dumps({"metric": "dry run of the N-rank launcher / exchange plumbing (no GPU work)", "value": None,
                          "unit": "frames/s", "n_gpus": dist.get_world_size() if world > 1 else 1, "dry_run": True,
                          "exchange_ok": ok, "scaling": args.scaling, "utterances_per_rank": B,
                          "shard_of_rank0": list(shard_bounds(B * world, world, 0)), "backend": args.backend}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    args = parse_args()
    # multi-process GPU work needs dmabuf IPC (the host driver has no legacy IPC): must be in the environment before the
    # HIP runtime initialises, i.e. before torch touches the device
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))

    import numpy as np
    import torch

    from s3prl_amd.synth import named_config, synth_weights
    from s3prl_amd.upstream.base import HipUpstreamExpert

    assert torch.cuda.is_available(), "bench.py needs an MI355X (use --dry-run for the CPU-only launcher check)"
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): pass the same N to both")
    ndev = torch.cuda.device_count()
    dist = None
    rccl = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl" and world > ndev:
            raise SystemExit(f"--gpus {world} over RCCL needs {world} visible devices, this box has {ndev} "
                             f"(use --backend gloo to let ranks share a device for a functional check)")
        local_rank %= max(1, ndev)
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = "unknown"
        else:
            dist.init_process_group(args.backend)
        world = dist.get_world_size()  # what the process group really has
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from s3prl_amd import _lib

    if args.tune:
        for kv in args.tune:
            k, v = kv.split("=")
            _lib.check(_lib.load().s3enc_set_tuning(k.encode(), int(v)), "s3enc_set_tuning")
    cfg = named_config(args.model)
    weights = synth_weights(cfg, 0)  # random-init weights of the named architecture (no checkpoints offline)

    class Expert(HipUpstreamExpert):
        family = cfg.family

    expert = Expert.from_weights(cfg, weights, dtype=args.dtype).eval()
    enc = expert._encoder_for(dev)
    n = int(args.secs * 16000)
    B = args.batch if args.scaling == "weak" else -(-args.global_batch // world)
    steps = args.steps or DEFAULT_STEPS[args.dtype]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    lens = [n] * B
    if args.mixed:
        rng = np.random.default_rng(1234 + rank)
        lens = [n] + [int(x) for x in rng.integers(16000, n, size=B - 1)]
    wavs = [torch.randn(m, device=dev, generator=gen) for m in lens]
    T = enc.num_output_frames(n)  # = num_frames(n) except for multires-HuBERT (states cut to their common length)
    frames_per_batch = sum(enc.num_frames(m) for m in lens)
    NL, D = cfg.encoder_layers, cfg.encoder_embed_dim
    NS = enc.num_states()
    gather = args.gather
    if gather == "auto":
        gather = "layers16" if args.dtype in ("bf16", "fp16", "fp16x2") else "layers"
    gather = gather if world > 1 else "none"
    if gather == "layers16" and args.dtype not in ("bf16", "fp16", "fp16x2"):
        raise SystemExit("--gather layers16 needs a 16-bit compute dtype")
    feat_w = torch.softmax(torch.linspace(-1.0, 1.0, NS), 0).tolist()  # a Featurizer's softmax(weights)
    events = enc.layer_events() if gather in ("layers", "layers16") else None
    gathered = None
    if gather in ("layers", "layers16"):
        gdt = torch.float32 if gather == "layers" else (torch.bfloat16 if args.dtype == "bf16" else torch.float16)
        gathered = torch.empty((NS, world * B, T, D), dtype=gdt, device=dev)
    elif gather == "featurized":
        gathered = torch.empty((world * B, T, D), dtype=torch.float32, device=dev)

    cabi = None
    copyc = None
    if world > 1 and args.exchange_algo == "copy":
        if gather not in ("layers", "layers16"):
            raise SystemExit("--exchange-algo copy moves the per-state slabs (--gather layers / layers16)")
        from s3prl_amd.parallel import CopyComm

        copyc = CopyComm(device=dev.index)  # the IPC handles travel over the existing process group (any backend)
    elif world > 1 and args.exchange_via == "cabi" and gather in ("layers", "layers16"):
        from s3prl_amd.parallel import RcclComm

        cabi = RcclComm(device=dev.index)  # the 128-byte RCCL id travels over the existing process group

    def step(exchange=True):
        if world == 1:
            return expert(wavs)  # UpstreamExpert.forward: the metric as SURVEY §8d defines it
        from s3prl_amd.parallel import gather_layers

        with torch.no_grad():
            if gather == "featurized":
                feat = expert.encode_featurized(wavs, feat_w, n_max=n)
                if exchange:
                    gather_layers(feat.unsqueeze(0), out=gathered.unsqueeze(0), algo=args.exchange_algo)
                return feat
            if copyc is not None and exchange:
                copyc.release()  # nothing reads the gathered slab of the previous step: the peers may write while this forward runs
            hs = expert.encode(wavs, n_max=n, out_dtype=args.dtype if gather == "layers16" else None)
            if exchange and gather != "none":
                # one all-gather per layer (hidden_states[l] stays a contiguous (B_global, T, D) block), issued on a
                # side stream as soon as layer l is final so it overlaps the remaining layers' compute
                if copyc is not None:
                    copyc.gather_layers(hs, overlap_events=events)  # (into the slab CopyComm registered for this shape)
                elif cabi is not None:
                    cabi.gather_layers(hs, overlap_events=events, out=gathered, algo=args.exchange_algo)
                else:
                    gather_layers(hs, overlap_events=events, out=gathered, algo=args.exchange_algo)
            return hs

    steal = {"stream": None, "ms": 0.0}

    def occupy():
        """--steal-cus: one launch of idle workgroups per step on a side stream, a little longer than a step — the side stream stays
        busy for the whole timed region (a launch queues behind its predecessor), the compute stream never waits for it."""
        if not args.steal_cus:
            return
        from s3prl_amd import _lib

        if steal["stream"] is None:
            steal["stream"] = torch.cuda.Stream(device=dev)
        _lib.check(_lib.load().s3enc_debug_occupy_cus(args.steal_cus, 256, steal["ms"], steal["stream"].cuda_stream), "s3enc_debug_occupy_cus")

    # average shader clock of a timed region (s3enc_debug_clock_sample): `clock_ghz` in the JSON line, so that a slow box reads as a
    # slow box and not as a regression (round 5: the driver's box ran every kernel 2.0-3.5 % slower than the builder's lease)
    clock = {"buf": torch.zeros((2, 3), dtype=torch.int64, device=dev), "ghz": None}

    def timed(k, exchange=True, profile=False):
        """k steps bracketed by barrier + synchronize on both sides; max over ranks."""
        prof_steps = 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        clk = clock["buf"]
        if clk is not None:  # shader / reference counter pair on the compute stream, in front of the first step (ops.hip)
            _lib.check(_lib.load().s3enc_debug_clock_sample(clk[0].data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        t0 = time.perf_counter()
        for i in range(k):
            on = profile and i % PROF_EVERY == 0
            if profile:
                enc.profile_enable(2 if on else 0)
            prof_steps += int(on)
            occupy()
            step(exchange)
        if clk is not None:  # ... and behind the last one: enqueued, not waited for — nothing is added to the timed region but two
            # one-wave launches; read after the region's own synchronize
            _lib.check(_lib.load().s3enc_debug_clock_sample(clk[1].data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        if steal["stream"] is not None:
            # the compute stream's last step is the end of the timed region; the idle workgroups queued past it are not work
            torch.cuda.current_stream(dev).synchronize()
            el_steal = time.perf_counter() - t0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if clk is not None:
            a, b = clk[0].cpu().tolist(), clk[1].cpu().tolist()
            if a[2] > 0 and b[2] == a[2] and b[1] > a[1]:  # [2] = the reference rate in kHz, written by the sampling workgroup
                clock["ghz"] = (b[0] - a[0]) / (b[1] - a[1]) * a[2] * 1e-6
        if steal["stream"] is not None:
            el = el_steal
        if profile:
            enc.profile_enable(False)
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, prof_steps

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if args.steal_cus:  # size one occupy launch to ~1.2 undisturbed steps
            t_w = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            steal["ms"] = (time.perf_counter() - t_w) / 5 * 1e3 * 1.2
        enc.profile_reset()
        # timed region: HIP events only around the launches of the dominant kernel (the GEMM), and only on every 4th
        # step — an event pair costs a ~5 us bubble on the stream: around all ~290 launches of a forward that is 0.8 ms
        # per batch (2 % fp32, 10 % bf16), around the 55 GEMM launches 0.6 ms.  The full per-kernel breakdown is measured
        # in extra, untimed steps after the timed region.
        PROF_EVERY = 4
        elapsed, prof_steps = timed(steps, True, not args.no_profile)
        headline_clock = clock["ghz"]
        prof = enc.profile_read()
        comm = None
        if world > 1:  # always reported for N > 1: what the exchange moves and how much of it the compute does not hide
            k2 = max(3, min(steps, 30))
            el2 = elapsed / steps * k2
            if gather != "none":
                for _ in range(2):
                    step(False)
                el2, _ = timed(k2, False, False)
            per_state = B * T * D * (2 if gather == "layers16" else 4)
            recv = 0 if gather == "none" else (world - 1) * per_state * (1 if gather == "featurized" else NS)
            comm = {"mode": gather, "algo": args.exchange_algo if gather != "none" else None,
                    "via": (("cabi" if args.exchange_algo == "copy" else args.exchange_via) if gather in ("layers", "layers16") else "torch") if gather != "none" else None,
                    "bytes_received_per_gpu_per_step": int(recv),
                    "ms_per_step_without_exchange": round(el2 / k2 * 1e3, 3),
                    "exposed_ms_per_step": round((elapsed / steps - el2 / k2) * 1e3, 3),
                    "steps_without_exchange": k2 if gather != "none" else 0}
        breakdown, bd_steps = [], 3
        if not args.no_profile:
            enc.profile_reset()
            enc.profile_enable(1)
            for _ in range(bd_steps):
                step(False)
            torch.cuda.synchronize()
            enc.profile_enable(0)
            breakdown = enc.profile_read()
            enc.profile_reset()

    devices = None
    if world > 1:
        ids = [None] * world
        dist.all_gather_object(ids, {"rank": rank, "device": dev.index, "name": torch.cuda.get_device_name(dev)})
        devices = ids
    if rank == 0:
        frames = world * frames_per_batch * steps
        ms_per_step = elapsed / steps * 1e3
        value = frames / elapsed
        gem = [p for p in prof if p["name"].startswith("gemm:")]
        g_ms = sum(p["ms"] for p in gem)
        g_fl = sum(p["flops"] for p in gem)
        g_n = sum(p["launches"] for p in gem)
        g_by = sum(p["bytes"] for p in gem)
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.dtype]
        bd_all = sum(p["ms"] for p in breakdown)
        bd_gemm = sum(p["ms"] for p in breakdown if p["name"].startswith("gemm:"))
        states_note = {"layers": "fp32", "layers16": "16-bit", "featurized": "reduced to the Featurizer's weighted sum", "none": "fp32"}[gather]
        line = {
            "metric": f"encoder-frames/sec (20 ms stride) {MODEL_NAMES.get(args.model, args.model)} {B}x{args.secs:g} s @16 kHz",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "timed_region_s": round(elapsed, 2),
            # average shader clock over the timed region: d(s_memtime) / d(s_memrealtime) x 100 MHz between two one-wave samples on
            # the compute stream (the part clocks to its power budget: 2.4 GHz nominal, ~2.1-2.2 under the fp32 MFMA stream)
            "clock_ghz": round(headline_clock, 3) if headline_clock else None,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": DTYPE_NAMES[args.dtype],
            "data": "synthetic",
            "config": {
                "workload": f"{args.model} random-init, {B}x{args.secs:g} s @16 kHz per GPU, one UpstreamExpert.forward per step: "
                            f"all {NS} hidden_states ({states_note}) written"
                            + (f"; exchange across {world} GPUs: {gather}" if world > 1 else ""),
                "utterances_per_gpu": B, "global_batch": B * world, "samples": n, "frames_per_utt": T, "parallelism": f"dp{world}",
                "lengths": "mixed (utt 0 = max, rest randint(16000, max), padding-masked)" if args.mixed else "equal",
                "gather": gather, "backend": args.backend if world > 1 else None, "rccl": rccl, "devices": devices,
                **({"steal_cus": args.steal_cus} if args.steal_cus else {}),
            },
            # the reference computes padded frames too, so the path's work is B x F_utt(n_max)
            "path_tflops": round(world * B * flops_per_utt(cfg, n) / (ms_per_step * 1e-3) / 1e12, 2),
            "roofline": {
                "kernel": ({"fp32": "gemm_tile_kernel<float> (gemmt.hip)", "fp32x3": "gemm_x3_kernel (gemm_x3.hip)"}.get(args.dtype, "gemm16_big_kernel (gemm16.hip)"))
                          + ": conv1-6 implicit GEMM + proj/qkv/out_proj/fc1/fc2 (+ heads / adapter convolutions where the model has them)",
                "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None, "algorithmic_bytes": round(g_by / max(g_n, 1)),
                "launches_per_step": g_n // max(prof_steps, 1), "avg_launch_ms": round(g_ms / max(g_n, 1), 4),
                "timed_steps_with_events": prof_steps,
                # share of the summed kernel time of a forward, from the all-kernel profiling steps (same steps for both)
                "share_of_kernel_time": round(bd_gemm / bd_all, 3) if bd_all else None,
            },
            # every kernel kind, from the untimed all-kernel profiling steps after the timed region
            "kernels_ms_per_step": {p["name"]: round(p["ms"] / bd_steps, 4) for p in sorted(breakdown, key=lambda p: -p["ms"])},
        }
        if comm:
            line["comm"] = comm
        tr, tr_note = pmc_traffic(args.traffic, args.model, args.dtype, B, args.secs)
        if tr is not None:
            line["roofline"]["traffic"] = tr["gemm_bytes_per_launch"]
            line["roofline"]["traffic_source"] = tr["source"] + f"; kernels {tr['csrc_sha16']}" + (f", commit {tr['commit']}" if tr.get("commit") else "")
        elif tr_note:
            line["roofline"]["traffic_note"] = tr_note
        if world == 1 and not args.no_parity and cfg.family != "distiller":
            # checkers + CPU baseline only; never on the product path
            from oracle import encoder_oracle as O
            from oracle import torch_oracle as TO

            large = cfg.encoder_layers > 12
            Wt = TO.prepare(cfg, weights)
            if args.mixed:
                pick = [0, int(np.argmin(lens))]  # longest + shortest, padded to the batch n_max
            else:
                pick = list(range(max(1, min(args.cpu_sample or (8 if large else 32), B))))
            sample = [wavs[i].cpu() for i in pick]
            ns = len(sample)
            line_cpu = None
            if not args.no_cpu_baseline:
                # the reference's recipe is set_num_threads(os.cpu_count()); on a many-core host that oversubscribes
                # oneDNN (256 threads: 32 s per utterance on the GPU box), so give the CPU path its best thread count:
                # sweep on a small batch, keep the fastest
                ncpu = os.cpu_count() or 1
                sweep = {}
                for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128)}):
                    torch.set_num_threads(th)
                    TO.forward(cfg, Wt, sample[:1], n_max=n)  # warm-up (thread pool, oneDNN primitive cache)
                    t1 = time.perf_counter()
                    TO.forward(cfg, Wt, sample[: min(ns, 2 if large else 4)], n_max=n)
                    sweep[th] = time.perf_counter() - t1
                torch.set_num_threads(min(sweep, key=sweep.get))
            t1 = time.perf_counter()
            ref_t = TO.forward(cfg, Wt, sample, n_max=n)
            cpu_s = time.perf_counter() - t1
            hs = enc.forward([wavs[i] for i in pick], n_max=n)
            torch.cuda.synchronize()
            err_t = max(O.rel_err(hs[l].cpu().numpy(), ref_t[l].numpy()) for l in range(len(ref_t)))
            npar = max(1, min(args.parity_sample if not large else 1, ns))
            sub = pick[-npar:] if args.mixed else pick[:npar]
            ref_n = O.forward(cfg, weights, [wavs[i].cpu().numpy() for i in sub], dtype=np.float32, n_max=n)
            hs_n = enc.forward([wavs[i] for i in sub], n_max=n)
            torch.cuda.synchronize()
            err_n = max(O.rel_err(hs_n[l].cpu().numpy(), ref_n[l]) for l in range(len(ref_n)))
            if not args.no_cpu_baseline:
                cpu_frames = sum(enc.num_frames(lens[i]) for i in pick)
                line["cpu_baseline"] = {
                    "value": round(cpu_frames / cpu_s, 1), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                    "sample": f"{ns} utterance(s) of the same workload (padded to its n_max) through oracle/torch_oracle.py (the "
                              f"reference's ATen call sites, PyTorch {torch.__version__} CPU fp32, {torch.get_num_threads()} "
                              f"threads on {os.cpu_count()} host cores — the fastest of a sweep "
                              f"{ {k: round(v, 2) for k, v in sweep.items()} } s), {cpu_s:.1f} s wall",
                }
            line["parity"] = {
                "max_layer_rel_err_vs_torch_oracle": float(f"{err_t:.3e}"),
                "torch_sample": f"{ns} utterance(s)" + (" (longest + shortest of the mixed batch, batch n_max)" if args.mixed else ""),
                "max_layer_rel_err_vs_numpy_oracle": float(f"{err_n:.3e}"), "numpy_sample": f"{npar} utterance(s)",
                "tolerance": 1e-3, "meets_tolerance": bool(max(err_t, err_n) < 1e-3),
                "note": None if args.dtype in ("fp32", "fp32x3", "fp16x2") else
                        "16-bit operand mode: reported next to its parity, not claimed to meet the 1e-3 target (DESIGN §5)",
            }
            if not args.no_other_modes and not args.mixed and args.dtype == "fp32" and args.model == "hubert_base":
                # side measurement (same workload, same inputs, untimed for the headline): the opt-in operand modes,
                # each with its own parity against the torch restatement of the sample
                other = {}
                out2 = torch.empty((NS, B, T, D), dtype=torch.float32, device=dev)
                for mode in ("fp32x3", "fp16x2", "bf16"):
                    from s3prl_amd.encoder import HipEncoder

                    enc2 = HipEncoder(cfg, weights, dtype=mode, device=dev.index)
                    for _ in range(3):
                        enc2.forward(wavs, out=out2)
                    torch.cuda.synchronize()
                    cs = torch.cuda.current_stream(dev).cuda_stream
                    _lib.check(_lib.load().s3enc_debug_clock_sample(clock["buf"][0].data_ptr(), cs))
                    t1 = time.perf_counter()
                    for _ in range(20):
                        enc2.forward(wavs, out=out2)
                    _lib.check(_lib.load().s3enc_debug_clock_sample(clock["buf"][1].data_ptr(), cs))
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t1) / 20
                    ca, cb = clock["buf"][0].cpu().tolist(), clock["buf"][1].cpu().tolist()
                    ghz = (cb[0] - ca[0]) / (cb[1] - ca[1]) * ca[2] * 1e-6 if ca[2] > 0 and cb[2] == ca[2] and cb[1] > ca[1] else None
                    err = max(O.rel_err(out2[l][:ns].cpu().numpy(), ref_t[l].numpy()) for l in range(NL + 1))
                    other[mode] = {"value": round(B * T / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                                   "clock_ghz": round(ghz, 3) if ghz else None,
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
