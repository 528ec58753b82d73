"""world_size-2 (and 3) gloo runs of the data-parallel path on CPU: sharding + per-layer all-gather must reproduce
the single-process full-batch result.  The compute inside each rank is the numpy oracle (test infrastructure) —
on the GPU the same code path wraps the HIP encoder (tests/test_encoder_gpu.py covers the shard==full property)."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, lengths, ret, algo="ring"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import encoder_oracle as O
    from s3prl_amd.parallel import encode_data_parallel
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config("tiny_hubert")
    weights = synth_weights(cfg, 1)
    wavs = [torch.from_numpy(w) for w in synth_wavs(lengths, 11)]

    def encode_fn(shard, n_max):
        hs = O.forward(cfg, weights, [w.numpy() for w in shard], dtype=np.float32, n_max=n_max)
        return torch.from_numpy(np.stack(hs))

    hidden = encode_data_parallel(encode_fn, wavs, algo=algo)
    if rank == 0:
        ret.put([h.numpy() for h in hidden])
    else:
        ret.put(float(sum(h.abs().sum() for h in hidden)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lengths,algo", [(2, [4000, 2345, 3111, 800], "ring"), (3, [4000, 2345, 3111, 800], "ring"),
                                                (2, [3000, 1500, 2000], "ring"),
                                                # the all-pairs send / receive form of the exchange (xGMI is point-to-point)
                                                (2, [4000, 2345, 3111, 800], "direct"), (3, [4000, 2345, 3111, 800], "direct"),
                                                (3, [3000, 1500, 2000, 900, 1200], "direct")])
def test_gloo_data_parallel_equals_full_batch(world, lengths, algo):
    from oracle import encoder_oracle as O
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() + world * 7 + len(lengths) + 13 * (algo == "direct")) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, ret, algo)) for r in range(world)]
    for p in procs:
        p.start()
    results = [ret.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hidden = next(r for r in results if isinstance(r, list))
    sums = [r for r in results if not isinstance(r, list)]
    cfg = named_config("tiny_hubert")
    full = O.forward(cfg, synth_weights(cfg, 1), synth_wavs(lengths, 11), dtype=np.float32)
    assert len(hidden) == len(full)
    for h, f in zip(hidden, full):
        assert h.shape == f.shape
        assert O.rel_err(h, f) < 1e-5
    total = float(sum(np.abs(h).sum() for h in hidden))
    for s in sums:  # every rank holds the same gathered result
        assert abs(s - total) / total < 1e-5


def test_shard_bounds_cover_batch():
    from s3prl_amd.parallel import shard_bounds

    for B in range(1, 20):
        for world in range(1, 9):
            seen = []
            for r in range(world):
                b, e, per = shard_bounds(B, world, r)
                assert 0 <= e - b <= per
                seen += list(range(b, e))
            assert seen == list(range(B))


def _worker_feat(rank, world, port, lengths, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import encoder_oracle as O
    from s3prl_amd.parallel import featurize_data_parallel
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config("tiny_hubert")
    weights = synth_weights(cfg, 1)
    wavs = [torch.from_numpy(w) for w in synth_wavs(lengths, 11)]
    lw = torch.softmax(torch.linspace(-1, 1, cfg.encoder_layers + 1), 0)

    def encode_fn(shard, n_max):
        return torch.from_numpy(np.stack(O.forward(cfg, weights, [w.numpy() for w in shard], dtype=np.float32, n_max=n_max)))

    feat = featurize_data_parallel(encode_fn, lambda hs: (lw.view(-1, 1, 1, 1) * hs).sum(0), wavs)
    ret.put((rank, feat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_featurize_before_gather_equals_full_batch():
    """§8f-1: weighted sum on the shard, then ONE all-gather of (B, T, D) — same result as featurizing the full batch."""
    from oracle import encoder_oracle as O
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    world, lengths = 2, [4000, 2345, 3111]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_feat, args=(r, world, port, lengths, ret)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(ret.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = named_config("tiny_hubert")
    full = np.stack(O.forward(cfg, synth_weights(cfg, 1), synth_wavs(lengths, 11), dtype=np.float32))
    lw = torch.softmax(torch.linspace(-1, 1, cfg.encoder_layers + 1), 0).numpy()
    ref = (lw[:, None, None, None] * full).sum(0)
    for r in range(world):
        assert results[r].shape == ref.shape
        assert O.rel_err(results[r], ref) < 1e-5


def _worker_16(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from s3prl_amd.parallel import gather_layers

    out = {}
    for dt in (torch.bfloat16, torch.float16):
        hs = (torch.arange(3 * 2 * 4 * 8, dtype=torch.float32).reshape(3, 2, 4, 8) * (rank + 1) / 64).to(dt)
        g = gather_layers(hs)
        out[str(dt)] = (g.dtype == dt, tuple(g.shape), g.float())
    ret.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_gather_of_16bit_states_moves_bytes():
    """16-bit states (s3enc_forward_ex out_dtype) go through the per-layer gathers as raw bytes: same values, same dtype,
    rank-major order."""
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_16, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(ret.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = torch.arange(3 * 2 * 4 * 8, dtype=torch.float32).reshape(3, 2, 4, 8) / 64
    for r in range(world):
        for dt in (torch.bfloat16, torch.float16):
            ok, shape, g = results[r][str(dt)]
            assert ok and shape == (3, 4, 4, 8)
            assert torch.equal(g[:, :2], base.to(dt).float()) and torch.equal(g[:, 2:], (base * 2).to(dt).float())
