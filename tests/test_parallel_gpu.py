"""The data-parallel wrapper on the GPU with the real HIP encoder: 2 ranks (gloo rendezvous, both on cuda:0 — the
GPU box has one device; RCCL refuses two ranks on one GPU) shard a ragged batch, each encodes its shard padded to the
GLOBAL n_max, the per-layer all-gathers run on the side stream behind the library's layer events, and every rank must
end up with the single-process full-batch result."""

import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, lengths, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from s3prl_amd.parallel import DataParallelUpstream
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights
    from s3prl_amd.upstream.hubert.expert import UpstreamExpert

    cfg = named_config("tiny_hubert")
    expert = UpstreamExpert.from_weights(cfg, synth_weights(cfg, 1))
    wavs = [torch.from_numpy(w).cuda() for w in synth_wavs(lengths, 11)]
    dp = DataParallelUpstream(expert, overlap=True)
    hidden = dp(wavs)["hidden_states"]
    torch.cuda.synchronize()
    full = expert(wavs)["hidden_states"] if rank == 0 else None
    torch.cuda.synchronize()
    if rank == 0:
        ret.put(("dp", [h.cpu().numpy() for h in hidden]))
        ret.put(("full", [h.cpu().numpy() for h in full]))
    else:
        ret.put(("sum", float(sum(h.double().abs().sum() for h in hidden))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lengths", [[4000, 2345, 3111, 800], [3000, 1500, 2000]])
def test_two_ranks_on_the_gpu_reproduce_the_full_batch(lengths):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 30500 + (os.getpid() + len(lengths)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=300) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got["dp"]) == len(got["full"])
    for a, b in zip(got["dp"], got["full"]):
        assert a.shape == b.shape
        assert np.array_equal(a, b)  # same kernels, same per-row arithmetic: bit-exact
    total = float(sum(np.abs(h.astype(np.float64)).sum() for h in got["dp"]))
    assert abs(got["sum"] - total) / total < 1e-9  # rank 1 holds the same gathered result
