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


def _worker_modes(rank, world, port, lengths, ret):
    try:
        _worker_modes_body(rank, world, port, lengths, ret)
    except Exception:  # a crashed rank must not leave the parent (and the GPU box) waiting for its timeout
        import traceback

        ret.put(("error", traceback.format_exc()))
        os._exit(1)


def _worker_modes_body(rank, world, port, lengths, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from s3prl_amd.parallel import encode_data_parallel, featurized_data_parallel
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights
    from s3prl_amd.upstream.wavlm.expert import UpstreamExpert

    cfg = named_config("tiny_wavlm_large")  # pre-LN, gated relative-position bias
    weights = synth_weights(cfg, 8)
    wavs = [torch.from_numpy(w).cuda() for w in synth_wavs(lengths, 18)]
    lw = torch.softmax(torch.linspace(-1, 1, cfg.encoder_layers + 1), 0).tolist()
    # (1) Featurizer epilogue inside the encoder + ONE all-gather
    fp32 = UpstreamExpert.from_weights(cfg, weights)
    feat = featurized_data_parallel(fp32, lw, wavs, normalize=True)
    ref = fp32.encode_featurized(wavs, lw, True) if rank == 0 else None
    # (2) 16-bit states: half the bytes through the exchange
    bf = UpstreamExpert.from_weights(cfg, weights, dtype="bf16")
    hs16 = encode_data_parallel(lambda mine, n_max: bf.encode(mine, n_max=n_max, out_dtype="bf16"), wavs)
    full16 = bf.encode(wavs, out_dtype="bf16") if rank == 0 else None
    torch.cuda.synchronize()
    if rank == 0:
        ret.put(("feat", feat.cpu().numpy(), ref.cpu().numpy()))
        ret.put(("hs16", torch.stack(hs16).float().cpu().numpy(), full16.float().cpu().numpy()))
    else:
        ret.put(("sum", float(feat.double().abs().sum()), float(torch.stack(hs16).double().abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_featurized_and_16bit_exchange():
    """The reduced-exchange modes of the data-parallel path: the weighted sum computed by the encoder's own
    epilogue on each shard + one (B, T, D) all-gather, and 16-bit states through the per-layer gathers — both must equal
    the single-process result bit for bit (same kernels, same rows; every batch coupling goes through the global n_max)."""
    import torch.multiprocessing as mp

    lengths = [4000, 2345, 3111]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 32500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_modes, args=(r, 2, port, lengths, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    for _ in range(3):
        item = ret.get(timeout=180)
        if item[0] == "error":
            for p in procs:
                p.kill()
            pytest.fail(item[1])
        got.append(item)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by = {g[0]: g[1:] for g in got}
    assert np.array_equal(by["feat"][0], by["feat"][1])
    assert np.array_equal(by["hs16"][0], by["hs16"][1])
    assert abs(by["sum"][0] - float(np.abs(by["feat"][0].astype(np.float64)).sum())) < 1e-6 * by["sum"][0]
    assert abs(by["sum"][1] - float(np.abs(by["hs16"][0].astype(np.float64)).sum())) < 1e-6 * by["sum"][1]


def test_cabi_rccl_exchange_world_1():
    """The exchange behind the C ABI (s3enc_comm_*: RCCL dlopen'ed by libs3enc.so) on a one-rank communicator — the only
    world size this box's single GPU allows RCCL (it refuses two ranks on one device): per-state all-gathers on the
    communicator's stream, each behind the encoder's "state l final" event, reproduce the slab bit for bit, for a strided
    receive buffer too; the same entry points serve N ranks."""
    import torch

    from s3prl_amd.parallel import RcclComm
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights
    from s3prl_amd.upstream.hubert.expert import UpstreamExpert

    cfg = named_config("tiny_hubert")
    expert = UpstreamExpert.from_weights(cfg, synth_weights(cfg, 2))
    wavs = [torch.from_numpy(w).cuda() for w in synth_wavs([4000, 3111, 2345], 9)]
    comm = RcclComm()
    assert (comm.world, comm.rank) == (1, 0)
    enc = expert._encoder_for(wavs[0].device)
    events = enc.layer_events()
    hs = expert.encode(wavs)  # (NS, B, T, D); the events were recorded as each state became final
    out = comm.gather_layers(hs, overlap_events=events)
    torch.cuda.synchronize()
    assert torch.equal(out, hs)
    # a receive slab whose state stride is wider than one rank's block (what an N-rank gather writes into)
    flat = torch.full((hs.shape[0], 3 * hs[0].numel()), float("nan"), device="cuda")
    view = flat.as_strided(hs.shape, (flat.stride(0),) + tuple(hs[0].stride()))
    comm.gather_layers(hs, out=view)
    torch.cuda.synchronize()
    assert torch.equal(view, hs) and torch.isnan(flat[:, hs[0].numel():]).all()
    # the all-pairs form (S3ENC_EXCHANGE_DIRECT: grouped ncclSend / ncclRecv, one peer per xGMI link): same layout contract
    flat.fill_(float("nan"))
    comm.gather_layers(hs, overlap_events=events, out=view, algo="direct")
    torch.cuda.synchronize()
    assert torch.equal(view, hs) and torch.isnan(flat[:, hs[0].numel():]).all()
    assert torch.equal(comm.gather_layers(hs, algo="direct"), hs)
    with pytest.raises(ValueError):
        comm.gather_layers(hs, overlap_events=events[:2])  # fewer events than states must not reach hipStreamWaitEvent(NULL)
    comm.close()


def test_all_pairs_exchange_executes_nccl_send_recv_on_one_gpu():
    """Round 6: before this test the all-pairs exchange (S3ENC_EXCHANGE_DIRECT) had never executed an ncclSend / ncclRecv — at
    world 1 comm.hip returned before ncclGroupStart and RCCL refuses two ranks on one device.  Tuning key `comm_self_p2p`: the
    rank's own block travels as a send-to-self / receive-from-self pair inside the state's group, so the dlopen'ed symbols'
    signatures, the byte counts / datatype, the group bracketing and the stream order behind the "state l final" events all run
    for real; same bytes in the same places as the device copy it replaces."""
    import torch

    from s3prl_amd import _lib
    from s3prl_amd.parallel import RcclComm
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights
    from s3prl_amd.upstream.hubert.expert import UpstreamExpert

    cfg = named_config("tiny_hubert")
    expert = UpstreamExpert.from_weights(cfg, synth_weights(cfg, 2))
    wavs = [torch.from_numpy(w).cuda() for w in synth_wavs([4000, 3111, 2345], 9)]
    lib = _lib.load()
    comm = RcclComm()
    enc = expert._encoder_for(wavs[0].device)
    events = enc.layer_events()
    try:
        _lib.check(lib.s3enc_set_tuning(b"comm_self_p2p", 1))
        for out_dtype in (None, "fp32"):
            flat = None
            for _ in range(3):  # repeated: the receive of call i + 1 must not overtake the encoder's writes of call i + 1
                hs = expert.encode(wavs)
                if flat is None:
                    flat = torch.full((hs.shape[0], 2 * hs[0].numel() + 5), float("nan"), device="cuda")
                    view = flat.as_strided(hs.shape, (flat.stride(0),) + tuple(hs[0].stride()))
                flat.fill_(float("nan"))
                torch.cuda.synchronize()  # (with layer events the exchange of state l waits for event l only, not for the fill above)
                comm.gather_layers(hs, overlap_events=events, out=view, algo="direct")
                torch.cuda.synchronize()
                assert torch.equal(view, hs) and torch.isnan(flat[:, hs[0].numel():]).all()
        # without events (ordered behind the caller's stream), contiguous output
        hs = expert.encode(wavs)
        assert torch.equal(comm.gather_layers(hs, algo="direct"), hs)
    finally:
        _lib.check(lib.s3enc_set_tuning(b"comm_self_p2p", 0))
        comm.close()


def _worker_copy(rank, world, port, lengths, ret):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["S3ENC_COPY_DEADLINE_MS"] = "20000"   # (both ranks share one GPU and a cold start here: be generous, stay bounded)
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from s3prl_amd.parallel import CopyComm, shard_bounds
        from s3prl_amd.synth import named_config, synth_wavs, synth_weights
        from s3prl_amd.upstream.hubert.expert import UpstreamExpert

        cfg = named_config("tiny_hubert")
        expert = UpstreamExpert.from_weights(cfg, synth_weights(cfg, 1))
        wavs = [torch.from_numpy(w).cuda() for w in synth_wavs(lengths, 11)]
        n_max = max(lengths)
        beg, end, per = shard_bounds(len(wavs), world, rank)
        mine = wavs[beg:end]
        enc = expert._encoder_for(wavs[0].device)
        events = enc.layer_events()
        comm = CopyComm()
        outs = []
        for step in range(4):  # repeated exchanges into the SAME slab: the ack protocol must keep step i + 1 out of step i's readers
            scale = 1.0 + step
            if step >= 2:
                comm.release()  # (steps 2, 3: the overlapped form — the clone below, the slab's only reader, is already enqueued)
            hs = expert.encode([w * scale for w in mine], n_max=n_max)
            got = comm.gather_layers(hs, overlap_events=events)
            outs.append(got.clone())  # (stream-ordered behind the exchange; the slab is overwritten by the next step)
        hs16 = expert.encode(mine, n_max=n_max).half()  # a second slab shape / dtype: its own registration
        got16 = comm.gather_layers(hs16).clone()
        torch.cuda.synchronize()
        assert comm.status() == 0, f"a wait of the copy exchange timed out: {comm.status():#x}"
        full = [expert.encode([w * (1.0 + step) for w in wavs]) for step in range(4)]
        torch.cuda.synchronize()
        ok = all(torch.equal(o, f) for o, f in zip(outs, full)) and torch.equal(got16, full[0].half())
        ret.put(("rank", rank, bool(ok), [float(o.double().abs().sum()) for o in outs]))
        dist.barrier()
        comm.close()
        dist.destroy_process_group()
    except Exception:
        import traceback

        ret.put(("error", traceback.format_exc()))
        os._exit(1)


def test_copy_engine_exchange_two_processes_one_gpu():
    """S3ENC_EXCHANGE_COPY (round 6; include/s3enc.h): two PROCESSES on this box's single GPU map each other's receive slab and
    mailbox through hipIpcGetMemHandle / hipIpcOpenMemHandle and push their shard's states with one hipMemcpyAsync per state and
    peer, behind the encoder's layer events.  Layout (rank r's block at [l][r]), ordering (four back-to-back exchanges into one
    slab; a second slab of another dtype) and the deadline word are checked against the single-process full-batch result, bit for
    bit, on both ranks.  On a real node the same code moves the blocks over xGMI on the SDMA engines."""
    import torch.multiprocessing as mp

    lengths = [4000, 2345, 3111, 800]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 34500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_copy, args=(r, 2, port, lengths, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    for _ in range(2):
        item = ret.get(timeout=150)
        if item[0] == "error":
            for p in procs:
                p.kill()
            pytest.fail(item[1])
        got.append(item)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(g[1] for g in got) == [0, 1] and all(g[2] for g in got), got
    assert got[0][3] == got[1][3]   # both ranks hold the same gathered states
