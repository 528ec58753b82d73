"""hipGraph replay of repeated forwards (s3enc_set_graph_replay): a replay must be bit-identical to the eager forward, read the
CURRENT call's waveforms / lengths (they reach the kernels through the uploaded table, not through kernel arguments), survive
a workspace re-allocation, and work from the NULL stream (private fenced stream) as well as from a side stream."""

import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(wavs):
    import torch

    return [torch.from_numpy(w).cuda() for w in wavs]


@pytest.mark.parametrize("side_stream", [False, True])
@pytest.mark.parametrize("name", ["tiny_hubert_pad", "tiny_wavlm_large_pad", "tiny_multires_pad", "tiny_distiller_pad"])
def test_graph_replay_is_bit_identical_to_eager(name, side_stream, golden_loader):
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import synth_wavs

    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    eager = HipEncoder(cfg, weights)
    enc = HipEncoder(cfg, weights)
    enc.graph_replay(True)
    lengths = meta["lengths"]
    ref = eager.forward(_dev(wavs))
    out = torch.empty_like(ref)
    ctx = torch.cuda.stream(torch.cuda.Stream()) if side_stream else contextlib.nullcontext()
    with ctx:
        dev = _dev(wavs)
        for i in range(4):  # eager, capture + launch, replay, replay
            out.zero_()
            enc.forward(dev, out=out)
            torch.cuda.current_stream().synchronize()
            assert torch.equal(out, ref), f"call {i}"
        assert enc.graph_stats() == {"captures": 1, "replays": 2}
        # same key, other waveforms and other (ragged) lengths below the same n_max: the replay reads the new table
        other = synth_wavs([lengths[0]] + [max(400, n - 137) for n in lengths[1:]], 77)
        ref2 = eager.forward(_dev(other))
        enc.forward(_dev(other), out=out)
        torch.cuda.current_stream().synchronize()
        assert torch.equal(out, ref2)
        assert enc.graph_stats()["replays"] == 3
        # another output block is another key: eager first, then its own graph
        out_b = torch.empty_like(ref)
        for _ in range(3):
            enc.forward(dev, out=out_b)
        torch.cuda.current_stream().synchronize()
        assert torch.equal(out_b, ref) and enc.graph_stats()["captures"] == 2
        # a larger batch grows the workspace: graphs captured before are discarded, not replayed on stale addresses
        big = synth_wavs([2 * max(lengths)] * (len(lengths) + 2), 5)
        ref_big = eager.forward(_dev(big))
        got_big = enc.forward(_dev(big))
        torch.cuda.current_stream().synchronize()
        assert torch.equal(got_big, ref_big)
        for _ in range(2):
            out.zero_()
            enc.forward(dev, out=out)
            torch.cuda.current_stream().synchronize()
            assert torch.equal(out, ref)
    enc.graph_replay(False)
    assert torch.equal(enc.forward(_dev(wavs)), ref)
    enc.close()
    eager.close()


def test_graph_replay_stays_out_of_the_way_of_profiling_events_and_featurize(golden_loader):
    import torch

    from s3prl_amd.encoder import HipEncoder

    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_hubert_pad")
    enc = HipEncoder(cfg, weights)
    enc.graph_replay(True)
    dev = _dev(wavs)
    ref = enc.forward(dev).clone()
    out = torch.empty_like(ref)
    w = [0.25] * cfg.num_hidden_states
    feat = enc.forward_featurized(dev, w).clone()
    for _ in range(3):
        assert torch.equal(enc.forward_featurized(dev, w), feat)  # never captured (its weights are kernel arguments)
    enc.profile_enable(True)
    for _ in range(3):
        enc.forward(dev, out=out)  # profiled forwards stay eager
    torch.cuda.synchronize()
    assert enc.graph_stats() == {"captures": 0, "replays": 0} and torch.equal(out, ref)
    assert sum(p["launches"] for p in enc.profile_read()) > 0
    enc.profile_enable(False)
    enc.layer_events()
    for _ in range(3):
        enc.forward(dev, out=out)  # layer events are recorded per state: eager
    torch.cuda.synchronize()
    assert enc.graph_stats()["captures"] == 0 and torch.equal(out, ref)
    enc.close()


@pytest.mark.parametrize("name", ["tiny_wavlm_large", "tiny_multires3"])
def test_random_call_sequence_with_graph_replay_equals_eager(name):
    """A serving-like stream of calls — a few recurring batch shapes, recurring output blocks, fresh waveforms every call,
    occasionally a new larger shape that re-allocates the workspace — with graph replay on: every result equals the eager
    encoder's, and most calls are replays."""
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config(name)
    weights = synth_weights(cfg, 41)
    eager, enc = HipEncoder(cfg, weights), HipEncoder(cfg, weights)
    enc.graph_replay(True)
    rng = np.random.default_rng(42)
    shapes = [[3000, 2500], [4000], [3500, 3500, 1200]]
    outs = {}
    side = torch.cuda.Stream()
    for it in range(60):
        if it == 35:
            shapes.append([9000, 7000, 8000, 6000])  # bigger than anything before: the workspace grows
        base = shapes[int(rng.integers(len(shapes)))]
        lengths = [base[0]] + [int(n - rng.integers(0, 300)) for n in base[1:]]  # same B and n_max, other ragged lengths
        wavs = synth_wavs(lengths, 100 + it)
        use_side = bool(rng.integers(2))
        with (torch.cuda.stream(side) if use_side else contextlib.nullcontext()):
            dev = _dev(wavs)
            ref = eager.forward(dev)
            key = (len(base), base[0], int(rng.integers(2)))
            if key not in outs:
                outs[key] = torch.empty_like(ref)
            enc.forward(dev, out=outs[key])
            torch.cuda.current_stream().synchronize()
            assert torch.equal(outs[key], ref), (it, lengths, key)
    st = enc.graph_stats()
    assert st["replays"] >= 25 and st["captures"] >= len(outs) - 2, st
    enc.close()
    eager.close()
