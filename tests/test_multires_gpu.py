"""multires-HuBERT (upstream/multires_hubert) on the GPU: beyond the golden-fixture parity of
tests/test_encoder_gpu.py (fp32 / fp32x3 / bf16 / fp16 against the reference expert's outputs) — the frame geometry the
library answers, the hub / expert contract, the 16-bit state output, the weighted sum, data-parallel shards."""

import numpy as np
import pytest

from oracle import encoder_oracle as O

pytestmark = pytest.mark.gpu


def _dev(wavs):
    import torch

    return [torch.from_numpy(w).cuda() for w in wavs]


@pytest.mark.parametrize("name", ["tiny_multires", "tiny_multires3", "tiny_multires_plain"])
def test_library_geometry_equals_the_host_plan(name):
    """s3enc_num_output_frames (C++ mr_plan) against EncoderConfig.multires_plan for every length of a sweep, and the
    forward's actual output shape."""
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config(name)
    enc = HipEncoder(cfg, synth_weights(cfg, 1))
    for n in list(range(2000, 2400, 7)) + [16000, 160000, 159999, 240000]:
        assert enc.num_output_frames(n) == cfg.num_output_frames(n), n
    assert enc.num_states() == cfg.num_hidden_states
    for lengths in ([3300], [3301, 2000], [3620, 3619, 500]):
        hs = enc.forward(_dev(synth_wavs(lengths, 3)))
        assert tuple(hs.shape) == (cfg.num_hidden_states, len(lengths), cfg.num_output_frames(max(lengths)), cfg.encoder_embed_dim)
        assert torch.isfinite(hs).all()
    enc.close()


def test_shard_padded_to_global_nmax_equals_full_batch_and_is_deterministic(golden_loader):
    """SURVEY §8e for the U-net: the GroupNorm(1, D) of the adapters runs per utterance over frames that depend only on the
    global n_max, so a shard padded to it reproduces the full-batch rows bit for bit."""
    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_multires_pad")
    from s3prl_amd.encoder import HipEncoder

    enc = HipEncoder(cfg, weights)
    full = enc.forward(_dev(wavs)).cpu().numpy()
    again = enc.forward(_dev(wavs)).cpu().numpy()
    assert np.array_equal(full, again)
    shard = enc.forward(_dev(wavs[2:]), n_max=max(meta["lengths"])).cpu().numpy()
    assert np.array_equal(shard, full[:, 2:])
    enc.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_16bit_states_are_the_rounded_fp32_states(dtype, golden_loader):
    import torch

    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_multires3_pad")
    from s3prl_amd.encoder import HipEncoder

    enc = HipEncoder(cfg, weights, dtype=dtype)
    a = enc.forward(_dev(wavs))
    b = enc.forward(_dev(wavs), out_dtype=dtype)
    assert b.dtype == (torch.bfloat16 if dtype == "bf16" else torch.float16)
    assert torch.equal(a.to(b.dtype), b)
    enc.close()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["tiny_multires_large_pad", "tiny_multires3_pad"])
@pytest.mark.parametrize("normalize", [False, True])
def test_featurized_equals_the_featurizer_oracle(normalize, name, dtype, golden_loader):
    """The featurize epilogue of the U-net (a state is never written: every block adds w * [layer_norm](state[t / factor])
    into one (B, T_out, D) block) against the weighted sum of the states the same encoder writes."""
    import torch

    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    from s3prl_amd.encoder import HipEncoder

    enc = HipEncoder(cfg, weights, dtype=dtype)
    w = torch.softmax(torch.linspace(-1, 1, cfg.num_hidden_states), 0).tolist()
    w[2] = 0.0
    got = enc.forward_featurized(_dev(wavs), w, normalize=normalize).cpu().numpy()
    hs = enc.forward(_dev(wavs)).cpu().numpy()
    ref = np.zeros(hs[0].shape, dtype=np.float64)  # Featurizer._weighted_sum (nn/upstream.py:312-328) with the given weights
    for wi, h in zip(w, hs.astype(np.float64)):
        ref += wi * (O.layer_norm(h, None, None) if normalize else h)
    assert O.rel_err(got, ref) < 2e-6
    assert not enc.forward_featurized(_dev(wavs), [0.0] * cfg.num_hidden_states, normalize=normalize).any()
    with pytest.raises(Exception, match="one selection"):
        enc.forward(_dev(wavs), selection="fairseq_layers")
    enc.close()


def test_hub_expert_contract_and_layer_events(tmp_path, golden_loader):
    """``multires_hubert_local(ckpt=...)`` on a checkpoint in the reference's converted format: the UpstreamBase dict with
    the reference's hook identifiers, CPU waveforms in -> CPU states out, and one layer event per state."""
    import torch

    import s3prl_amd.hub as hub
    from s3prl_amd.ckpt import save_checkpoint

    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_multires_pad")
    path = str(tmp_path / "mr.pt")
    save_checkpoint(path, cfg, weights)
    expert = hub.multires_hubert_local(ckpt=path, refresh=True)
    assert expert.get_downsample_rates("hidden_states") == 320
    out = expert([torch.from_numpy(w) for w in wavs])
    hs = out["hidden_states"]
    assert len(hs) == cfg.num_hidden_states == len(out["_hidden_states_info"]) and hs[0].device.type == "cpu"
    assert out["_hidden_states_info"][0] == "self.model.encoders[0].layers[0]"
    assert out["_hidden_states_info"][-1] == "self.model.decoders[0]"
    assert out["last_hidden_state"] is hs[-1]
    for l, g in enumerate(golden):
        assert O.rel_err(hs[l].numpy(), g) < 1e-4
    enc = expert._encoder_for(torch.device("cuda", torch.cuda.current_device()))
    evs = enc.layer_events()
    assert len(evs) == cfg.num_hidden_states
    enc.forward(_dev(wavs))
    for ev in evs:
        ev.synchronize()


def test_base_shape_matches_torch_free_oracle_on_a_ragged_batch():
    """The base architecture (D = 768, three 4-layer blocks) on a ragged 3-utterance batch against the numpy oracle, all
    frames (padded ones included: they feed the adapters' GroupNorm statistics)."""
    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config("multires_hubert_base")
    weights = synth_weights(cfg, 5)
    wavs = synth_wavs([9000, 12345, 6000], 6)
    enc = HipEncoder(cfg, weights)
    hs = enc.forward(_dev(wavs)).cpu().numpy()
    ref = O.forward(cfg, weights, wavs, dtype=np.float32)
    errs = [O.rel_err(hs[l], ref[l]) for l in range(len(ref))]
    assert max(errs) < 1e-4, ["%.2e" % e for e in errs]
    enc.close()


@pytest.mark.parametrize("name", ["tiny_multires", "tiny_multires3", "tiny_multires_large"])
def test_extra_short_and_odd_lengths_match_the_oracle(name):
    """EXTRA_SHORT_SEC-style inputs (test/test_upstream.py:24): one or two frames at the finest rate, i.e. ONE frame at
    the coarsest; and odd / even frame counts on both sides of every adapter."""
    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config(name)
    weights = synth_weights(cfg, 21)
    enc = HipEncoder(cfg, weights)
    for lengths in ([400], [800, 400], [1039, 720], [1360, 1359], [2000, 1681, 400]):
        wavs = synth_wavs(lengths, 22)
        hs = enc.forward(_dev(wavs)).cpu().numpy()
        ref = O.forward(cfg, weights, wavs, dtype=np.float32)
        assert hs.shape[2] == ref[0].shape[1] == cfg.num_output_frames(max(lengths))
        errs = [O.rel_err(hs[l], ref[l]) for l in range(len(ref))]
        assert max(errs) < 1e-4, (lengths, ["%.2e" % e for e in errs])
    enc.close()
