"""Host logic of s3prl_amd.nn that needs no GPU: S3PRLUpstream's length matching / re-padding / MIN_SECOND extension and
UpstreamFeaturizer's weight scattering, driven by a stub expert whose "hidden states" come from the numpy oracle, against
the fixtures produced by the reference's own s3prl.nn classes (tests/golden/feat_*.npz)."""

import numpy as np
import pytest
import torch

from oracle import encoder_oracle as O
from test_featurizer_cpu import FEAT, load_feat


class _StubExpert(torch.nn.Module):
    """Same surface as HipUpstreamExpert (num_layers / hidden_sizes / get_downsample_rates / forward dict); the states
    are computed by the oracle — this test is about the Python-side length logic, not the encoder."""

    def __init__(self, cfg, weights):
        super().__init__()
        self.cfg, self.w = cfg, weights
        self.seen_lengths = None

    num_layers = property(lambda self: self.cfg.encoder_layers + 1)
    hidden_sizes = property(lambda self: [self.cfg.encoder_embed_dim] * self.num_layers)

    def get_downsample_rates(self, key=None):
        return self.cfg.downsample_rate

    def forward(self, wavs):
        self.seen_lengths = [int(w.numel()) for w in wavs]
        hs = O.forward(self.cfg, self.w, [w.numpy() for w in wavs], dtype=np.float32)
        return {"hidden_states": tuple(torch.from_numpy(h) for h in hs)}


@pytest.mark.parametrize("name", FEAT)
def test_s3prl_upstream_length_logic_matches_the_reference(name, monkeypatch):
    import s3prl_amd.hub as hub
    from s3prl_amd.nn import S3PRLUpstream

    meta, cfg, weights, wavs, z = load_feat(name)
    stub = _StubExpert(cfg, weights)
    monkeypatch.setattr(hub, "stub_local", lambda **kw: stub, raising=False)
    up = S3PRLUpstream("stub_local", normalize=meta["upstream_normalize"])
    assert up.num_layers == meta["num_layers"] and up.hidden_sizes == [cfg.encoder_embed_dim] * up.num_layers
    n = max(meta["lengths"])
    padded = torch.zeros(len(wavs), n, 1)  # (B, n, 1) is accepted like (B, n) (nn/upstream.py:180-181)
    for b, w in enumerate(wavs):
        padded[b, : len(w), 0] = torch.from_numpy(w)
    all_hs, all_lens = up(padded, torch.tensor(meta["lengths"]))
    assert stub.seen_lengths == [x + (800 - n if n < 800 else 0) for x in meta["lengths"]]  # MIN_SECOND = 0.05 s
    for l, h in enumerate(all_hs):
        assert tuple(h.shape) == z[f"hs{l}"].shape
        assert O.rel_err(h.numpy(), z[f"hs{l}"]) < 1e-4
        assert np.array_equal(all_lens[l].numpy(), z["lens"][l])
    with pytest.raises(NotImplementedError):  # an expert without a host copy of its weights cannot be re-drawn
        S3PRLUpstream("stub_local", randomize=True)


def test_randomize_redraws_every_tensor_like_randomize_upstream():
    """nn/upstream.py:27-35: vectors ~ N(mean, std) of themselves, >= 2-D tensors Xavier-normal (receptive field included)."""
    from s3prl_amd.nn import randomize_weights
    from s3prl_amd.synth import named_config, synth_weights
    from s3prl_amd.upstream.hubert.expert import UpstreamExpert

    cfg = named_config("tiny_hubert")
    w = synth_weights(cfg, 3)
    r = randomize_weights(w, seed=5)
    assert set(r) == set(w) and all(r[k].shape == np.asarray(w[k]).shape and r[k].dtype == np.float32 for k in w)
    assert all(not np.array_equal(r[k], w[k]) for k in w if np.asarray(w[k]).size > 1)
    k2 = max((k for k in w if np.asarray(w[k]).ndim == 2), key=lambda k: np.asarray(w[k]).size)
    fan = sum(np.asarray(w[k2]).shape)
    assert abs(r[k2].std() / (2.0 / fan) ** 0.5 - 1) < 0.1 and abs(r[k2].mean()) < 0.05 * r[k2].std() + 1e-3
    k3 = next(k for k in w if np.asarray(w[k]).ndim == 3 and np.asarray(w[k]).shape[2] > 1 and np.asarray(w[k]).size > 4096)
    sh = np.asarray(w[k3]).shape
    assert abs(r[k3].std() / (2.0 / ((sh[0] + sh[1]) * sh[2])) ** 0.5 - 1) < 0.1
    k1 = max((k for k in w if np.asarray(w[k]).ndim == 1), key=lambda k: np.asarray(w[k]).size)
    assert abs(r[k1].mean() - np.asarray(w[k1]).mean()) < 4 * np.asarray(w[k1]).std() / np.sqrt(r[k1].size) + 1e-6
    assert np.array_equal(randomize_weights(w, seed=5)[k2], r[k2])  # seeded: reproducible
    expert = UpstreamExpert.from_weights(cfg, w)
    expert._encoders[0] = object()
    expert.randomize_(seed=1)
    assert not expert._encoders and not np.array_equal(expert._weights[k2], w[k2])


LEGACY = [n for n in __import__("conftest").golden_names(encoder_only=False) if n.startswith("legacyfeat_")]


class _DictExpert(torch.nn.Module):
    """Returns the result dict of a reference expert (its states come from the fixture) for the fixture's batch, and a
    same-shaped dummy for the probe forward of the constructor."""

    def __init__(self, states, lengths, rate=320):
        super().__init__()
        self.states, self.lengths, self.rate = states, lengths, rate

    def get_downsample_rates(self, key=None):
        return self.rate

    def forward(self, wavs):
        if [int(w.numel()) for w in wavs] == self.lengths:
            hs = tuple(torch.from_numpy(h) for h in self.states)
        else:
            hs = tuple(torch.zeros(len(wavs), max(1, max(int(w.numel()) for w in wavs) // self.rate), h.shape[-1]) for h in self.states)
        out = {"hidden_states": hs, "last_hidden_state": hs[-1]}
        out.update({f"hidden_state_{i}": h for i, h in enumerate(hs)})
        return out


@pytest.mark.parametrize("name", LEGACY)
def test_legacy_featurizer_matches_the_reference_class(name):
    """s3prl.upstream.interfaces.Featurizer (the class downstream/runner.py builds) ran on a reference expert to make these
    fixtures; the mirror must pick the same feature, apply the same weighted sum and cut the same per-utterance lengths."""
    import json
    import os

    from conftest import GOLDEN_DIR
    from s3prl_amd.nn import LegacyFeaturizer
    from s3prl_amd.synth import synth_wavs

    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    states = [z[f"hs{l}"] for l in range(meta["num_states"])]
    expert = _DictExpert(states, meta["lengths"])
    fz = LegacyFeaturizer(expert, feature_selection=meta["feature_selection"], upstream_device="cpu",
                          layer_selection=meta["layer_selection"], normalize=meta["normalize"])
    assert fz.feature_selection == meta["resolved_selection"] and fz.output_dim == meta["output_dim"]
    assert fz.downsample_rate == meta["downsample_rate"]
    if "feat_weights" in z:
        with torch.no_grad():
            fz.weights.copy_(torch.from_numpy(z["feat_weights"]))
    else:
        assert not hasattr(fz, "weights")
    wavs = [torch.from_numpy(w) for w in synth_wavs(meta["lengths"], meta["wav_seed"])]
    outs = fz(wavs, expert(wavs))
    assert len(outs) == len(meta["lengths"])
    for b, o in enumerate(outs):
        assert tuple(o.shape) == z[f"out{b}"].shape
        assert O.rel_err(o.detach().numpy(), z[f"out{b}"]) < 1e-5, f"{name} utterance {b}"


def test_upstream_featurizer_scatters_softmax_weights():
    import types

    from s3prl_amd.nn import Featurizer, UpstreamFeaturizer

    up = types.SimpleNamespace(num_layers=5, hidden_sizes=[8] * 5, downsample_rates=[320] * 5, normalize=False, upstream=None)
    fz = Featurizer(up, layer_selections=[4, 0, 2])
    with torch.no_grad():
        fz.weights.copy_(torch.tensor([0.3, -1.0, 2.0]))
    w = UpstreamFeaturizer(up, fz).layer_weights()
    sm = torch.softmax(torch.tensor([0.3, -1.0, 2.0]), 0).tolist()
    assert w[1] == 0.0 and w[3] == 0.0 and np.allclose([w[0], w[2], w[4]], sm) and abs(sum(w) - 1) < 1e-6
