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
    with pytest.raises(NotImplementedError):
        S3PRLUpstream("stub_local", randomize=True)


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
