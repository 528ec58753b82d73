"""The Featurizer / S3PRLUpstream restatement (oracle/featurizer_oracle.py) against fixtures produced by RUNNING the
reference's ``s3prl.nn.S3PRLUpstream`` + ``s3prl.nn.Featurizer`` (tests/golden/feat_*.npz) — this pins the consumer-side
oracle like test_oracle_golden.py pins the encoder oracle."""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_names
from oracle import encoder_oracle as O
from oracle import featurizer_oracle as FO

FEAT = [n for n in golden_names(encoder_only=False) if n.startswith("feat_")]


def load_feat(name):
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = named_config(meta["config"])
    return meta, cfg, synth_weights(cfg, meta["weight_seed"]), synth_wavs(meta["lengths"], meta["wav_seed"]), z


@pytest.mark.parametrize("name", FEAT)
def test_featurizer_oracle_matches_reference(name):
    meta, cfg, weights, wavs, z = load_feat(name)
    enc_len = FO.encoded_lengths(meta["lengths"])
    padded = [np.concatenate([w, np.zeros(n - len(w), np.float32)]) for w, n in zip(wavs, enc_len)]
    hs = O.forward(cfg, weights, padded, dtype=np.float32)
    all_hs, all_lens = FO.upstream_outputs(hs, meta["lengths"], cfg.downsample_rate, meta["upstream_normalize"])
    assert len(all_hs) == meta["num_layers"]
    for l, h in enumerate(all_hs):
        assert h.shape == z[f"hs{l}"].shape
        assert O.rel_err(h, z[f"hs{l}"]) < 1e-4, f"{name} layer {l}"
        assert np.array_equal(all_lens[l], z["lens"][l])
    feat = FO.weighted_sum(all_hs, z["feat_weights"], meta["layer_selections"], meta["featurizer_normalize"])
    assert feat.shape == z["feat"].shape and O.rel_err(feat, z["feat"]) < 1e-4
    assert np.array_equal(all_lens[0], z["feat_len"])
