"""The numpy oracle (oracle/encoder_oracle.py) against fixtures produced by RUNNING THE REFERENCE
(tests/golden/make_golden.py, PyTorch CPU fp32).  This is what pins the oracle."""

import numpy as np
import pytest

from conftest import golden_meta, golden_names
from oracle import encoder_oracle as O

# fp32-vs-fp32 of two different summation orders; the reference's own regression tolerance is atol 1e-2
# (test/test_upstream.py:22)
REL_TOL = 1e-4


def _cpu_oracle_cases():
    """Every fixture but the bulk of round 6's weight-seed sweep: the sweep exists for the GPU modes (tests/test_encoder_gpu.py runs
    all 32 of its fixtures); the oracles are pinned by seed 1 of its three base models and by the 56 fixtures of earlier rounds (which
    hold both large models at three shapes each) — the other 29 would add a quarter of an hour of CPU time and no new code path."""
    keep = {"hubert_base_s1_pl", "wav2vec2_base_s1_pl", "data2vec_base_s1_pl"}  # (the large models: `*_large_pl`, `*_large_10s*`, `*_15s_*` of rounds 2-5)
    return [n for n in golden_names() if not golden_meta(n).get("seed_sweep") or n in keep]


# the numpy restatement is O(minutes) on the full-size batches: one full-size fixture per large model stays (hubert_large_10s,
# wavlm_large_15s_pad); their pretrained-like twins are pinned through the torch restatement below (same fixtures, a tenth of the time)
_NUMPY_SKIP = {"hubert_large_10s_pl", "wavlm_large_15s_pl"}


@pytest.mark.parametrize("name", [n for n in _cpu_oracle_cases() if n not in _NUMPY_SKIP])
def test_oracle_matches_reference_golden(name, golden_loader):
    meta, cfg, weights, wavs, golden, norms = golden_loader(name)
    hs = O.forward(cfg, weights, wavs, dtype=np.float32, selection=meta.get("selection"))
    assert len(hs) == meta.get("n_states", cfg.encoder_layers + 1) == len(golden)
    assert list(hs[0].shape) == meta["shape"]
    ts, cs = meta["t_stride"], meta["c_stride"]
    for l, (h, g) in enumerate(zip(hs, golden)):
        assert h.dtype == np.float32
        err = O.rel_err(h[:, ::ts, ::cs], g)
        assert err < REL_TOL, f"{name} layer {l}: rel-err {err:.3e}"
        full = np.linalg.norm(h.astype(np.float64))
        assert abs(full - norms[l]) / norms[l] < REL_TOL


def test_frames_and_mask_rules():
    """Closed forms of SURVEY A.2 (hubert chunk rule vs wav2vec2 conv-length rule)."""
    from s3prl_amd.synth import named_config

    hub, w2v = named_config("hubert_base"), named_config("wav2vec2_base")
    assert hub.conv_lengths(160000) == [31999, 15999, 7999, 3999, 1999, 999, 499]
    assert hub.conv_lengths(240000)[-1] == 749
    assert hub.num_frames(800) == 2  # EXTRA_SHORT_SEC = 0.05 s (test/test_upstream.py:24)
    assert hub.downsample_rate == 320
    # len = 16123 in a 160000-sample batch: HuBERT keeps 51 frames, wav2vec2 50
    assert hub.valid_frames(16123, 160000) == 51
    assert w2v.valid_frames(16123, 160000) == 50
    # n_max = 32000 → T = 99, chunk = 323
    assert hub.num_frames(32000) == 99 and hub.valid_frames(32000, 32000) == 99
    assert hub.valid_frames(323 * 10 + 1, 32000) == 11


def test_shard_padded_to_global_max_reproduces_full_batch(golden_loader):
    """a data-parallel shard must be padded to the GLOBAL n_max; then its rows equal
    the single-batch reference rows."""
    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_hubert_pad")
    n_max = max(meta["lengths"])
    shard = O.forward(cfg, weights, wavs[2:], dtype=np.float32, n_max=n_max)
    for l, g in enumerate(golden):
        assert O.rel_err(shard[l], g[2:]) < REL_TOL
    local = O.forward(cfg, weights, wavs[2:], dtype=np.float32)  # padded to the shard's own max: different
    assert local[0].shape[1] != golden[0].shape[1] or O.rel_err(local[0], golden[0][2:]) > 1e-2


def _torch_oracle_cases():
    # the ATen-call-site restatement covers the families bench.py times (no DistilHuBERT heads, no feature_selection)
    return [n for n in _cpu_oracle_cases() if not golden_meta(n).get("selection") and "distil" not in golden_meta(n)["config"]]


@pytest.mark.parametrize("name", _torch_oracle_cases())
def test_torch_oracle_matches_reference_golden(name, golden_loader):
    """oracle/torch_oracle.py (the ATen-call-site restatement timed as bench.py's cpu_baseline) against the same
    reference-generated fixtures."""
    import torch

    from oracle import torch_oracle as TO

    meta, cfg, weights, wavs, golden, norms = golden_loader(name)
    hs = TO.forward(cfg, TO.prepare(cfg, weights), [torch.from_numpy(w) for w in wavs])
    assert len(hs) == len(golden)
    ts, cs = meta["t_stride"], meta["c_stride"]
    for l, (h, g) in enumerate(zip(hs, golden)):
        h = h.numpy()
        assert list(h.shape) == meta["shape"]
        err = O.rel_err(h[:, ::ts, ::cs], g)
        assert err < REL_TOL, f"{name} layer {l}: rel-err {err:.3e}"
        assert abs(np.linalg.norm(h.astype(np.float64)) - norms[l]) / norms[l] < REL_TOL
