"""Seeded sweep over architectures the C ABI accepts — widths, depths, conv stacks, positional-conv shapes, every family and
norm placement, ragged batches — against the numpy oracle (fp32: the 1e-4 bar of the golden tests; bf16: finite and
within its usual band).  The goldens pin the released architectures; this guards the generality the config struct promises."""

import numpy as np
import pytest

from oracle import encoder_oracle as O

pytestmark = pytest.mark.gpu


def _random_config(rng, medium=False):
    from s3prl_amd.config import EncoderConfig

    family = str(rng.choice(["hubert", "wav2vec2", "wavlm", "distiller", "multires_hubert"]))
    heads = int(rng.choice([6, 8, 12] if medium else [1, 2, 3, 4]))
    D = 64 * heads
    groups = [g for g in (1, 2, 3, 4, 6, 8, 12, 16, 24) if D % g == 0 and D // g in (32, 48, 64)]
    C = int(rng.choice([128, 256, 512] if medium else [32, 64, 96]))
    n_mid = int(rng.integers(1, 5))
    conv = [(C, 10, 5)] + [(C, int(rng.choice([2, 3])), int(rng.choice([1, 2]))) for _ in range(n_mid)] + [(C, 2, 2)]
    kw = dict(family=family, conv_layers=conv, encoder_embed_dim=D, encoder_attention_heads=heads,
              encoder_ffn_embed_dim=int(rng.choice([512, 1000, 1536] if medium else [64, 136, 256, 320])),
              encoder_layers=int(rng.integers(1, 3 if medium else 4)),
              conv_pos=int(rng.choice([3, 8, 15, 16, 31, 32])), conv_pos_groups=int(rng.choice(groups)),
              layer_norm_first=bool(rng.integers(2)), extractor_mode=str(rng.choice(["default", "layer_norm"])),
              conv_bias=bool(rng.integers(2)), normalize=bool(rng.integers(2)))
    if family == "wavlm" and rng.integers(2):
        kw.update(relative_position_embedding=True, num_buckets=int(rng.choice([16, 32, 64])), max_distance=int(rng.choice([40, 64, 128])),
                  gru_rel_pos=bool(rng.integers(2)))
    if family == "distiller":
        kw.update(feature_layer_norm=False, pred_heads=int(rng.integers(1, 4)), normalize=False)
    if family == "wav2vec2" and rng.integers(3) == 0:  # data2vec-audio positional stack
        kw.update(pos_conv_depth=int(rng.integers(2, 5)), conv_pos=int(rng.choice([9, 15, 20])))
    if family == "multires_hubert":
        pairs = int(rng.integers(1, 3))
        plain = bool(rng.integers(2))
        blocks = [int(rng.integers(1, 3)) for _ in range(2 * pairs + 1)]
        kw.update(label_rate_ratios=[1, 2] * pairs if plain else [int(x) for p in range(pairs) for x in ((1, 2) if rng.integers(2) else (2, 3))],
                  block_layers=blocks, encoder_layers=sum(blocks), use_plain_updownsample=plain,
                  conv_adapter_kernel=int(rng.choice([7, 13])) if not plain else 7)
    cfg = EncoderConfig(**kw)
    cfg.validate()
    return cfg


@pytest.mark.parametrize("seed", range(40))
def test_random_architecture_matches_the_oracle(seed):
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import synth_wavs, synth_weights

    rng = np.random.default_rng(1000 + seed)
    cfg = _random_config(rng)
    weights = synth_weights(cfg, seed)
    rate = cfg.downsample_rate
    B = int(rng.integers(1, 5))
    lengths = [int(rng.integers(14 * rate, 40 * rate)) for _ in range(B)]
    wavs = synth_wavs(lengths, seed + 1, dc=float(rng.choice([0.0, 0.2])), scale=float(rng.choice([1.0, 0.1])))
    ref = O.forward(cfg, weights, wavs, dtype=np.float32)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    enc = HipEncoder(cfg, weights)
    hs = enc.forward(dev).cpu().numpy()
    assert hs.shape == (len(ref),) + ref[0].shape, (cfg, hs.shape, ref[0].shape)
    errs = [O.rel_err(hs[l], ref[l]) for l in range(len(ref))]
    assert max(errs) < 1e-4, (cfg, lengths, ["%.2e" % e for e in errs])
    enc.close()
    for dtype, tol in (("fp32x3", 1e-4), ("bf16", 6e-2)):
        enc2 = HipEncoder(cfg, weights, dtype=dtype)
        h2 = enc2.forward(dev).cpu().numpy()
        assert np.isfinite(h2).all()
        e2 = max(O.rel_err(h2[l], ref[l]) for l in range(len(ref)))
        assert e2 < tol, (dtype, cfg, lengths, e2)
        enc2.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_medium_architecture_matches_the_oracle(seed):
    """The same sweep at widths where the large-tile 16-bit GEMM, the three-wave LayerNorm rows and multi-tile attention
    engage (D = 384..768, several hundred frames per batch)."""
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import synth_wavs, synth_weights

    rng = np.random.default_rng(5000 + seed)
    cfg = _random_config(rng, medium=True)
    weights = synth_weights(cfg, seed)
    rate = cfg.downsample_rate
    B = int(rng.integers(2, 6))
    lengths = [int(rng.integers(60 * rate, 160 * rate)) for _ in range(B)]
    wavs = synth_wavs(lengths, seed + 1)
    ref = O.forward(cfg, weights, wavs, dtype=np.float32)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    for dtype, tol in (("fp32", 1e-4), ("fp32x3", 1e-4), ("bf16", 6e-2), ("fp16", 8e-3)):
        enc = HipEncoder(cfg, weights, dtype=dtype)
        hs = enc.forward(dev).cpu().numpy()
        assert hs.shape == (len(ref),) + ref[0].shape and np.isfinite(hs).all()
        err = max(O.rel_err(hs[l], ref[l]) for l in range(len(ref)))
        assert err < tol, (dtype, cfg, lengths, err)
        enc.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_forward_options_are_consistent(seed):
    """The options of s3enc_forward_ex on random architectures: a padded-to-n_max shard against the oracle, the wav2vec2
    feature selections, the featurize epilogue against the weighted sum of the states the same encoder writes, 16-bit state
    output against the rounded fp32 output."""
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import synth_wavs, synth_weights

    rng = np.random.default_rng(9000 + seed)
    cfg = _random_config(rng)
    weights = synth_weights(cfg, seed)
    rate = cfg.downsample_rate
    B = int(rng.integers(1, 5))
    lengths = [int(rng.integers(14 * rate, 40 * rate)) for _ in range(B)]
    n_max = max(lengths) + int(rng.integers(0, 6 * rate))
    wavs = synth_wavs(lengths, seed + 1)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    selection = None
    if cfg.family in ("hubert", "wav2vec2", "wavlm"):
        selection = [None, "fairseq_layers", "fairseq_layers_before_residual"][int(rng.integers(3))]
    ref = O.forward(cfg, weights, wavs, dtype=np.float32, n_max=n_max, selection=selection)
    enc = HipEncoder(cfg, weights)
    hs = enc.forward(dev, n_max=n_max, selection=selection)
    got = hs.cpu().numpy()
    assert got.shape == (len(ref),) + ref[0].shape
    assert max(O.rel_err(got[l], ref[l]) for l in range(len(ref))) < 1e-4, (cfg, lengths, n_max, selection)
    # featurize epilogue = weighted sum of exactly those states
    normalize = bool(rng.integers(2))
    w = rng.random(len(ref))
    w = w + 0.05
    if len(ref) > 1:
        w[int(rng.integers(len(ref)))] = 0.0  # an unselected layer
    w = (w / w.sum()).tolist()
    feat = enc.forward_featurized(dev, w, normalize=normalize, n_max=n_max, selection=selection).cpu().numpy()
    want = np.zeros(got[0].shape, dtype=np.float64)
    for wi, h in zip(w, got.astype(np.float64)):
        want += wi * (O.layer_norm(h, None, None) if normalize else h)
    assert O.rel_err(feat, want) < 3e-6, (cfg, selection, normalize)
    enc.close()
    # 16-bit states are the rounded fp32 states of the same 16-bit encoder
    dtype = ["bf16", "fp16"][int(rng.integers(2))]
    enc16 = HipEncoder(cfg, weights, dtype=dtype)
    a = enc16.forward(dev, n_max=n_max, selection=selection)
    b = enc16.forward(dev, n_max=n_max, selection=selection, out_dtype=dtype)
    assert torch.equal(a.to(b.dtype), b), (cfg, selection, dtype)
    f16 = enc16.forward_featurized(dev, w, normalize=normalize, n_max=n_max, selection=selection).cpu().numpy()
    want16 = np.zeros(got[0].shape, dtype=np.float64)
    for wi, h in zip(w, a.cpu().numpy().astype(np.float64)):
        want16 += wi * (O.layer_norm(h, None, None) if normalize else h)
    assert O.rel_err(f16, want16) < 3e-6, (cfg, selection, normalize, dtype)
    enc16.close()


@pytest.mark.parametrize("name,lengths", [
    ("tiny_wavlm", [640000]),                      # 40 s: T = 1999, far beyond the relative-position table's +-max_distance
    ("tiny_wavlm_large", [400000, 30000, 401]),    # one long, one short, one single-frame utterance in the same batch
    ("tiny_hubert", None),                         # 48 utterances, 400 .. 8000 samples
    ("tiny_multires3", None),
    ("tiny_distiller", None),
    ("tiny_data2vec", [200000, 123457]),
])
def test_shape_extremes_match_the_oracle(name, lengths):
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config(name)
    weights = synth_weights(cfg, 31)
    if lengths is None:
        rng = np.random.default_rng(32)
        lengths = [int(x) for x in rng.integers(400, 8000, size=48)]
    wavs = synth_wavs(lengths, 33)
    ref = O.forward(cfg, weights, wavs, dtype=np.float32)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    valid = [cfg.valid_frames(n, max(lengths)) for n in lengths]
    for dtype, tol in (("fp32", 1e-4), ("fp32x3", 2e-4), ("bf16", 6e-2)):
        enc = HipEncoder(cfg, weights, dtype=dtype)
        hs = enc.forward(dev).cpu().numpy()
        assert hs.shape == (len(ref),) + ref[0].shape and np.isfinite(hs).all()
        err = max(O.rel_err(hs[l], ref[l]) for l in range(len(ref)))
        assert err < tol, (name, dtype, err)
        if cfg.family != "multires_hubert":  # per utterance, over its valid frames only (SURVEY §8d, cfg5 rule)
            for b, v in enumerate(valid):
                e_b = max(O.rel_err(hs[l][b, :v], ref[l][b, :v]) for l in range(len(ref)))
                assert e_b < tol * 3, (name, dtype, b, v, e_b)
        enc.close()
