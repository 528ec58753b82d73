#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (PyTorch CPU).

Run in the build container only (it imports ``/root/reference``, which does not exist on the GPU box):

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py tiny_hubert_pad   # one case

For every case the seeded numpy weights of ``s3prl_amd.synth.synth_weights`` are loaded into the
reference model class, saved in the reference's converted-checkpoint format, and the
reference ``UpstreamExpert(ckpt)(wavs)["hidden_states"]`` is recorded.  Tests rebuild the same weights
and waveforms from the seeds stored in the fixture, so only outputs are committed.

Nothing here is copied from the reference; it is only imported and executed.
"""

from __future__ import annotations

import dataclasses
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from s3prl_amd.synth import named_config, synth_weights, synth_wavs  # noqa: E402

REFERENCE = "/root/reference"

# name → (config, weight seed, wav seed, lengths, (t_stride, c_stride) subsampling of stored tensors,
#         waveform dc offset, waveform scale)
CASES = {
    "tiny_hubert_pad": ("tiny_hubert", 1, 11, [4000, 2345, 3111, 800], (1, 1), 0.0, 1.0),
    "tiny_hubert_eq": ("tiny_hubert", 2, 12, [3200, 3200], (1, 1), 0.1, 0.5),
    "tiny_wav2vec2_pad": ("tiny_wav2vec2", 3, 13, [4000, 2345, 3111, 800], (1, 1), 0.0, 1.0),
    "tiny_wav2vec2_eq": ("tiny_wav2vec2", 4, 14, [2999, 2999], (1, 1), 0.0, 1.0),
    "tiny_hubert_large_pad": ("tiny_hubert_large", 5, 15, [4000, 2345, 3111], (1, 1), 0.3, 2.0),
    "tiny_wav2vec2_large_pad": ("tiny_wav2vec2_large", 6, 16, [3500, 4000, 1700], (1, 1), 0.0, 1.0),
    "tiny_wavlm_pad": ("tiny_wavlm", 7, 17, [4000, 2345, 3111], (1, 1), 0.0, 1.0),
    "tiny_wavlm_large_pad": ("tiny_wavlm_large", 8, 18, [4000, 2345, 3111], (1, 1), -0.2, 0.7),
    "hubert_base_pseudo": ("hubert_base", 0, 21, [23456, 16000], (4, 8), 0.0, 1.0),
    "wav2vec2_base_pseudo": ("wav2vec2_base", 0, 22, [20000, 27123], (4, 8), 0.0, 1.0),
    "wavlm_base_plus_pseudo": ("wavlm_base_plus", 0, 23, [23456, 16000], (4, 8), 0.0, 1.0),
    "hubert_large_pseudo": ("hubert_large", 0, 24, [16000, 12000], (4, 16), 0.0, 1.0),
    "wavlm_large_pseudo": ("wavlm_large", 0, 25, [16000, 12000], (4, 16), 0.0, 1.0),
    # full-size shapes of BASELINE configs[3] / [4] (round 2): T = 499 / 749, D = 1024, H = 16, Dg = 64
    "hubert_large_10s": ("hubert_large", 0, 26, [160000, 160000], (8, 32), 0.0, 1.0),
    "wavlm_large_15s_pad": ("wavlm_large", 0, 27, [240000, 61234], (8, 32), 0.0, 1.0),
    # SURVEY §8f-3 siblings
    "tiny_wavlm_norel_pad": ("tiny_wavlm_norel", 9, 19, [4000, 2345, 3111], (1, 1), 0.0, 1.0),
    "wavlm_base_norel_pseudo": ("wavlm_base", 0, 28, [20000, 27123], (4, 8), 0.0, 1.0),
    "tiny_unispeech_sat_pad": ("tiny_wavlm", 10, 20, [4000, 2345, 3111], (1, 1), 0.0, 1.0, {"expert": "unispeech_sat"}),
    "unispeech_sat_base_pseudo": ("wavlm_base", 1, 29, [23456, 16000], (4, 8), 0.0, 1.0, {"expert": "unispeech_sat"}),
    "tiny_distiller_pad": ("tiny_distiller", 11, 30, [4000, 2345, 3111, 800], (1, 1), 0.0, 1.0),
    "distilhubert_pseudo": ("distilhubert", 0, 31, [23456, 16000], (4, 8), 0.0, 1.0),
    "tiny_data2vec_pad": ("tiny_data2vec", 12, 32, [4000, 2345, 3111], (1, 1), 0.0, 1.0),
    "data2vec_base_pseudo": ("data2vec_base", 0, 33, [23456, 16000], (4, 8), 0.0, 1.0),
    # multi-resolution HuBERT (upstream/multires_hubert): ConvAdapter U-net, post-LN and pre-LN, three resolutions, the plain
    # ConvDownsampler / ConvUpsampler variant, an odd and an even frame count, and the base shape (20 ms -> 40 ms -> 20 ms)
    "tiny_multires_pad": ("tiny_multires", 13, 34, [4000, 2345, 3111, 800], (1, 1), 0.0, 1.0),
    "tiny_multires_eq": ("tiny_multires", 14, 35, [3300, 3300], (1, 1), 0.1, 0.5),
    "tiny_multires_large_pad": ("tiny_multires_large", 15, 36, [4000, 2345, 3111], (1, 1), 0.3, 2.0),
    "tiny_multires3_pad": ("tiny_multires3", 16, 37, [4000, 2345, 3111, 1500], (1, 1), 0.0, 1.0),
    "tiny_multires_plain_pad": ("tiny_multires_plain", 17, 38, [3500, 4000, 1700], (1, 1), 0.0, 1.0),
    "multires_hubert_base_pseudo": ("multires_hubert_base", 0, 39, [23456, 16000], (4, 8), 0.0, 1.0),
    # wav2vec2 feature_selection (wav2vec2/expert.py:81-93)
    "tiny_wav2vec2_fslayers": ("tiny_wav2vec2", 3, 13, [4000, 2345, 3111, 800], (1, 1), 0.0, 1.0,
                               {"selection": "fairseq_layers"}),
    "tiny_wav2vec2_fsbefore": ("tiny_wav2vec2", 3, 13, [4000, 2345, 3111, 800], (1, 1), 0.0, 1.0,
                               {"selection": "fairseq_layers_before_residual"}),
    "tiny_wav2vec2_large_fslayers": ("tiny_wav2vec2_large", 6, 16, [3500, 4000, 1700], (1, 1), 0.0, 1.0,
                                     {"selection": "fairseq_layers"}),
    "tiny_wav2vec2_large_fsbefore": ("tiny_wav2vec2_large", 6, 16, [3500, 4000, 1700], (1, 1), 0.0, 1.0,
                                     {"selection": "fairseq_layers_before_residual"}),
    # round 4: released-checkpoint statistics (synth_weights(profile="pretrained_like"): residual-stream outlier channels,
    # large LayerNorm gains, Student-t matrices, score-shifting q / k biases, near-silent / near-constant conv0 filters, a wide
    # relative-position table) — what the reference's own regression test exercises by loading released checkpoints
    # (test/test_upstream.py:118-136).  Models without waveform normalisation get int16-scale PCM with a DC offset.
    "tiny_hubert_pl": ("tiny_hubert", 21, 51, [4000, 2345, 3111, 800], (1, 1), 60.0, 3000.0, {"profile": "pretrained_like"}),
    "tiny_wavlm_large_pl": ("tiny_wavlm_large", 22, 52, [4000, 2345, 3111], (1, 1), 0.0, 1.0, {"profile": "pretrained_like"}),
    "hubert_base_pl": ("hubert_base", 0, 53, [23456, 16000], (4, 8), 60.0, 3000.0, {"profile": "pretrained_like"}),
    "wav2vec2_base_pl": ("wav2vec2_base", 0, 54, [20000, 27123], (4, 8), -35.0, 2000.0, {"profile": "pretrained_like"}),
    "hubert_large_pl": ("hubert_large", 0, 55, [16000, 12000], (4, 16), 0.0, 1.0, {"profile": "pretrained_like"}),
    "wavlm_large_pl": ("wavlm_large", 0, 56, [16000, 12000], (4, 16), 0.0, 1.0, {"profile": "pretrained_like"}),
    "hubert_base_10s_pl": ("hubert_base", 1, 57, [160000, 123456], (8, 16), 60.0, 3000.0, {"profile": "pretrained_like"}),
    "hubert_large_10s_pl": ("hubert_large", 1, 58, [160000, 160000], (8, 32), 0.0, 1.0, {"profile": "pretrained_like"}),
    # round 5: BASELINE configs[4]'s shape on released-checkpoint statistics — the only `*_pl` fixture whose relative positions
    # leave the exact bucket region (|j - i| >= 80: the log-spaced buckets, the wide relative-position table the profile
    # plants, the T = 749 LDS window of the bias) and whose padded batch masks two thirds of the keys of one utterance
    "wavlm_large_15s_pl": ("wavlm_large", 1, 59, [240000, 61234], (8, 32), 0.0, 1.0, {"profile": "pretrained_like"}),
}


# round 6: the fp16x2 mode's "inside 1e-3" claim on a DISTRIBUTION of weight seeds (round 5: the one extra seed that was tried
# measured 1.17e-3 on WavLM-large).  Weight seeds 1-6 x five pretrained-like models on 1-2 s ragged inputs (int16-scale PCM with a
# DC offset where the model does not normalise its waveform), plus WavLM-large seeds 2 and 3 at BASELINE configs[4]'s 15 s
# ragged shape (seed 1 is `wavlm_large_15s_pl` above).  Same recipe as every other case: the reference expert is executed.
SEED_MODELS = {  # config -> (subsampling, dc, scale)
    "hubert_base": ((4, 16), 60.0, 3000.0),
    "wav2vec2_base": ((4, 16), -35.0, 2000.0),
    "hubert_large": ((4, 32), 0.0, 1.0),
    "wavlm_large": ((4, 32), 0.0, 1.0),
    "data2vec_base": ((4, 16), 0.0, 1.0),
}
for _mi, (_m, (_sub, _dc, _sc)) in enumerate(SEED_MODELS.items()):
    for _s in range(1, 7):
        _rng = np.random.default_rng(1000 * _mi + _s)
        _lens = [int(_rng.integers(16000, 32000)), int(_rng.integers(16000, 32000))]
        CASES[f"{_m}_s{_s}_pl"] = (_m, _s, 100 + 10 * _mi + _s, _lens, _sub, _dc, _sc, {"profile": "pretrained_like", "seed_sweep": True})
for _s, _short in ((2, 88888), (3, 131071)):
    CASES[f"wavlm_large_15s_s{_s}_pl"] = ("wavlm_large", _s, 160 + _s, [240000, _short], (16, 32), 0.0, 1.0,
                                           {"profile": "pretrained_like", "seed_sweep": True})


def _import_reference():
    import torch  # noqa: F401

    # s3prl/util/pseudo_data.py:15 imports torchaudio at module level; placeholder modules stand in.
    sys.path.insert(0, HERE)
    import ref_shim

    ref_shim.import_reference()


def build_reference_expert(cfg, weights, tmpdir, extras=None):
    """Instantiate the reference model for ``cfg``, load ``weights`` and return its UpstreamExpert."""
    import torch

    extras = extras or {}

    _import_reference()
    conv_str = str([tuple(t) for t in cfg.conv_layers])
    common = dict(
        extractor_mode=cfg.extractor_mode, conv_bias=cfg.conv_bias, encoder_layers=cfg.encoder_layers,
        encoder_embed_dim=cfg.encoder_embed_dim, encoder_ffn_embed_dim=cfg.encoder_ffn_embed_dim,
        encoder_attention_heads=cfg.encoder_attention_heads, layer_norm_first=cfg.layer_norm_first,
        conv_pos=cfg.conv_pos, conv_pos_groups=cfg.conv_pos_groups, conv_feature_layers=conv_str,
    )
    path = os.path.join(tmpdir, "ckpt.pt")
    torch.manual_seed(0)
    if cfg.family == "hubert":
        from s3prl.upstream.hubert.hubert_model import HubertConfig, HubertModel, HubertPretrainingConfig
        from s3prl.upstream.hubert.expert import UpstreamExpert

        mc = HubertConfig(label_rate=50.0, final_dim=32, **common)
        tc = HubertPretrainingConfig(label_rate=50.0, normalize=cfg.normalize)
        model = HubertModel(mc, tc, [["a"] * 8])
        _load(model, weights)
        torch.save({"task_cfg": dataclasses.asdict(tc), "model_cfg": dataclasses.asdict(mc),
                    "model_weight": model.state_dict(), "dictionaries_symbols": [["a"] * 8]}, path)
    elif cfg.family == "multires_hubert":
        from s3prl.upstream.multires_hubert.expert import UpstreamExpert
        from s3prl.upstream.multires_hubert.hubert_model import (MultiresHubertConfig, MultiresHubertModel,
                                                                  MultiresHubertPretrainingConfig)

        R = len(cfg.rate_pairs) + 1
        bl = cfg.block_layers
        override = bl[:R] + [bl[2 * R - 2 - i] for i in range(R - 1)]  # encoders, middle, then decoders reversed (:415-424)
        mc = MultiresHubertConfig(label_rate=50.0, label_rate_ratios=list(cfg.label_rate_ratios), final_dim=32,
                                  override_encoder_layers=str(override), conv_adapator_kernal=cfg.conv_adapter_kernel,
                                  use_plain_updownsample=cfg.use_plain_updownsample, untie_final_proj=False,
                                  **{**common, "encoder_layers": bl[0]})
        tc = MultiresHubertPretrainingConfig(label_rate=50.0, label_rate_ratios=list(cfg.label_rate_ratios), sample_rate=16000,
                                             normalize=cfg.normalize, enable_padding=False, max_keep_size=None,
                                             max_sample_size=None, min_sample_size=None, single_target=False,
                                             random_crop=True, pad_audio=False)
        symbols = [["a"] * 8] * R
        model = MultiresHubertModel(mc, tc, symbols)
        _load(model, weights)
        torch.save({"task_cfg": dataclasses.asdict(tc), "model_cfg": dataclasses.asdict(mc),
                    "model_weight": model.state_dict(), "dictionaries_symbols": symbols}, path)
    elif cfg.family == "wav2vec2" and cfg.pos_conv_depth > 1:  # data2vec-audio (upstream/data2vec)
        from s3prl.upstream.data2vec.data2vec_model import Data2VecAudioConfig, Data2VecAudioModel
        from s3prl.upstream.data2vec.expert import UpstreamExpert
        from s3prl.upstream.wav2vec2.wav2vec2_model import AudioPretrainingConfig

        mc = Data2VecAudioConfig(pos_conv_depth=cfg.pos_conv_depth, **common)
        tc = AudioPretrainingConfig(normalize=cfg.normalize)
        model = Data2VecAudioModel(mc)
        model.remove_pretraining_modules()
        _load(model, weights)
        sd = dict(model.state_dict())
        sd["_ema"] = {}  # load_converted_model deletes this key (data2vec/convert.py:49)
        torch.save({"task_cfg": dataclasses.asdict(tc), "model_cfg": dataclasses.asdict(mc), "model_weight": sd}, path)
    elif cfg.family == "wav2vec2":
        from s3prl.upstream.wav2vec2.wav2vec2_model import AudioPretrainingConfig, Wav2Vec2Config, Wav2Vec2Model
        from s3prl.upstream.wav2vec2.expert import UpstreamExpert

        mc = Wav2Vec2Config(quantize_targets=True, final_dim=32, latent_vars=8, latent_groups=2, **common)
        tc = AudioPretrainingConfig(normalize=cfg.normalize)
        model = Wav2Vec2Model(mc)
        _load(model, weights)
        torch.save({"task_cfg": dataclasses.asdict(tc), "model_cfg": dataclasses.asdict(mc),
                    "model_weight": model.state_dict()}, path)
    elif cfg.family == "distiller":
        from s3prl.upstream.distiller.expert import UpstreamExpert
        from s3prl.upstream.distiller.model import DistillerConfig, DistillerModel

        d = dict(extractor_mode=cfg.extractor_mode, extractor_conv_feature_layers=conv_str, conv_pos=cfg.conv_pos,
                 conv_pos_groups=cfg.conv_pos_groups, encoder_layers=cfg.encoder_layers,
                 encoder_embed_dim=cfg.encoder_embed_dim, encoder_ffn_embed_dim=cfg.encoder_ffn_embed_dim,
                 encoder_attention_heads=cfg.encoder_attention_heads, layer_norm_first=cfg.layer_norm_first,
                 final_dim=cfg.encoder_embed_dim, n_tasks=cfg.pred_heads,
                 pred_layer_id=[4 * (i + 1) for i in range(cfg.pred_heads)], task_emb_type="expand-last",
                 out_layer_type="expand-last")
        model = DistillerModel(DistillerConfig(d))
        _load(model, weights)
        torch.save({"Config": {"distiller": d}, "Distiller": model.state_dict()}, path)
    else:
        from s3prl.upstream.wavlm.WavLM import WavLM, WavLMConfig
        if extras.get("expert") == "unispeech_sat":
            from s3prl.upstream.unispeech_sat.expert import UpstreamExpert
        else:
            from s3prl.upstream.wavlm.expert import UpstreamExpert

        d = dict(common, normalize=cfg.normalize, relative_position_embedding=cfg.relative_position_embedding,
                 num_buckets=cfg.num_buckets, max_distance=cfg.max_distance, gru_rel_pos=cfg.gru_rel_pos)
        model = WavLM(WavLMConfig(d))
        _load(model, weights)
        torch.save({"cfg": d, "model": model.state_dict()}, path)
    kw = {"feature_selection": extras["selection"]} if extras.get("selection") else {}
    expert = UpstreamExpert(path, **kw).eval()
    return expert, path


def _load(model, weights):
    import torch

    sd = model.state_dict()
    missing = [k for k in weights if k not in sd]
    assert not missing, f"names not in the reference state_dict: {missing[:5]}"
    for k, v in weights.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)


def reference_hidden_states(cfg, weights, wavs, extras=None):
    import torch

    with tempfile.TemporaryDirectory() as tmp:
        expert, _ = build_reference_expert(cfg, weights, tmp, extras)
        with torch.no_grad():
            out = expert([torch.from_numpy(w.copy()) for w in wavs])
    return [h.numpy() for h in out["hidden_states"]], out


def make_case(name: str):
    cfg_name, wseed, xseed, lengths, (ts, cs), dc, scale = CASES[name][:7]
    extras = CASES[name][7] if len(CASES[name]) > 7 else {}
    cfg = named_config(cfg_name)
    weights = synth_weights(cfg, wseed, extras.get("profile", "synthetic"))
    wavs = synth_wavs(lengths, xseed, dc=dc, scale=scale)
    hs, out = reference_hidden_states(cfg, weights, wavs, extras)
    assert len(hs) == (cfg.encoder_layers if extras.get("selection") else cfg.num_hidden_states)
    meta = dict(config=cfg_name, weight_seed=wseed, wav_seed=xseed, lengths=lengths, t_stride=ts, c_stride=cs,
                dc=dc, scale=scale, shape=list(hs[0].shape), reference="s3prl 0.4.18 @ /root/reference, torch CPU fp32",
                n_states=len(hs), **extras)
    arrays = {f"hs{l}": np.ascontiguousarray(h[:, ::ts, ::cs]) for l, h in enumerate(hs)}
    # full-tensor norms so a subsampled fixture still pins the global scale of every layer
    arrays["norms"] = np.array([np.linalg.norm(h.astype(np.float64)) for h in hs])
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {len(hs)} x {hs[0].shape} -> {os.path.getsize(path) / 1e3:.0f} kB")


# Featurizer / S3PRLUpstream fixtures: the reference's own ``s3prl.nn.S3PRLUpstream`` + ``s3prl.nn.Featurizer``
# (nn/upstream.py:102-349) on a reference expert built from the seeded weights.
# name -> (config, weight seed, wav seed, lengths, upstream normalize, featurizer normalize, layer_selections, weight seed)
FEAT_CASES = {
    "feat_tiny_hubert": ("tiny_hubert", 1, 41, [4000, 2345, 3111, 801], False, False, None, 5),
    "feat_tiny_hubert_norm_sel": ("tiny_hubert", 1, 42, [3999, 2345, 3111], False, True, [0, 2, 3], 6),
    "feat_tiny_wavlm_large_upnorm": ("tiny_wavlm_large", 8, 43, [4000, 2345, 3111], True, False, None, 7),
    "feat_tiny_hubert_short": ("tiny_hubert", 1, 44, [640, 500], False, False, None, 8),  # < MIN_SECOND: zero-extended
}


def make_feat_case(name: str):
    import torch

    cfg_name, wseed, xseed, lengths, up_norm, f_norm, sel, fseed = FEAT_CASES[name]
    cfg = named_config(cfg_name)
    weights = synth_weights(cfg, wseed)
    wavs = synth_wavs(lengths, xseed)
    _import_reference()
    from s3prl.nn.upstream import Featurizer, S3PRLUpstream

    with tempfile.TemporaryDirectory() as tmp:
        _, path = build_reference_expert(cfg, weights, tmp)
        hub_name = {"hubert": "hubert_local", "wav2vec2": "wav2vec2_local", "wavlm": "wavlm_local"}[cfg.family]
        up = S3PRLUpstream(hub_name, path_or_url=path, normalize=up_norm).eval()
        feat = Featurizer(up, layer_selections=sel, normalize=f_norm).eval()
        fw = np.random.default_rng(fseed).standard_normal(len(feat.weights)).astype(np.float32)
        with torch.no_grad():
            feat.weights.copy_(torch.from_numpy(fw))
            n = max(lengths)
            padded = torch.zeros(len(wavs), n)
            for b, w in enumerate(wavs):
                padded[b, : len(w)] = torch.from_numpy(w)
            all_hs, all_lens = up(padded, torch.tensor(lengths))
            hs, hs_len = feat(all_hs, all_lens)
    meta = dict(config=cfg_name, weight_seed=wseed, wav_seed=xseed, lengths=lengths, upstream_normalize=up_norm,
                featurizer_normalize=f_norm, layer_selections=sel, num_layers=len(all_hs),
                reference="s3prl 0.4.18 @ /root/reference: s3prl.nn.S3PRLUpstream + s3prl.nn.Featurizer, torch CPU fp32")
    arrays = {f"hs{l}": h.numpy() for l, h in enumerate(all_hs)}
    arrays["lens"] = np.stack([x.numpy() for x in all_lens])
    arrays["feat_weights"] = fw
    arrays["feat"] = hs.numpy()
    arrays["feat_len"] = hs_len.numpy()
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {len(all_hs)} x {tuple(all_hs[0].shape)} -> feat {tuple(hs.shape)}, {os.path.getsize(path) / 1e3:.0f} kB")


# The OLD Featurizer interface — s3prl.upstream.interfaces.Featurizer (interfaces.py:134-272), the class downstream/runner.py
# instantiates — on a reference expert:  name -> (config, weight seed, wav seed, lengths, feature_selection, layer_selection,
# normalize, weights seed).  Stored: the per-utterance features forward() returns, and the states they were computed from.
LEGACY_FEAT_CASES = {
    "legacyfeat_tiny_hubert_sum": ("tiny_hubert", 1, 61, [4000, 2345, 3111], "hidden_states", None, False, 11),
    "legacyfeat_tiny_hubert_norm": ("tiny_hubert", 1, 62, [3999, 2345], "hidden_states", None, True, 12),
    "legacyfeat_tiny_hubert_layer2": ("tiny_hubert", 1, 63, [4000, 3111], "hidden_states", 2, False, 13),
    "legacyfeat_tiny_wavlm_last": ("tiny_wavlm_large", 8, 64, [4000, 2345, 3111], "last_hidden_state", None, False, 14),
    "legacyfeat_tiny_hubert_unknown_key": ("tiny_hubert", 1, 65, [4000, 1234], "no_such_key", None, False, 15),
}


def make_legacy_feat_case(name: str):
    import torch

    cfg_name, wseed, xseed, lengths, fsel, lsel, norm, fseed = LEGACY_FEAT_CASES[name]
    cfg = named_config(cfg_name)
    weights = synth_weights(cfg, wseed)
    wavs = synth_wavs(lengths, xseed)
    _import_reference()
    from s3prl.upstream.interfaces import Featurizer

    with tempfile.TemporaryDirectory() as tmp:
        expert, _ = build_reference_expert(cfg, weights, tmp)
        expert.eval()
        feat = Featurizer(expert, feature_selection=fsel, upstream_device="cpu", layer_selection=lsel, normalize=norm).eval()
        fw = None
        with torch.no_grad():
            if hasattr(feat, "weights"):
                fw = np.random.default_rng(fseed).standard_normal(len(feat.weights)).astype(np.float32)
                feat.weights.copy_(torch.from_numpy(fw))
            tw = [torch.from_numpy(w) for w in wavs]
            result = expert(tw)
            outs = feat(tw, result)
    meta = dict(config=cfg_name, weight_seed=wseed, wav_seed=xseed, lengths=lengths, feature_selection=fsel,
                resolved_selection=feat.feature_selection, layer_selection=lsel, normalize=norm, output_dim=int(feat.output_dim),
                downsample_rate=int(feat.downsample_rate), num_states=len(result["hidden_states"]),
                reference="s3prl 0.4.18 @ /root/reference: s3prl.upstream.interfaces.Featurizer on the reference expert, torch CPU fp32")
    arrays = {f"out{b}": o.numpy() for b, o in enumerate(outs)}
    for l, h in enumerate(result["hidden_states"]):
        arrays[f"hs{l}"] = h.numpy()
    if fw is not None:
        arrays["feat_weights"] = fw
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {len(outs)} utterances -> {[tuple(o.shape) for o in outs]}, {os.path.getsize(path) / 1e3:.0f} kB")


# Hugging Face checkpoints through the reference's hf_hubert / hf_wav2vec2 experts (upstream/hf_hubert/expert.py:12-41):
# transformers' HubertModel / Wav2Vec2Model + Wav2Vec2FeatureExtractor on a checkpoint directory written by tests/hf_util.py
# name -> (config, model_type, weight seed, wav seed, lengths, subsampling, dc, scale)
HF_CASES = {
    "hf_tiny_hubert_pad": ("tiny_hubert", "hubert", 21, 51, [4000, 2345, 3111, 800], (1, 1), 0.0, 1.0),
    "hf_tiny_hubert_large_pad": ("tiny_hubert_large", "hubert", 22, 52, [4000, 2345, 3111], (1, 1), 0.3, 0.05),
    "hf_tiny_wav2vec2_pad": ("tiny_wav2vec2", "wav2vec2", 23, 53, [3500, 4000, 1700], (1, 1), 0.0, 1.0),
    "hf_hubert_base_pseudo": ("hubert_base", "hubert", 0, 54, [23456, 16000], (4, 8), 0.0, 1.0),
    "hf_wav2vec2_large_pseudo": ("wav2vec2_large", "wav2vec2", 0, 55, [16000, 12000], (4, 16), 0.1, 0.02),
}


def make_hf_case(name: str):
    import torch

    cfg_name, mtype, wseed, xseed, lengths, (ts, cs), dc, scale = HF_CASES[name]
    cfg = named_config(cfg_name)
    weights = synth_weights(cfg, wseed)
    wavs = synth_wavs(lengths, xseed, dc=dc, scale=scale)
    _import_reference()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hf_util import write_hf_dir

    if mtype == "hubert":
        from s3prl.upstream.hf_hubert.expert import UpstreamExpert
    else:
        from s3prl.upstream.hf_wav2vec2.expert import UpstreamExpert
    with tempfile.TemporaryDirectory() as tmp:
        write_hf_dir(tmp, cfg, weights, mtype)
        expert = UpstreamExpert(tmp).eval()
        with torch.no_grad():
            hs = [h.numpy() for h in expert([torch.from_numpy(w.copy()) for w in wavs])["hidden_states"]]
    assert len(hs) == cfg.encoder_layers + 1
    meta = dict(config=cfg_name, hf=mtype, weight_seed=wseed, wav_seed=xseed, lengths=lengths, t_stride=ts, c_stride=cs, dc=dc,
                scale=scale, shape=list(hs[0].shape), n_states=len(hs),
                reference="s3prl 0.4.18 @ /root/reference upstream/hf_%s on transformers %s, torch CPU fp32" % (mtype, __import__("transformers").__version__))
    arrays = {f"hs{l}": np.ascontiguousarray(h[:, ::ts, ::cs]) for l, h in enumerate(hs)}
    arrays["norms"] = np.array([np.linalg.norm(h.astype(np.float64)) for h in hs])
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {len(hs)} x {hs[0].shape} -> {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    names = sys.argv[1:] or (list(CASES) + list(FEAT_CASES) + list(LEGACY_FEAT_CASES) + list(HF_CASES))
    for n in names:
        if n in LEGACY_FEAT_CASES:
            make_legacy_feat_case(n)
            continue
        make_feat_case(n) if n in FEAT_CASES else (make_hf_case(n) if n in HF_CASES else make_case(n))
