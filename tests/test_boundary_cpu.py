"""Host logic of the drop-in boundary that needs no GPU: checkpoint formats, config parsing, hub names."""

import numpy as np
import pytest
import torch


def test_conv_layer_spec_parser():
    from s3prl_amd.config import parse_conv_layers

    assert parse_conv_layers("[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2") == \
        [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2
    with pytest.raises(ValueError):
        parse_conv_layers("__import__('os').system('true')")


@pytest.mark.parametrize("name", ["tiny_hubert", "tiny_wav2vec2_large", "tiny_wavlm_large", "tiny_distiller", "tiny_data2vec",
                                  "tiny_multires", "tiny_multires3", "tiny_multires_plain"])
def test_checkpoint_roundtrip(tmp_path, name):
    from s3prl_amd.ckpt import load_checkpoint, save_checkpoint
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config(name)
    w = synth_weights(cfg, 3)
    path = str(tmp_path / "c.pt")
    save_checkpoint(path, cfg, w)
    cfg2, w2 = load_checkpoint(path, cfg.family)
    assert cfg2.to_dict() == cfg.to_dict()
    assert set(w2) == set(w)
    for k in w:
        assert np.array_equal(w[k], w2[k])


def test_invalid_checkpoint_message(tmp_path):
    from s3prl_amd.ckpt import load_checkpoint

    path = str(tmp_path / "bad.pt")
    torch.save({"model_cfg": {}}, path)
    with pytest.raises(ValueError, match="required key"):
        load_checkpoint(path, "hubert")


def test_hub_entries_follow_the_reference_naming():
    import s3prl_amd.hub as hub

    names = hub.options()
    for fam in ("hubert", "wav2vec2", "wavlm", "unispeech_sat", "data2vec"):
        assert fam in names and f"{fam}_local" in names and f"{fam}_custom" in names
    assert "distiller_local" in names and "distilhubert" in names
    assert "multires_hubert_local" in names and "multires_hubert_base" in names
    assert all(not n.endswith("_local") for n in hub.options(only_registered_ckpt=True))
    for n in ("fbank", "fbank_no_cmvn", "baseline", "baseline_local"):
        assert n in names
    with pytest.raises(RuntimeError, match="no network"):
        hub.hubert()


def test_expert_refuses_cpu_tensors(tmp_path):
    """No silent CPU fallback on the product path."""
    from s3prl_amd.ckpt import save_checkpoint
    from s3prl_amd.synth import named_config, synth_weights
    import s3prl_amd.hub as hub

    cfg = named_config("tiny_hubert")
    path = str(tmp_path / "c.pt")
    save_checkpoint(path, cfg, synth_weights(cfg, 0))
    expert = hub.hubert_local(ckpt=path, refresh=True, legacy=False)  # unknown kwargs are tolerated like the reference
    assert expert.get_downsample_rates("hidden_states") == 320
    with pytest.raises(RuntimeError, match="MI355X only"):
        expert([torch.zeros(16000)])


def test_multires_geometry_and_config_parsing():
    """The U-net's frame plan (EncoderConfig.multires_plan; the C++ side mirrors it, tests/test_multires_gpu.py) against
    the shapes the reference expert returned for the golden fixtures, and MultiresHubertConfig's override rule."""
    from conftest import golden_meta
    from s3prl_amd.config import config_from_multires
    from s3prl_amd.synth import named_config

    for name in ("tiny_multires_pad", "tiny_multires_eq", "tiny_multires3_pad", "tiny_multires_plain_pad",
                 "multires_hubert_base_pseudo"):
        meta = golden_meta(name)
        cfg = named_config(meta["config"])
        assert cfg.num_output_frames(max(meta["lengths"])) == meta["shape"][1]
        assert cfg.num_hidden_states == meta["n_states"]
    cfg = named_config("multires_hubert_base")
    blocks, t_out = cfg.multires_plan(499)
    assert [(b["T"], b["factor"]) for b in blocks] == [(499, 1), (250, 2), (500, 1)] and t_out == 499
    # override_encoder_layers = [enc0, enc1, middle, dec(last), dec(first)] (hubert_model.py:415-424)
    c3 = config_from_multires(dict(label_rate_ratios=[1, 2, 1, 2], override_encoder_layers="[1, 2, 3, 4, 5]"))
    assert c3.block_layers == [1, 2, 3, 5, 4] and c3.encoder_layers == 15 and c3.num_hidden_states == 20
    with pytest.raises(ValueError, match="exactly as the Hubert model"):
        config_from_multires(dict(label_rate_ratios="None"))
    with pytest.raises(ValueError, match="divide"):
        config_from_multires(dict(label_rate_ratios=[1, 5]))


def test_released_name_is_served_from_the_reference_cache_file(tmp_path, monkeypatch):
    """``hubert_base()`` (hubert/hubconf.py:85-95) resolves its URL to ``<dir>/<sha256(url)>.<basename>`` — the file the
    reference's downloader writes (util/download.py:186-208) — and loads it when present; absent and without network the
    error names that path."""
    import hashlib

    import s3prl_amd.hub as amd
    from s3prl_amd import download
    from s3prl_amd.ckpt import save_checkpoint
    from s3prl_amd.synth import named_config, synth_weights

    monkeypatch.setattr(download, "TIMEOUT_SECS", 2.0)
    old = download.get_dir()
    download.set_dir(tmp_path / "cache")
    try:
        url = amd.hubert_base.url
        assert url.startswith("https://huggingface.co/s3prl/converted_ckpts/") and url.endswith("hubert_base_ls960.pt")
        expected = tmp_path / "cache" / f"{hashlib.sha256(url.encode()).hexdigest()}.hubert_base_ls960.pt"
        assert download.cache_path(url) == expected
        with pytest.raises(RuntimeError, match="hubert_base_ls960.pt"):   # no network here: nothing to load yet
            amd.hubert_base()
        cfg = named_config("tiny_hubert")
        save_checkpoint(str(expected), cfg, synth_weights(cfg, 0))
        expert = amd.hubert_base(refresh=False)
        assert expert.cfg.to_dict() == cfg.to_dict()
        assert amd.hubert().cfg.to_dict() == cfg.to_dict()                 # the family default = Base (hubconf.py:77-82)
        # legacy=True selects the ORIGINAL fairseq file (hubert/hubconf.py:85-96); the reference then needs the `fairseq`
        # package, here the same file goes through the fairseq-layout conversion: a stand-in of that layout at the cache path
        import torch

        lurl = amd.hubert_base.legacy_url
        assert lurl == "https://dl.fbaipublicfiles.com/hubert/hubert_base_ls960.pt"
        with pytest.raises(RuntimeError, match="hubert_base_ls960.pt"):
            amd.hubert_base(legacy=True)                                   # the legacy file is not in the cache yet
        lcfg = named_config("tiny_hubert")
        model_cfg = dict(extractor_mode="default", encoder_layers=3, encoder_embed_dim=128, encoder_ffn_embed_dim=256,
                         encoder_attention_heads=2, conv_pos=16, conv_pos_groups=4, activation_fn="gelu",
                         conv_feature_layers=str([tuple(t) for t in lcfg.conv_layers]))
        torch.save({"cfg": {"task": {"normalize": False, "label_rate": 50.0}, "model": model_cfg},
                    "model": {k: torch.from_numpy(v) for k, v in synth_weights(lcfg, 4).items()},
                    "task_state": {"dictionaries": [["a", "b"]]}}, str(download.cache_path(lurl)))
        legacy = amd.hubert_base(legacy=True)
        assert legacy.cfg.to_dict() == lcfg.to_dict()
        assert np.array_equal(legacy._weights["encoder.layers.0.fc1.weight"], synth_weights(lcfg, 4)["encoder.layers.0.fc1.weight"])
        with pytest.raises(NotImplementedError, match="conformer"):
            amd.wav2vec2_conformer_relpos()
        wcfg = named_config("tiny_wavlm")
        wurl = amd.wavlm_base_plus.url
        save_checkpoint(str(download.cache_path(wurl)), wcfg, synth_weights(wcfg, 0))
        assert amd.wavlm().cfg.to_dict() == wcfg.to_dict()                 # WavLM's default = Base+ (wavlm/hubconf.py:37-42)
    finally:
        download.set_dir(old)


def test_fine_tuning_flow_raises_at_backward_instead_of_silently_freezing():
    """The reference's `upstream_trainable` flow (downstream/runner.py:258-262,296-301): `.train()` + a forward with autograd
    on.  The expert has no parameters, so the states of such a forward carry a node whose backward raises; frozen use
    (`.eval()` or `no_grad`) returns plain constants."""
    import torch

    from s3prl_amd.upstream.base import HipUpstreamExpert
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("tiny_hubert")
    expert = HipUpstreamExpert.from_weights(cfg, synth_weights(cfg, 0))
    assert list(expert.parameters()) == [] and not expert.training   # constructed in eval mode: nothing to train
    slab = torch.zeros(4, 2, 5, 8)
    assert expert._guard_backward(slab) is slab
    expert.train()                                                    # what the runner's fine-tuning flow does
    guarded = expert._guard_backward(slab)
    assert guarded.requires_grad and torch.equal(guarded, slab)
    with pytest.raises(RuntimeError, match="inference-only"):
        (guarded.sum() * 2).backward()
    with torch.no_grad():
        assert not expert._guard_backward(slab).requires_grad
    expert.eval()
    assert expert._guard_backward(slab) is slab


def test_frozen_policy_lets_a_head_train_in_train_mode(monkeypatch):
    """A parent in `.train()` with no `no_grad` around a FROZEN upstream (s3prl.nn.S3PRLUpstream leaves its expert in train
    mode, nn/upstream.py:127): `freeze()` / S3PRL_AMD_TRAIN_MODE=detach make the states constants — the head above still gets
    its gradient — instead of a backward that raises."""
    import warnings

    import torch

    from s3prl_amd.upstream.base import HipUpstreamExpert
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("tiny_hubert")
    expert = HipUpstreamExpert.from_weights(cfg, synth_weights(cfg, 0)).train()
    slab = torch.ones(4, 2, 5, 8)
    monkeypatch.setattr(HipUpstreamExpert, "_warned_detached", False)
    monkeypatch.setenv("S3PRL_AMD_TRAIN_MODE", "detach")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        a, b = expert._guard_backward(slab), expert._guard_backward(slab)
    assert a is slab and b is slab and len(rec) == 1 and "constants" in str(rec[0].message)
    head = torch.nn.Linear(8, 3)
    head(a).sum().backward()
    assert head.weight.grad is not None and not a.requires_grad
    monkeypatch.setenv("S3PRL_AMD_TRAIN_MODE", "raise")
    assert expert._guard_backward(slab).requires_grad          # back to the default
    assert expert.freeze() is expert and expert._guard_backward(slab) is slab
    monkeypatch.setenv("S3PRL_AMD_TRAIN_MODE", "nonsense")
    expert.train_mode_policy = None
    with pytest.raises(ValueError, match="raise' or 'detach"):
        expert._guard_backward(slab)


def test_outlier_writer_scaling_touches_only_the_residual_writers():
    """tools/fp16_cliff.py's knob: rows of fc2 / out_proj (and their biases) on the profile's outlier channels, nothing else."""
    from s3prl_amd.synth import named_config, outlier_channels, scale_outlier_writers, synth_weights

    cfg = named_config("tiny_hubert_large")
    w = synth_weights(cfg, 3, "pretrained_like")
    hot = outlier_channels(cfg, 3)
    assert len(set(hot.tolist())) == 4 and hot.max() < cfg.encoder_embed_dim
    w10 = scale_outlier_writers(cfg, w, 3, 10.0)
    assert set(w10) == set(w)
    cold = np.setdiff1d(np.arange(cfg.encoder_embed_dim), hot)
    for name in w:
        if name.endswith((".fc2.weight", ".out_proj.weight", ".fc2.bias", ".out_proj.bias")):
            assert np.allclose(w10[name][hot], 10.0 * w[name][hot]) and np.array_equal(w10[name][cold], w[name][cold]), name
        else:
            assert w10[name] is w[name], name
    # the pretrained-like profile made exactly these rows loud in the first place
    fc2 = w["encoder.layers.0.fc2.weight"]
    assert np.abs(fc2[hot]).mean() > 10 * np.abs(fc2[cold]).mean()


def test_mx_second_term_tuning_key_is_a_bit_mask():
    """`gemm16_mx` (include/s3enc.h): bit 1 conv1, 2 q|k|v, 4 fc1, 8 fc2, 16 force; default 14.  (The packer and the K step are checked
    on the GPU against a float64 product: tests/test_ops_gpu.py::test_gemm_f16x2_mx_second_term.)"""
    from s3prl_amd import _lib

    lib = _lib.load()
    for mask in (0, 1, 2, 4, 8, 14, 15, 31):
        _lib.check(lib.s3enc_set_tuning(b"gemm16_mx", mask))
    assert lib.s3enc_set_tuning(b"gemm16_mx", 32) != 0
    _lib.check(lib.s3enc_set_tuning(b"gemm16_mx", 14))


def test_fp16x2_hybrids_are_decided_by_the_real_three_term_predicate(tmp_path):
    """s3enc_create switches the fp16x2 hybrids on (conv2.. / post_extract_proj / out_proj on fp32 activations through the
    three-term GEMM) iff `x3_shape_ok` — gemm_x3_eligible itself on a representative call — says the kernel takes the shape: the
    path's shapes must all pass (else the mode silently loses its parity margin), degenerate ones must not (else the forward fails)."""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "x3_shape_harness")
    lib_dir = os.path.join(root, "s3prl_amd")
    build = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(lib_dir, "csrc"), "-I", os.path.join(root, "include"),
                            os.path.join(root, "tests", "native", "x3_shape_harness.hip"), "-L", lib_dir, "-ls3enc", f"-Wl,-rpath,{lib_dir}", "-o", exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    cases = [  # (N, K, lda) -> expected
        ((512, 1536, 1024), 1), ((512, 1024, 1024), 1),          # conv2-4 (k = 3, s = 2), conv5-6 (k = 2, s = 2) at C = 512
        ((768, 512, 512), 1), ((1024, 512, 512), 1),            # post_extract_proj base / large
        ((768, 768, 768), 1), ((1024, 1024, 1024), 1),          # out_proj base / large
        ((64, 128, 128), 0),                                    # N < 128: the tiny test models keep the 16-bit path
        ((512, 1520, 1024), 0),                                 # K not a multiple of 32
        ((512, 1536, 1022), 0),                                 # A rows not 16-byte granular
        ((510, 1536, 1024), 0),                                 # N not a multiple of 4
    ]
    args = [str(v) for shape, _ in cases for v in shape]
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got = [int(x) for x in out.stdout.split()]
    assert got == [e for _, e in cases], list(zip(cases, got))


def test_mx_second_term_weight_rule_is_per_weight_not_per_batch(tmp_path):
    """gemm16_mx_weight_rule (gemm16.hip): the MX K step exists for the 192-row tile only, so a weight whose 256-row tiling needs
    fewer CU-rounds x rows at the path's reference batch (M = 15968) keeps two fp16 terms — HuBERT-base q|k|v, fc1, fc2 take the MX
    image, HuBERT-large q|k|v does, its fc1 / fc2 do not (306 vs 237 us when forced: profiles/r05_mx_second_term.md).  The decision
    is made per WEIGHT at s3enc_create, never per call: gemm16_mx_eligible (the per-call check) must say the same for one utterance
    (M = 99), a shard (M = 1996) and the full batch, or a row's bits would depend on the batch it sits in."""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "mx_rule_harness")
    lib_dir = os.path.join(root, "s3prl_amd")
    build = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(lib_dir, "csrc"), "-I", os.path.join(root, "include"),
                            os.path.join(root, "tests", "native", "mx_rule_harness.hip"), "-L", lib_dir, "-ls3enc", f"-Wl,-rpath,{lib_dir}", "-o", exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    weights = {(2304, 768): 1, (3072, 768): 1, (768, 3072): 1,          # HuBERT-base q|k|v, fc1, fc2
               (3072, 1024): 1, (4096, 1024): 0, (1024, 4096): 0,       # HuBERT-large / WavLM-large q|k|v, fc1, fc2
               (2304, 704): 0, (64, 768): 0}                            # K % 128 != 0; N < 128
    batches = (99, 1996, 15968, 23968)
    cases = [(M, N, K) for (N, K) in weights for M in batches]
    out = subprocess.run([exe] + [str(v) for c in cases for v in c], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    rows = [tuple(int(x) for x in l.split()) for l in out.stdout.strip().splitlines()]
    assert len(rows) == len(cases)
    for (M, N, K), (rule, eligible) in zip(cases, rows):
        assert rule == weights[(N, K)], (M, N, K, rule)
        # the per-call check depends on the weight's shape and alignment only — never on M
        assert eligible == (1 if (K % 128 == 0 and N >= 128) else 0), (M, N, K, eligible)
