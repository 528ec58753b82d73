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


@pytest.mark.parametrize("name", ["tiny_hubert", "tiny_wav2vec2_large", "tiny_wavlm_large", "tiny_distiller", "tiny_data2vec"])
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
