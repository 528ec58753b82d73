"""The registration promise of INTEGRATION.md §1 against the REAL reference ``s3prl.hub`` / ``s3prl.nn`` (imported from
/root/reference behind placeholder torchaudio / omegaconf modules — build container only; skipped where the reference
tree is absent, e.g. on the GPU box): ``register_into_s3prl`` installs our entries without clobbering the reference's,
and the hub-entry signatures / the expert attributes are what ``S3PRLUpstream.__init__`` (nn/upstream.py:102-140) uses."""

import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")


_SCRIPT = r"""
import inspect, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "golden"))
import ref_shim
ref_shim.import_reference()
import s3prl.hub as ref_hub          # the REAL reference hub
import s3prl_amd.hub as amd
from s3prl_amd.ckpt import save_checkpoint
from s3prl_amd.synth import named_config, synth_weights

before = {n: getattr(ref_hub, n) for n in ("hubert_local", "wavlm_local", "wav2vec2_custom", "fbank")}
installed = amd.register_into_s3prl(prefix="amd_")
assert {"amd_hubert_local", "amd_distiller_local", "amd_unispeech_sat_local", "amd_wav2vec2_custom"} <= set(installed)
amd.register_into_s3prl()            # default: override=False -> same-named reference entries are left alone
for n, f in before.items():
    assert getattr(ref_hub, n) is f, n
assert "amd_hubert_local" in ref_hub.options()
cfg = named_config("tiny_hubert")
path = os.path.join(sys.argv[2], "c.pt")
save_checkpoint(path, cfg, synth_weights(cfg, 0))
expert = getattr(ref_hub, "amd_hubert_local")(ckpt=path, refresh=False)
assert expert.get_downsample_rates("hidden_states") == 320
assert expert.num_layers == cfg.encoder_layers + 1 and expert.hidden_sizes == [cfg.encoder_embed_dim] * expert.num_layers
for name in ("hubert_custom", "wav2vec2_custom"):   # hubert/hubconf.py:29-66, wav2vec2/hubconf.py:28-66
    ref, ours = inspect.signature(getattr(ref_hub, name)), inspect.signature(getattr(amd, name))
    assert list(ref.parameters) == list(ours.parameters), name
    assert all(ours.parameters[k].default is False for k in ("legacy", "fairseq", "refresh"))
print("REFERENCE_HUB_OK")
"""


def test_register_into_the_real_hub_and_signatures(tmp_path):
    """Runs in a subprocess: importing the reference installs placeholder torchaudio / omegaconf modules process-wide."""
    import subprocess

    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, "-c", _SCRIPT, root, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "REFERENCE_HUB_OK" in out.stdout, out.stderr[-3000:]


_NAMES_SCRIPT = r"""
import importlib, inspect, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "golden"))
import ref_shim
ref_shim.import_reference()
import s3prl.hub as ref_hub          # the REAL reference hub
import s3prl_amd.hub as amd
checked = 0
for family in ("hubert", "wav2vec2", "wavlm", "unispeech_sat", "data2vec", "distiller", "multires_hubert", "hf_hubert", "hf_wav2vec2"):
    ref_conf = importlib.import_module(f"s3prl.upstream.{family}.hubconf")
    for name, fn in vars(ref_conf).items():
        if name.startswith("_") or not inspect.isfunction(fn) or fn.__module__ != ref_conf.__name__:
            continue
        assert getattr(ref_hub, name) is fn, name                   # it IS a name the reference hub serves
        assert hasattr(amd, name), f"{family}: s3prl_amd.hub lacks {name}"
        ref, ours = inspect.signature(fn), inspect.signature(getattr(amd, name))
        assert [(k, v.kind, v.default) for k, v in ref.parameters.items()] == \
               [(k, v.kind, v.default) for k, v in ours.parameters.items()], (name, str(ref), str(ours))
        assert name in amd.options(), name
        checked += 1
assert checked >= 60, checked
# the released names are not "_local / _url / _custom" entries: S3PRLUpstream.available_names(only_registered_ckpt=True)
reg = set(amd.options(only_registered_ckpt=True))
assert {"hubert_base", "hubert_large_ll60k", "wav2vec2_large_ll60k", "xls_r_300m", "wavlm_large", "wavlm_base_plus",
        "unispeech_sat_large", "data2vec_large_ll60k", "contentvec"} <= reg
# cache naming rule = the reference's (util/download.py:186-208)
from s3prl.util import download as ref_dl
from s3prl_amd import download as our_dl
ref_dl.set_dir(sys.argv[2]); our_dl.set_dir(sys.argv[2])
for url in (amd.hubert_base.url, amd.wavlm_large.url, amd.xlsr_53.legacy_url):
    assert ref_dl._urls_to_filepaths(url, download=False) == our_dl.urls_to_filepaths(url, download=False), url
print("REFERENCE_NAMES_OK", checked)
"""


def test_every_released_name_of_the_reference_hubconfs_exists_with_the_same_signature(tmp_path):
    """`-u hubert_large_ll60k`, `S3PRLUpstream("wavlm_large")`: how every SUPERB recipe names an upstream."""
    import subprocess

    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, "-c", _NAMES_SCRIPT, root, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "REFERENCE_NAMES_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_custom_entry_rejects_legacy_plus_fairseq():
    import s3prl_amd.hub as amd

    with pytest.raises(AssertionError):
        amd.hubert_custom("x.pt", legacy=True, fairseq=True)  # same mutual exclusion as hubert/hubconf.py:36-41


def test_fairseq_layout_conversion(tmp_path):
    """``fairseq=True``: {"cfg": {"task", "model"}, "model"} -> the converted format, next to the source file."""
    import torch

    import s3prl_amd.hub as amd
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("tiny_hubert")
    sd = {k: torch.from_numpy(v) for k, v in synth_weights(cfg, 3).items()}
    model_cfg = dict(extractor_mode="default", encoder_layers=3, encoder_embed_dim=128, encoder_ffn_embed_dim=256,
                     encoder_attention_heads=2, conv_pos=16, conv_pos_groups=4, activation_fn="gelu",
                     conv_feature_layers=str([tuple(t) for t in cfg.conv_layers]))
    src = tmp_path / "fairseq_style.pt"
    torch.save({"cfg": {"task": {"normalize": False, "label_rate": 50.0}, "model": model_cfg}, "model": sd,
                "task_state": {"dictionaries": [["a", "b"]]}}, str(src))
    expert = amd.hubert_custom(str(src), fairseq=True)
    assert (tmp_path / "fairseq_style.converted.pt").is_file()
    assert expert.cfg.to_dict() == cfg.to_dict()


def test_fairseq_layout_conversion_of_an_omegaconf_container(tmp_path, monkeypatch):
    """Released fairseq checkpoints pickle ``cfg`` as an ``omegaconf.DictConfig`` (hubert/convert.py:22-40 reads it through
    fairseq).  ``omegaconf`` is not installed here, so a stand-in package with the two things the branch touches — a picklable
    container class and ``OmegaConf.to_container`` — exercises it: the converted checkpoint must equal the plain-dict one."""
    import sys

    import torch

    pkg = tmp_path / "fake_site" / "omegaconf"
    pkg.mkdir(parents=True)
    (pkg / "__init__.py").write_text(
        "class DictConfig:\n"
        "    def __init__(self, content):\n"
        "        self._content = {k: DictConfig(v) if isinstance(v, dict) else v for k, v in content.items()}\n"
        "    def __getitem__(self, k):\n"
        "        return self._content[k]\n"
        "class OmegaConf:\n"
        "    @staticmethod\n"
        "    def to_container(cfg):\n"
        "        return {k: OmegaConf.to_container(v) if isinstance(v, DictConfig) else v for k, v in cfg._content.items()}\n")
    monkeypatch.syspath_prepend(str(tmp_path / "fake_site"))
    sys.modules.pop("omegaconf", None)
    import omegaconf

    import s3prl_amd.hub as amd
    from s3prl_amd.synth import named_config, synth_weights

    try:
        cfg = named_config("tiny_hubert")
        sd = {k: torch.from_numpy(v) for k, v in synth_weights(cfg, 3).items()}
        model_cfg = dict(extractor_mode="default", encoder_layers=3, encoder_embed_dim=128, encoder_ffn_embed_dim=256,
                         encoder_attention_heads=2, conv_pos=16, conv_pos_groups=4, activation_fn="gelu",
                         conv_feature_layers=str([tuple(t) for t in cfg.conv_layers]))
        src = tmp_path / "fairseq_omegaconf.pt"
        torch.save({"cfg": omegaconf.DictConfig({"task": {"normalize": False, "label_rate": 50.0}, "model": model_cfg}), "model": sd,
                    "task_state": {"dictionaries": [["a", "b"]]}}, str(src))
        expert = amd.hubert_custom(str(src), fairseq=True)
        assert expert.cfg.to_dict() == cfg.to_dict()
        conv = torch.load(str(tmp_path / "fairseq_omegaconf.converted.pt"), map_location="cpu", weights_only=False)
        assert isinstance(conv["model_cfg"], dict) and conv["model_cfg"]["encoder_layers"] == 3
        assert conv["task_cfg"] == {"normalize": False, "label_rate": 50.0} and conv["dictionaries_symbols"] == [["a", "b"]]
    finally:
        sys.modules.pop("omegaconf", None)
