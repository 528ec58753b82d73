"""hub entries of the wav2vec 2.0 family under the reference's names and signatures
(s3prl/upstream/wav2vec2/hubconf.py:28-270): ``wav2vec2_custom(ckpt, legacy=False, fairseq=False, refresh=False,
**kwargs)``, ``wav2vec2_local`` / ``wav2vec2_url``, and every released model (``wav2vec2``, ``wav2vec2_base_960``,
``wav2vec2_large_960``, ``wav2vec2_large_ll60k``, ``wav2vec2_large_lv60_cv_swbd_fsh``, ``xlsr_53``, ``xls_r_300m/1b/2b``,
the VoxPopuli / S2ST transformer models).  ``http`` checkpoints resolve to the reference's cache file
(``s3prl_amd.download``).  The conformer-typed names (``layer_type="conformer"``, wav2vec2_model.py:2100,2958) are
registered and raise ``NotImplementedError``: the MI355X path builds the Transformer block only."""

import os

from ...ckpt import convert_fairseq_checkpoint as _convert_fairseq_checkpoint
from ...download import urls_to_filepaths as _urls_to_filepaths
from .. import _released
from .expert import UpstreamExpert as _UpstreamExpert

_CONVERTED = "https://huggingface.co/s3prl/converted_ckpts/resolve/main/"
_FAIRSEQ = "https://dl.fbaipublicfiles.com/fairseq/"


def wav2vec2_custom(ckpt: str, legacy: bool = False, fairseq: bool = False, refresh: bool = False, **kwargs):
    # AssertionError like the reference entry (wav2vec2/hubconf.py:35-40): the two loaders are mutually exclusive
    assert not (legacy and fairseq), (
        f"{__name__}: pass either legacy=True (load through the fairseq package) or fairseq=True (convert the fairseq "
        "checkpoint first), not both")
    if str(ckpt).startswith("http"):
        ckpt = _urls_to_filepaths(str(ckpt), refresh=refresh)
    if fairseq or legacy:
        # legacy=True: the reference hands the ORIGINAL fairseq file to LegacyUpstreamExpert, which needs the `fairseq`
        # package (wav2vec2/hubconf.py, wav2vec2/expert.py).  The same file is read here without that package: its layout is
        # exactly what fairseq=True converts, and the hidden states are the same network's.
        ckpt = _convert_fairseq_checkpoint(str(ckpt), "wav2vec2", refresh=refresh)
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def wav2vec2_local(*args, **kwargs):
    return wav2vec2_custom(*args, **kwargs)


def wav2vec2_url(*args, **kwargs):
    return wav2vec2_custom(*args, **kwargs)


wav2vec2 = _released.alias("wav2vec2", lambda: wav2vec2_base_960, "The default model - Base (wav2vec2/hubconf.py:76-81)")

# name -> (converted file under s3prl/converted_ckpts, original fairseq URL used with legacy=True)
_TRANSFORMER_MODELS = {
    "wav2vec2_base_960": ("wav2vec_small.pt", _FAIRSEQ + "wav2vec/wav2vec_small.pt"),
    "wav2vec2_large_960": ("libri960_big.pt", _FAIRSEQ + "wav2vec/libri960_big.pt"),
    "wav2vec2_large_ll60k": ("wav2vec_vox_new.pt", _FAIRSEQ + "wav2vec/wav2vec_vox_new.pt"),
    "wav2vec2_large_lv60_cv_swbd_fsh": ("w2v_large_lv_fsh_swbd_cv.pt", _FAIRSEQ + "wav2vec/w2v_large_lv_fsh_swbd_cv.pt"),
    "xlsr_53": ("xlsr_53_56k.pt", _FAIRSEQ + "wav2vec/xlsr_53_56k.pt"),
    "xls_r_300m": ("xlsr2_300m.pt", _FAIRSEQ + "wav2vec/xlsr2_300m.pt"),
    "xls_r_1b": ("xlsr2_960m_1000k.pt", _FAIRSEQ + "wav2vec/xlsr2_960m_1000k.pt"),
    "xls_r_2b": ("xlsr2_2B_1000k.pt", _FAIRSEQ + "wav2vec/xlsr2_2B_1000k.pt"),
    "wav2vec2_large_voxpopuli_100k": ("wav2vec2_large_100k.pt", "https://dl.fbaipublicfiles.com/voxpopuli/models/wav2vec2_large_100k.pt"),
    "wav2vec2_base_s2st_es_voxpopuli": ("wav2vec2_base_s2st_es_voxpopuli.pt", _FAIRSEQ + "speech_to_speech/s2st_finetuning/w2v2/es/transformer_B.pt"),
    "wav2vec2_base_s2st_en_librilight": ("wav2vec2_base_s2st_en_librilight.pt", _FAIRSEQ + "speech_to_speech/s2st_finetuning/w2v2/en/transformer_B.pt"),
}
for _name, (_file, _legacy_url) in _TRANSFORMER_MODELS.items():
    globals()[_name] = _released.with_legacy(_name, wav2vec2_custom, _CONVERTED + _file, _legacy_url)

_CONFORMER = ("conformer encoder layers (layer_type='conformer': depthwise-conv module + relative / rotary attention, "
              "wav2vec2_model.py:2100,2958) are outside the MI355X hot path, which builds the Transformer block")
for _name in ("wav2vec2_conformer_relpos", "wav2vec2_conformer_rope", "wav2vec2_conformer_large_s2st_es_voxpopuli",
              "wav2vec2_conformer_large_s2st_en_librilight"):
    globals()[_name] = _released.unsupported(_name, _CONFORMER)
del _name, _file, _legacy_url
