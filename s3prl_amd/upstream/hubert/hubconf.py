"""hub entries of the HuBERT family under the reference's names and signatures (s3prl/upstream/hubert/hubconf.py:29-156):
``hubert_custom(ckpt, legacy=False, fairseq=False, refresh=False, **kwargs)``, its aliases ``hubert_local`` /
``hubert_url``, and every released model (``hubert``, ``hubert_base``, ``hubert_large_ll60k``, ``hubert_base_robust_mgr``,
``mhubert_base_vp_en_es_fr_it3``, ``contentvec*``, ``ms_hubert``).  An ``http`` checkpoint resolves to the reference's own
cache file (``s3prl_amd.download``: ``~/.cache/s3prl/download/<sha256(url)>.<name>``, fetched when absent and a network
exists); ``fairseq=True`` reads the fairseq checkpoint layout directly (``s3prl_amd.ckpt``); ``legacy=True`` (the
reference's LegacyUpstreamExpert imports the ``fairseq`` package itself; the released names then select the ORIGINAL
fairseq file, hubert/hubconf.py:85-96) takes the same route — the file is converted, never handed to ``fairseq``."""

import os

from ...ckpt import convert_fairseq_checkpoint as _convert_fairseq_checkpoint
from ...download import urls_to_filepaths as _urls_to_filepaths
from .. import _released
from .expert import UpstreamExpert as _UpstreamExpert

_CONVERTED = "https://huggingface.co/s3prl/converted_ckpts/resolve/main/"


def hubert_custom(ckpt: str, legacy: bool = False, fairseq: bool = False, refresh: bool = False, **kwargs):
    # AssertionError like the reference entry (hubert/hubconf.py:36-41): the two loaders are mutually exclusive
    assert not (legacy and fairseq), (
        f"{__name__}: pass either legacy=True (load through the fairseq package) or fairseq=True (convert the fairseq "
        "checkpoint first), not both")
    if str(ckpt).startswith("http"):
        ckpt = _urls_to_filepaths(str(ckpt), refresh=refresh)
    if fairseq or legacy:
        # legacy=True: the reference hands the ORIGINAL fairseq file to LegacyUpstreamExpert, which needs the `fairseq`
        # package (hubert/hubconf.py, hubert/expert.py).  The same file is read here without that package: its layout is
        # exactly what fairseq=True converts, and the hidden states are the same network's.
        ckpt = _convert_fairseq_checkpoint(str(ckpt), "hubert", refresh=refresh)
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def hubert_local(*args, **kwargs):
    return hubert_custom(*args, **kwargs)


def hubert_url(*args, **kwargs):
    return hubert_custom(*args, **kwargs)


hubert = _released.alias("hubert", lambda: hubert_base, "The default model - Base (hubert/hubconf.py:77-82)")
hubert_base = _released.with_legacy(
    "hubert_base", hubert_custom, _CONVERTED + "hubert_base_ls960.pt",
    "https://dl.fbaipublicfiles.com/hubert/hubert_base_ls960.pt")
hubert_large_ll60k = _released.with_legacy(
    "hubert_large_ll60k", hubert_custom, _CONVERTED + "hubert_large_ll60k.pt",
    "https://dl.fbaipublicfiles.com/hubert/hubert_large_ll60k.pt")
hubert_base_robust_mgr = _released.with_legacy(
    "hubert_base_robust_mgr", hubert_custom, _CONVERTED + "HuBERT_base_robust_mgr_best_loss_2.7821.pt",
    "https://huggingface.co/kphuang68/HuBERT_base_robust_mgr/resolve/main/HuBERT_base_robust_mgr_best_loss_2.7821.pt")
mhubert_base_vp_en_es_fr_it3 = _released.converted_only(
    "mhubert_base_vp_en_es_fr_it3", hubert_custom, _CONVERTED + "mhubert_base_vp_en_es_fr_it3.pt")
contentvec = _released.converted_only("contentvec", hubert_custom, _CONVERTED + "contentvec_km100.pt")
contentvec_km100 = _released.converted_only("contentvec_km100", hubert_custom, _CONVERTED + "contentvec_km100.pt")
contentvec_km500 = _released.converted_only("contentvec_km500", hubert_custom, _CONVERTED + "contentvec_km500.pt")
ms_hubert = _released.converted_only("ms_hubert", hubert_custom, "https://huggingface.co/s3prl/MS-HuBERT/resolve/main/iter3.pt")
