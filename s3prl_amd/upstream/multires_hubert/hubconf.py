"""hub entries of multires-HuBERT under the reference's names and signatures
(s3prl/upstream/multires_hubert/hubconf.py:29-95): ``multires_hubert_custom(ckpt, refresh=False, **kwargs)`` /
``multires_hubert_local`` and the five released names.  URLs resolve to the reference's cache file (``s3prl_amd.download``)."""

import os

from ...download import urls_to_filepaths as _urls_to_filepaths
from .. import _released
from .expert import UpstreamExpert as _UpstreamExpert

_MR = "https://huggingface.co/s3prl/mr_hubert/resolve/main/"


def multires_hubert_custom(ckpt: str, refresh: bool = False, **kwargs):
    if str(ckpt).startswith("http"):
        ckpt = _urls_to_filepaths(str(ckpt), refresh=refresh)
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def multires_hubert_local(*args, **kwargs):
    return multires_hubert_custom(*args, **kwargs)


multires_hubert_base = _released.converted_only("multires_hubert_base", multires_hubert_custom, _MR + "mrhubert_mono_base.pt", kw="kwargs")
multires_hubert_large = _released.converted_only("multires_hubert_large", multires_hubert_custom, _MR + "mrhubert_mono_large.pt", kw="kwargs")
multires_hubert_multilingual_base = _released.converted_only(
    "multires_hubert_multilingual_base", multires_hubert_custom, _MR + "multi_base.pt", kw="kwargs")
multires_hubert_multilingual_large400k = _released.converted_only(
    "multires_hubert_multilingual_large400k", multires_hubert_custom, _MR + "multi_large_400k.pt", kw="kwargs")
multires_hubert_multilingual_large600k = _released.converted_only(
    "multires_hubert_multilingual_large600k", multires_hubert_custom, _MR + "multi_large_600k.pt", kw="kwargs")
