"""hub entries of UniSpeech-SAT under the reference's names and signatures (s3prl/upstream/unispeech_sat/hubconf.py:19-81):
``unispeech_sat_local`` / ``unispeech_sat_url`` and the released ``unispeech_sat`` (= Base+), ``unispeech_sat_base``,
``unispeech_sat_base_plus``, ``unispeech_sat_large``.  URLs resolve to the reference's cache file (``s3prl_amd.download``)."""

import os

from ...download import urls_to_filepaths as _urls_to_filepaths
from .. import _released
from .expert import UpstreamExpert as _UpstreamExpert

_CONVERTED = "https://huggingface.co/s3prl/converted_ckpts/resolve/main/"


def unispeech_sat_local(ckpt, *args, **kwargs):
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(ckpt, *args, **kwargs)


def unispeech_sat_custom(ckpt, *args, **kwargs):
    return unispeech_sat_local(ckpt, *args, **kwargs)


def unispeech_sat_url(ckpt, refresh=False, *args, **kwargs):
    if str(ckpt).startswith("http"):
        ckpt = _urls_to_filepaths(str(ckpt), refresh=refresh)
    return unispeech_sat_local(ckpt, *args, **kwargs)


unispeech_sat = _released.alias("unispeech_sat", lambda: unispeech_sat_base_plus,
                                "The default model - Base-Plus (unispeech_sat/hubconf.py:39-44)")
unispeech_sat_base = _released.positional("unispeech_sat_base", unispeech_sat_url, _CONVERTED + "unispeech_sat_base.pt")
unispeech_sat_base_plus = _released.positional("unispeech_sat_base_plus", unispeech_sat_url,
                                               _CONVERTED + "unispeech_sat_base_plus.pt")
unispeech_sat_large = _released.positional("unispeech_sat_large", unispeech_sat_url, _CONVERTED + "unispeech_sat_large.pt")
