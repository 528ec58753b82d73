"""hub entries of WavLM under the reference's names and signatures (s3prl/upstream/wavlm/hubconf.py:19-79):
``wavlm_local(ckpt, *args, **kwargs)``, ``wavlm_url(ckpt, refresh=False, ...)`` and the released ``wavlm`` (= Base+),
``wavlm_base``, ``wavlm_base_plus``, ``wavlm_large``.  URLs resolve to the reference's cache file (``s3prl_amd.download``)."""

import os

from ...download import urls_to_filepaths as _urls_to_filepaths
from .. import _released
from .expert import UpstreamExpert as _UpstreamExpert

_CONVERTED = "https://huggingface.co/s3prl/converted_ckpts/resolve/main/"


def wavlm_local(ckpt, *args, **kwargs):
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(ckpt, *args, **kwargs)


def wavlm_custom(ckpt, *args, **kwargs):
    return wavlm_local(ckpt, *args, **kwargs)


def wavlm_url(ckpt, refresh=False, *args, **kwargs):
    if str(ckpt).startswith("http"):
        ckpt = _urls_to_filepaths(str(ckpt), refresh=refresh)
    return wavlm_local(ckpt, *args, **kwargs)


wavlm = _released.alias("wavlm", lambda: wavlm_base_plus, "The default model - Base-Plus (wavlm/hubconf.py:37-42)")
wavlm_base = _released.positional("wavlm_base", wavlm_url, _CONVERTED + "wavlm_base.pt")
wavlm_base_plus = _released.positional("wavlm_base_plus", wavlm_url, _CONVERTED + "wavlm_base_plus.pt")
wavlm_large = _released.positional("wavlm_large", wavlm_url, _CONVERTED + "wavlm_large.pt")
