"""hub entries of DistilHuBERT under the reference's names and signatures (s3prl/upstream/distiller/hubconf.py:13-46):
``distiller_local`` / ``distiller_url`` and the released ``distilhubert`` / ``distilhubert_base``.  URLs resolve to the
reference's cache file (``s3prl_amd.download``)."""

import os

from ...download import urls_to_filepaths as _urls_to_filepaths
from .. import _released
from .expert import UpstreamExpert as _UpstreamExpert


def distiller_local(ckpt, *args, **kwargs):
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(ckpt, *args, **kwargs)


def distiller_url(ckpt, refresh=False, *args, **kwargs):
    if str(ckpt).startswith("http"):
        ckpt = _urls_to_filepaths(str(ckpt), refresh=refresh)
    return distiller_local(ckpt, *args, **kwargs)


distilhubert = _released.alias("distilhubert", lambda: distilhubert_base, "DistilHuBERT (distiller/hubconf.py:31-35)")
distilhubert_base = _released.positional(
    "distilhubert_base", distiller_url,
    "https://huggingface.co/leo19941227/distilhubert/resolve/main/distilhubert_ls960_4-8-12.ckpt")
