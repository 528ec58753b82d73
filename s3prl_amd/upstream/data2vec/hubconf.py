"""hub entries of data2vec-audio under the reference's names and signatures (s3prl/upstream/data2vec/hubconf.py:10-52):
``data2vec_custom(ckpt, refresh=False, **kw)``, its aliases, and the released ``data2vec`` (= Base), ``data2vec_base_960``,
``data2vec_large_ll60k``.  URLs resolve to the reference's cache file (``s3prl_amd.download``)."""

import os

from ...download import urls_to_filepaths as _urls_to_filepaths
from .. import _released
from .expert import UpstreamExpert as _UpstreamExpert

_CONVERTED = "https://huggingface.co/s3prl/converted_ckpts/resolve/main/"


def data2vec_custom(ckpt: str, refresh: bool = False, **kwargs):
    if str(ckpt).startswith("http"):
        ckpt = _urls_to_filepaths(str(ckpt), refresh=refresh)
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def data2vec_local(*args, **kwargs):
    return data2vec_custom(*args, **kwargs)


def data2vec_url(*args, **kwargs):
    return data2vec_custom(*args, **kwargs)


data2vec = _released.alias("data2vec", lambda: data2vec_base_960, "The default model - Base (data2vec/hubconf.py:25-30)")
data2vec_base_960 = _released.positional("data2vec_base_960", data2vec_custom, _CONVERTED + "audio_base_ls.pt")
data2vec_large_ll60k = _released.positional("data2vec_large_ll60k", data2vec_custom, _CONVERTED + "vox_pretrained.pt")
