"""hub entries in the reference's naming (s3prl/upstream/data2vec/hubconf.py): ``data2vec_custom(ckpt, refresh=False,
**kw)`` and its aliases; the URL-backed names need ``ckpt=`` (no network in this build)."""

import os

from .expert import UpstreamExpert as _UpstreamExpert


def data2vec_custom(ckpt: str, refresh: bool = False, **kwargs):
    if str(ckpt).startswith("http"):
        raise RuntimeError(f"data2vec: no network in this build, cannot fetch {ckpt} — pass a local checkpoint path")
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def data2vec_local(*args, **kwargs):
    return data2vec_custom(*args, **kwargs)


def data2vec_url(*args, **kwargs):
    return data2vec_custom(*args, **kwargs)


def data2vec(refresh=False, *args, **kwargs):
    if "ckpt" not in kwargs and not args:
        raise RuntimeError("data2vec: no network in this build — pass ckpt=<converted checkpoint> (see data2vec_local)")
    return data2vec_custom(*args, refresh=refresh, **kwargs)
