"""hub entries in the reference's naming (s3prl/upstream/distiller/hubconf.py:13-46).  No network in this build: the
URL-backed names need ``ckpt=`` pointing at a local file."""

import os

from .expert import UpstreamExpert as _UpstreamExpert


def distiller_local(ckpt, *args, **kwargs):
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(ckpt, *args, **kwargs)


def distiller_url(ckpt, refresh=False, *args, **kwargs):
    if str(ckpt).startswith("http"):
        raise RuntimeError(f"distiller: no network in this build, cannot fetch {ckpt} — pass a local checkpoint path")
    return distiller_local(ckpt, *args, **kwargs)


def distilhubert(refresh=False, *args, **kwargs):
    return distilhubert_base(refresh=refresh, *args, **kwargs)


def distilhubert_base(refresh=False, *args, **kwargs):
    if "ckpt" not in kwargs and not args:
        raise RuntimeError("distilhubert: no network in this build — pass ckpt=<checkpoint> (see distiller_local)")
    return distiller_local(*args, **kwargs)
