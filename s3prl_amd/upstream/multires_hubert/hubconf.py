"""hub entries in the reference's naming convention (s3prl/upstream/multires_hubert/hubconf.py:22-95):
``multires_hubert_custom(ckpt, refresh=False, **kwargs)`` / ``multires_hubert_local``; the released-checkpoint names need
``ckpt=`` here (no network)."""

import os

from .expert import UpstreamExpert as _UpstreamExpert


def multires_hubert_custom(ckpt: str, refresh: bool = False, **kwargs):
    if str(ckpt).startswith("http"):
        raise RuntimeError(f"multires_hubert: no network in this build, cannot fetch {ckpt} — pass a local checkpoint path")
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def multires_hubert_local(*args, **kwargs):
    return multires_hubert_custom(*args, **kwargs)


def _released(name):
    def entry(refresh=False, *args, **kwargs):
        if "ckpt" not in kwargs and not args:
            raise RuntimeError(f"{name}: no network in this build — pass ckpt=<converted checkpoint> (see multires_hubert_local)")
        return multires_hubert_custom(*args, refresh=refresh, **kwargs)

    entry.__name__ = name
    return entry


multires_hubert_base = _released("multires_hubert_base")
multires_hubert_large = _released("multires_hubert_large")
multires_hubert_multilingual_base = _released("multires_hubert_multilingual_base")
multires_hubert_multilingual_large400k = _released("multires_hubert_multilingual_large400k")
multires_hubert_multilingual_large600k = _released("multires_hubert_multilingual_large600k")
