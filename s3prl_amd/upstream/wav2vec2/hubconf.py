"""hub entries in the reference's naming convention (s3prl/upstream/wav2vec2/hubconf.py): ``<name>_local(ckpt, ...)``
and ``<name>_custom``.  The URL-named entries need the network, which this build never has: they accept ``ckpt=``
pointing at an already-converted file and otherwise raise."""

from .expert import UpstreamExpert as _UpstreamExpert


def wav2vec2_custom(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def wav2vec2_local(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def wav2vec2(ckpt: str = None, *args, **kwargs):
    if ckpt is None:
        raise RuntimeError("wav2vec2: no network in this build — pass ckpt=<converted checkpoint> (see wav2vec2_local)")
    return wav2vec2_local(ckpt, *args, **kwargs)
