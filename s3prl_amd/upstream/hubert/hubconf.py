"""hub entries in the reference's naming convention (s3prl/upstream/hubert/hubconf.py): ``<name>_local(ckpt, ...)``
and ``<name>_custom``.  The URL-named entries need the network, which this build never has: they accept ``ckpt=``
pointing at an already-converted file and otherwise raise."""

from .expert import UpstreamExpert as _UpstreamExpert


def hubert_custom(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def hubert_local(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def hubert(ckpt: str = None, *args, **kwargs):
    if ckpt is None:
        raise RuntimeError("hubert: no network in this build — pass ckpt=<converted checkpoint> (see hubert_local)")
    return hubert_local(ckpt, *args, **kwargs)
