"""hub entries in the reference's naming convention (s3prl/upstream/wavlm/hubconf.py): ``<name>_local(ckpt, ...)``
and ``<name>_custom``.  The URL-named entries need the network, which this build never has: they accept ``ckpt=``
pointing at an already-converted file and otherwise raise."""

from .expert import UpstreamExpert as _UpstreamExpert


def wavlm_custom(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def wavlm_local(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def wavlm(ckpt: str = None, *args, **kwargs):
    if ckpt is None:
        raise RuntimeError("wavlm: no network in this build — pass ckpt=<converted checkpoint> (see wavlm_local)")
    return wavlm_local(ckpt, *args, **kwargs)
