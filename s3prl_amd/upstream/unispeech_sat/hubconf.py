"""hub entries in the reference's naming (s3prl/upstream/unispeech_sat/hubconf.py:19-70); URL-named entries need the
network, which this build never has: they accept ``ckpt=`` pointing at an already-downloaded file and otherwise raise."""

from .expert import UpstreamExpert as _UpstreamExpert


def unispeech_sat_local(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def unispeech_sat_custom(ckpt: str, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)


def unispeech_sat(ckpt: str = None, *args, **kwargs):
    if ckpt is None:
        raise RuntimeError("unispeech_sat: no network in this build — pass ckpt=<checkpoint> (see unispeech_sat_local)")
    return unispeech_sat_local(ckpt, *args, **kwargs)
