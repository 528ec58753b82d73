"""hub entries in the reference's naming (s3prl/upstream/unispeech_sat/hubconf.py:19-41): ``unispeech_sat_local(ckpt, *args, **kwargs)``,
``unispeech_sat_url(ckpt, refresh=False, ...)``.  This build has no network: URL sources raise unless they are local files."""

import os

from .expert import UpstreamExpert as _UpstreamExpert


def unispeech_sat_local(ckpt, *args, **kwargs):
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(ckpt, *args, **kwargs)


def unispeech_sat_custom(ckpt, *args, **kwargs):
    return unispeech_sat_local(ckpt, *args, **kwargs)


def unispeech_sat_url(ckpt, refresh=False, *args, **kwargs):
    if str(ckpt).startswith("http"):
        raise RuntimeError(f"unispeech_sat: no network in this build, cannot fetch {ckpt} — pass a local checkpoint path")
    return unispeech_sat_local(ckpt, *args, **kwargs)


def unispeech_sat(refresh=False, *args, **kwargs):
    """The reference's default entry downloads a released checkpoint; here it needs ``ckpt=`` (a local file)."""
    if "ckpt" not in kwargs and not args:
        raise RuntimeError("unispeech_sat: no network in this build — pass ckpt=<checkpoint> (see unispeech_sat_local)")
    return unispeech_sat_local(*args, **kwargs)
