"""hub entries in the reference's naming (s3prl/upstream/wavlm/hubconf.py:19-41): ``wavlm_local(ckpt, *args, **kwargs)``,
``wavlm_url(ckpt, refresh=False, ...)``.  This build has no network: URL sources raise unless they are local files."""

import os

from .expert import UpstreamExpert as _UpstreamExpert


def wavlm_local(ckpt, *args, **kwargs):
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(ckpt, *args, **kwargs)


def wavlm_custom(ckpt, *args, **kwargs):
    return wavlm_local(ckpt, *args, **kwargs)


def wavlm_url(ckpt, refresh=False, *args, **kwargs):
    if str(ckpt).startswith("http"):
        raise RuntimeError(f"wavlm: no network in this build, cannot fetch {ckpt} — pass a local checkpoint path")
    return wavlm_local(ckpt, *args, **kwargs)


def wavlm(refresh=False, *args, **kwargs):
    """The reference's default entry downloads a released checkpoint; here it needs ``ckpt=`` (a local file)."""
    if "ckpt" not in kwargs and not args:
        raise RuntimeError("wavlm: no network in this build — pass ckpt=<checkpoint> (see wavlm_local)")
    return wavlm_local(*args, **kwargs)
