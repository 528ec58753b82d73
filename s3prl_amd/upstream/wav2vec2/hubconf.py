"""hub entries in the reference's naming convention (s3prl/upstream/wav2vec2/hubconf.py:28-81): ``wav2vec2_custom(ckpt,
legacy=False, fairseq=False, refresh=False, **kwargs)`` and its aliases ``wav2vec2_local`` / ``wav2vec2_url``.  This build has
no network: ``http`` sources raise; ``fairseq=True`` reads the fairseq checkpoint layout directly (``s3prl_amd.ckpt``);
``legacy=True`` (the reference's LegacyUpstreamExpert imports the ``fairseq`` package itself) raises."""

import os

from ...ckpt import convert_fairseq_checkpoint as _convert_fairseq_checkpoint
from .expert import UpstreamExpert as _UpstreamExpert


def wav2vec2_custom(ckpt: str, legacy: bool = False, fairseq: bool = False, refresh: bool = False, **kwargs):
    # AssertionError like the reference entry (hubert/hubconf.py:36-41): the two loaders are mutually exclusive
    assert not (legacy and fairseq), (
        f"{__name__}: pass either legacy=True (load through the fairseq package) or fairseq=True (convert the fairseq "
        "checkpoint first), not both")
    if legacy:
        raise NotImplementedError(
            "wav2vec2: legacy=True loads the checkpoint through the `fairseq` package (LegacyUpstreamExpert), which the "
            "MI355X path does not depend on — convert the checkpoint (fairseq=True) instead")
    if str(ckpt).startswith("http"):
        raise RuntimeError(f"wav2vec2: no network in this build, cannot fetch {ckpt} — pass a local checkpoint path")
    if fairseq:
        ckpt = _convert_fairseq_checkpoint(str(ckpt), "wav2vec2", refresh=refresh)
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def wav2vec2_local(*args, **kwargs):
    return wav2vec2_custom(*args, **kwargs)


def wav2vec2_url(*args, **kwargs):
    return wav2vec2_custom(*args, **kwargs)


def wav2vec2(refresh=False, *args, **kwargs):
    """The reference's default entry downloads a released checkpoint; here it needs ``ckpt=`` (a local file)."""
    if "ckpt" not in kwargs and not args:
        raise RuntimeError("wav2vec2: no network in this build — pass ckpt=<converted checkpoint> (see wav2vec2_local)")
    return wav2vec2_custom(*args, refresh=refresh, **kwargs)
