"""hub entries in the reference's naming convention (s3prl/upstream/hubert/hubconf.py:29-75): ``hubert_custom(ckpt,
legacy=False, fairseq=False, refresh=False, **kwargs)`` and its aliases ``hubert_local`` / ``hubert_url``.  This build has
no network: ``http`` sources raise; ``fairseq=True`` reads the fairseq checkpoint layout directly (``s3prl_amd.ckpt``);
``legacy=True`` (the reference's LegacyUpstreamExpert imports the ``fairseq`` package itself) raises."""

import os

from ...ckpt import convert_fairseq_checkpoint as _convert_fairseq_checkpoint
from .expert import UpstreamExpert as _UpstreamExpert


def hubert_custom(ckpt: str, legacy: bool = False, fairseq: bool = False, refresh: bool = False, **kwargs):
    # AssertionError like the reference entry (hubert/hubconf.py:36-41): the two loaders are mutually exclusive
    assert not (legacy and fairseq), (
        f"{__name__}: pass either legacy=True (load through the fairseq package) or fairseq=True (convert the fairseq "
        "checkpoint first), not both")
    if legacy:
        raise NotImplementedError(
            "hubert: legacy=True loads the checkpoint through the `fairseq` package (LegacyUpstreamExpert), which the "
            "MI355X path does not depend on — convert the checkpoint (fairseq=True) instead")
    if str(ckpt).startswith("http"):
        raise RuntimeError(f"hubert: no network in this build, cannot fetch {ckpt} — pass a local checkpoint path")
    if fairseq:
        ckpt = _convert_fairseq_checkpoint(str(ckpt), "hubert", refresh=refresh)
    assert os.path.isfile(ckpt), ckpt
    return _UpstreamExpert(str(ckpt), **kwargs)


def hubert_local(*args, **kwargs):
    return hubert_custom(*args, **kwargs)


def hubert_url(*args, **kwargs):
    return hubert_custom(*args, **kwargs)


def hubert(refresh=False, *args, **kwargs):
    """The reference's default entry downloads a released checkpoint; here it needs ``ckpt=`` (a local file)."""
    if "ckpt" not in kwargs and not args:
        raise RuntimeError("hubert: no network in this build — pass ckpt=<converted checkpoint> (see hubert_local)")
    return hubert_custom(*args, refresh=refresh, **kwargs)
