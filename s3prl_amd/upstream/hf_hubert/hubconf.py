"""hub entry in the reference's naming (s3prl/upstream/hf_hubert/hubconf.py): ``hf_hubert_custom(ckpt)`` — ``ckpt`` is a LOCAL Hugging
Face checkpoint directory here (the reference also accepts hub ids, which need the network)."""

from .expert import UpstreamExpert as _UpstreamExpert


def hf_hubert_custom(ckpt, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)
