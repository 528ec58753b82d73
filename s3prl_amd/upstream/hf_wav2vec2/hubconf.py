"""hub entry in the reference's naming (s3prl/upstream/hf_wav2vec2/hubconf.py): ``hf_wav2vec2_custom(ckpt)`` — ``ckpt`` is a LOCAL Hugging
Face checkpoint directory here (the reference also accepts hub ids, which need the network)."""

from .expert import UpstreamExpert as _UpstreamExpert


def hf_wav2vec2_custom(ckpt, *args, **kwargs):
    return _UpstreamExpert(ckpt, *args, **kwargs)
