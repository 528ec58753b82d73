"""MI355X-native stand-in for ``s3prl/upstream/hf_wav2vec2/expert.py::UpstreamExpert``: identical to the ``hf_hubert``
expert (the reference's two files differ only in the ``transformers`` class they instantiate) for ``model_type: wav2vec2``."""

from ..hf_hubert.expert import UpstreamExpert as _HfExpert


class UpstreamExpert(_HfExpert):
    model_type = "wav2vec2"
