"""MI355X-native stand-in for ``s3prl/upstream/unispeech_sat/expert.py::UpstreamExpert``.

The reference builds UniSpeech-SAT on the SAME ``WavLM`` / ``WavLMConfig`` classes and the same ``{"cfg", "model"}``
checkpoint format as its WavLM expert (unispeech_sat/expert.py:21,36-39) with an identical forward
(unispeech_sat/expert.py:70-87 vs wavlm/expert.py:71-87), so the WavLM path of libs3enc serves it unchanged."""

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "wavlm"
