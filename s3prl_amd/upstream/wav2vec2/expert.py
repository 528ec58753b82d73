"""MI355X-native stand-in for ``s3prl/upstream/wav2vec2/expert.py::UpstreamExpert`` (same constructor / forward /
get_downsample_rates contract; the forward runs in libs3enc's HIP kernels)."""

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "wav2vec2"
