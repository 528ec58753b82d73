"""MI355X-native stand-in for ``s3prl/upstream/wavlm/expert.py::UpstreamExpert`` (same constructor / forward /
get_downsample_rates contract; the forward runs in libs3enc's HIP kernels)."""

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "wavlm"
