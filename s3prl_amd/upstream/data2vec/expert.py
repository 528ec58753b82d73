"""MI355X-native stand-in for ``s3prl/upstream/data2vec/expert.py::UpstreamExpert`` (data2vec-audio).

The reference builds ``Data2VecAudioModel`` on the wav2vec2 blocks (``data2vec_model.py:224-273``) with the same
checkpoint format as wav2vec2 (``data2vec/convert.py:34-56``: ``{"task_cfg", "model_cfg", "model_weight"}``), the same
forward / hook list / trim as the wav2vec2 expert (``data2vec/expert.py:18-68``) and wav2vec2's conv-length frame mask
(``data2vec_model.py:455-474``).  What differs on the hot path is the positional encoder: ``pos_conv_depth`` (5) blocks of
``Conv1d(D, D, max(3, conv_pos // depth), groups) -> SamePad -> LayerNorm(no affine) -> GELU`` instead of one
weight-normed conv (``wav2vec2_model.py:2995-3023``) — ``s3enc_config.pos_conv_depth`` in libs3enc."""

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "wav2vec2"
