"""MI355X-native stand-in for ``s3prl/upstream/wav2vec2/expert.py::UpstreamExpert`` (same constructor / forward /
get_downsample_rates contract; the forward runs in libs3enc's HIP kernels).

``feature_selection`` (wav2vec2/expert.py:21,36-40,81-93): ``None`` is the hook-captured list (layer inputs + encoder
output, with the dict entries ``UpstreamBase.__call__`` adds).  ``"fairseq_layers"`` returns every layer's output
(``layer_results[i][0]``: for pre-LN models the raw residual stream, including the un-normalised last layer) and
``"fairseq_layers_before_residual"`` every layer's fc2 output before the residual (``layer_results[i][2]``); both as
``{"hidden_states": [...]}`` only, like the reference.  The library exports those tensors directly
(``S3ENC_SEL_LAYER_OUT`` / ``S3ENC_SEL_FFN_OUT``)."""

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "wav2vec2"

    def __init__(self, ckpt: str = None, model_config: str = None, feature_selection: str = None, **kwargs):
        assert feature_selection is None or feature_selection in ["fairseq_layers", "fairseq_layers_before_residual"]
        super().__init__(ckpt, model_config, **kwargs)
        self.feature_selection = feature_selection

    @property
    def num_layers(self) -> int:
        sel = getattr(self, "feature_selection", None)
        return self.cfg.encoder_layers if sel else self.cfg.num_hidden_states

    def forward(self, wavs):
        sel = getattr(self, "feature_selection", None)
        if sel is None:
            return super().forward(wavs)
        return self._result(self.encode(wavs, selection=sel), wavs[0].device, full=False)
