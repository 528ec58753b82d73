"""MI355X-native stand-in for ``s3prl/upstream/wav2vec2/expert.py::UpstreamExpert`` (same constructor / forward /
get_downsample_rates contract; the forward runs in libs3enc's HIP kernels).

``feature_selection`` (wav2vec2/expert.py:21,36-40,81-93): ``None`` is the hook-captured list (layer inputs + encoder
output).  ``"fairseq_layers"`` returns every layer's OUTPUT; for post-LN models (wav2vec2-base) that is
``hidden_states[1:]`` of the slab the library already writes.  For pre-LN models the last layer's un-normalised output,
and for ``"fairseq_layers_before_residual"`` the pre-residual FFN outputs, are not tapped by the library — those
selections raise instead of returning something else."""

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "wav2vec2"

    def __init__(self, ckpt: str = None, model_config: str = None, feature_selection: str = None, **kwargs):
        assert feature_selection is None or feature_selection in ["fairseq_layers", "fairseq_layers_before_residual"]
        super().__init__(ckpt, model_config, **kwargs)
        self.feature_selection = feature_selection
        if feature_selection == "fairseq_layers_before_residual" or (
                feature_selection == "fairseq_layers" and self.cfg.layer_norm_first):
            raise NotImplementedError(
                f"feature_selection={feature_selection!r} needs taps the MI355X encoder does not export "
                f"(pre-residual FFN outputs / the un-normalised last layer of a pre-LN model)")

    def forward(self, wavs):
        result = super().forward(wavs)
        if getattr(self, "feature_selection", None) == "fairseq_layers":
            return {"hidden_states": list(result["hidden_states"][1:])}
        return result
