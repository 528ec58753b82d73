"""MI355X-native stand-in for ``s3prl/upstream/multires_hubert/expert.py::UpstreamExpert``: the multi-resolution HuBERT
U-net (encoders -> conv adapters -> middle encoder -> conv adapters -> decoders) in libs3enc's HIP kernels.  The
``hidden_states`` are what the reference's hooks + ``hook_postprocess`` return (expert.py:49-101): per block its layer
inputs and its output, every tensor repeated to the finest frame rate and cut to the common length."""

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "multires_hubert"

    def _states_info(self, n: int):
        """Hook identifiers in the order the reference registers them (multires_hubert/expert.py:49-91)."""
        R = len(self.cfg.rate_pairs) + 1
        names = [f"self.model.encoders[{i}]" for i in range(R - 1)] + ["self.model.middle_encoder"] + \
                [f"self.model.decoders[{i}]" for i in range(R - 1)]
        info = []
        for name, layers in zip(names, self.cfg.block_layers):
            info += [f"{name}.layers[{j}]" for j in range(layers)] + [name]
        return tuple(info) if len(info) == n else tuple(f"state_{i}" for i in range(n))
