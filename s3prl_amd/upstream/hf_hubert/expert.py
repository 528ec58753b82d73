"""MI355X-native stand-in for ``s3prl/upstream/hf_hubert/expert.py::UpstreamExpert`` (and, through a subclass, ``hf_wav2vec2``):
a Hugging Face ``HubertModel`` / ``Wav2Vec2Model`` checkpoint directory, encoded by libs3enc instead of ``transformers``.

Reference behaviour kept (hf_hubert/expert.py:12-41): ``UpstreamExpert(ckpt)``; ``forward(wavs)`` runs the checkpoint's
``Wav2Vec2FeatureExtractor`` (per-utterance zero-mean / unit-variance with eps 1e-7 when ``do_normalize``, zero padding,
attention mask ALWAYS passed) and returns ``{"hidden_states": output.hidden_states}``; ``get_downsample_rates -> 320``.
``s3prl_amd.hf.load_hf_checkpoint`` reads ``config.json`` + ``model.safetensors`` without importing ``transformers``."""

from typing import Dict

import torch

from ...encoder import HipEncoder
from ...hf import load_hf_checkpoint
from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "wav2vec2"          # the frame-mask rule HF uses for every architecture
    model_type = "hubert"

    def __init__(self, ckpt: str = None, dtype: str = None, **kwds):
        torch.nn.Module.__init__(self)
        if ckpt is None:
            raise ValueError("a local Hugging Face checkpoint directory is required (no network in this build)")
        self.cfg, self._weights, self.preprocess = load_hf_checkpoint(ckpt)
        import json
        import os

        mt = json.load(open(os.path.join(ckpt, "config.json"))).get("model_type")
        if mt != self.model_type:
            raise ValueError(f"{ckpt} is a {mt!r} checkpoint, this upstream loads {self.model_type!r}")
        self.dtype = dtype or os.environ.get("S3PRL_AMD_DTYPE", "fp32")
        self._encoders: Dict[int, HipEncoder] = {}
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)
        self.eval()

    def forward(self, wavs):
        hs = self.encode(wavs)
        if hs.device != wavs[0].device:
            hs = hs.to(wavs[0].device)
        return {"hidden_states": tuple(hs[l] for l in range(hs.shape[0]))}
