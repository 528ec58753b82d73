"""MI355X-native stand-in for ``s3prl/upstream/distiller/expert.py::UpstreamExpert`` (DistilHuBERT).

Same contract: ``UpstreamExpert(ckpt, model_config=None, **kwargs)``; ``forward(wavs, no_pred=False)`` returns
``{"last_hidden_state", "hidden_states", "pad_mask", "paper"}`` with ``hidden_states = [feat_final] + layer outputs +
prediction heads`` (distiller/expert.py:43-60).  The whole forward — conv stack, ``post_extract_proj`` (no LayerNorm
in front of it, model.py:170-176), positional conv, the Transformer layers and the heads
``Linear -> GELU -> SplitLinear`` (model.py:155-161, module.py:55-90) — runs in libs3enc (family ``S3ENC_DISTILLER``)."""

from typing import List

import torch

from ..base import HipUpstreamExpert


class UpstreamExpert(HipUpstreamExpert):
    family = "distiller"

    def __init__(self, ckpt: str = None, model_config: str = None, **kwargs):
        if model_config is not None:
            raise NotImplementedError("distiller: build from the checkpoint's own Config (model_config files are not read)")
        super().__init__(ckpt, None, **kwargs)

    def forward(self, wavs: List[torch.Tensor], no_pred: bool = False):
        wav_dev = wavs[0].device
        hs = self.encode(wavs)
        if hs.device != wav_dev:
            hs = hs.to(wav_dev)
        NL, NH = self.cfg.encoder_layers, self.cfg.pred_heads
        states = [hs[i] for i in range(hs.shape[0] if not no_pred else 1 + NL)]
        # pad_mask: 1 for valid frames, conv-length rule (distiller/model.py:271-285)
        n_max = max(int(w.numel()) for w in wavs)
        T = hs.shape[2]
        valid = torch.tensor([self.cfg.valid_frames(int(w.numel()), n_max) for w in wavs], device=wav_dev)
        pad_mask = (torch.arange(T, device=wav_dev)[None, :] < valid[:, None]).to(torch.float32)
        return {"last_hidden_state": None if no_pred else states[-1], "hidden_states": states, "pad_mask": pad_mask,
                "paper": states[NL]}
