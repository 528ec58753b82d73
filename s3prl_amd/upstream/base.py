"""Shared body of the three ``UpstreamExpert`` mirrors.

Contract kept from the reference (upstream/example/expert.py:11-77, upstream/interfaces.py:100-131):

* ``UpstreamExpert(ckpt, model_config=None, **kwargs)`` — unknown kwargs (``refresh=``, ``legacy=``, …) are accepted
  and ignored;
* ``forward(wavs: List[FloatTensor (n_i,)]) -> dict`` with ``"hidden_states"`` (tuple of NL+1 ``(B, T, D)`` fp32
  tensors on the wavs' device: the input of every Transformer layer, then the encoder output), plus
  ``"last_hidden_state"`` and ``"hidden_state_{i}"`` exactly as ``UpstreamBase.__call__`` adds them.  Because the
  dict is complete, this module registers NO hooks;
* ``get_downsample_rates(key) -> 320``.

The forward is inference-only (the HIP path has no backward): asking for gradients raises.
"""

from __future__ import annotations

import os
from typing import Dict, List

import torch

from ..ckpt import load_checkpoint
from ..config import EncoderConfig
from ..encoder import HipEncoder


class HipUpstreamExpert(torch.nn.Module):
    family = "hubert"

    def __init__(self, ckpt: str = None, model_config: str = None, dtype: str = None, **kwargs):
        super().__init__()
        if ckpt is None:
            raise ValueError("a converted checkpoint path is required (no network: `*_local(ckpt=...)`)")
        self.cfg, self._weights = load_checkpoint(ckpt, self.family)
        self.dtype = dtype or os.environ.get("S3PRL_AMD_DTYPE", "fp32")
        self._encoders: Dict[int, HipEncoder] = {}
        # a buffer so that .to(device) / .cuda() of the enclosing model has something to move and report
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)

    @classmethod
    def from_weights(cls, cfg: EncoderConfig, weights, dtype: str = "fp32"):
        """Build directly from in-memory weights (benchmarks / tests)."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        assert cfg.family == cls.family
        self.cfg, self._weights, self.dtype, self._encoders = cfg, dict(weights), dtype, {}
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)
        return self

    def get_downsample_rates(self, key: str = None) -> int:
        return self.cfg.downsample_rate

    def _encoder_for(self, device: torch.device) -> HipEncoder:
        if device.type != "cuda":
            raise RuntimeError(
                "s3prl_amd runs the encoder on an MI355X only; move the waveforms to the GPU "
                "(there is deliberately no CPU fallback — use the reference s3prl expert on CPU)")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._encoders:
            self._encoders[idx] = HipEncoder(self.cfg, self._weights, dtype=self.dtype, device=idx)
        return self._encoders[idx]

    def encode(self, wavs: List[torch.Tensor], n_max: int = None) -> torch.Tensor:
        """(NL+1, B, T, D) fp32.  ``n_max``: global pad-to length for data-parallel shards."""
        if torch.is_grad_enabled() and any(w.requires_grad for w in wavs):
            raise RuntimeError("s3prl_amd upstream experts are inference-only (no backward through the HIP encoder)")
        return self._encoder_for(wavs[0].device).forward(wavs, n_max=n_max)

    def forward(self, wavs: List[torch.Tensor]):
        hs = self.encode(wavs)
        hidden_states = tuple(hs[l] for l in range(hs.shape[0]))
        result = {"hidden_states": hidden_states, "last_hidden_state": hidden_states[-1]}
        for i, h in enumerate(hidden_states):
            result[f"hidden_state_{i}"] = h
        return result
