"""Mirror of ``s3prl.nn.upstream`` for the MI355X path: ``S3PRLUpstream`` — padded ``(wavs, wavs_len)``
in, ``(List[hs], List[hs_len])`` out, with the reference's length matching and re-padding (nn/upstream.py:166-231) —
``Featurizer`` (re-exported from ``s3prl_amd.featurizer``) and ``UpstreamFeaturizer``, the two fused: the weighted sum
runs as the encoder's epilogue, so only one ``(B, T, D)`` tensor ever leaves the library.

Differences from the reference, all on the cheap side: no probe forward at construction (layer count / hidden size
come from the checkpoint's config), and the per-layer ``F.layer_norm`` of ``normalize=True`` is done by the library
where it can be (fused path) — the default path applies ``F.layer_norm`` with torch exactly like the reference.
"""

from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hub
from .featurizer import Featurizer

SAMPLE_RATE = 16000
MIN_SECOND = 0.05  # nn/upstream.py:18-19

__all__ = ["S3PRLUpstream", "Featurizer", "UpstreamFeaturizer"]


def _match_length(xs: torch.Tensor, target_max_len: int) -> torch.Tensor:
    """nn/upstream.py:150-164: trim, or repeat the last frame, when the frame count is off by rounding."""
    n = xs.size(1)
    if n > target_max_len:
        assert n // target_max_len == 1, f"{n}, {target_max_len}"
        xs = xs[:, :target_max_len, :]
    elif n < target_max_len:
        assert target_max_len // n == 1, f"{target_max_len}, {n}"
        xs = torch.cat((xs, xs[:, -1:, :].repeat(1, target_max_len - n, 1)), dim=1)
    return xs


def _unpad(wavs: torch.Tensor, wavs_len: torch.Tensor):
    """(padded (B, n) or (B, n, 1), lengths) -> (list of 1-D waveforms, lengths actually encoded, original lengths);
    batches shorter than MIN_SECOND are zero-extended first (nn/upstream.py:183-192)."""
    if wavs.dim() == 3:
        wavs = wavs.squeeze(-1)
    original = wavs_len
    if int(max(original)) < MIN_SECOND * SAMPLE_RATE:
        extra = int(MIN_SECOND * SAMPLE_RATE) - int(max(original))
        wavs = torch.cat((wavs, wavs.new_zeros(wavs.size(0), extra)), dim=1)
        wavs_len = wavs_len + extra
    return [w[: int(n)] for w, n in zip(wavs, wavs_len)], wavs_len, original


class S3PRLUpstream(nn.Module):
    """``S3PRLUpstream(name, path_or_url=None, refresh=False, normalize=False, extra_conf=None, randomize=False)``
    (nn/upstream.py:102-140) over ``s3prl_amd.hub``; ``forward(wavs, wavs_len) -> (all_hs, all_lens)``."""

    @classmethod
    def available_names(cls, only_registered_ckpt: bool = False) -> List[str]:
        return hub.options(only_registered_ckpt)

    def __init__(self, name: str, path_or_url: str = None, refresh: bool = False, normalize: bool = False,
                 extra_conf: dict = None, randomize: bool = False):
        super().__init__()
        if randomize:
            raise NotImplementedError("randomize=True re-initialises a torch module's parameters; the MI355X experts hold "
                                      "packed device weights — build the checkpoint with the weights you want instead")
        conf = {"refresh": refresh, **(extra_conf or {})}
        if path_or_url is not None:
            conf["ckpt"] = path_or_url
        self.upstream = getattr(hub, name)(**conf)
        self.normalize = normalize
        self._num_layers = int(self.upstream.num_layers)
        self._hidden_sizes = list(self.upstream.hidden_sizes)
        rates = self.upstream.get_downsample_rates("hidden_states")
        if isinstance(rates, int):
            self._downsample_rates = [rates] * self._num_layers
        elif isinstance(rates, (tuple, list)):
            self._downsample_rates = list(rates)
        else:
            raise ValueError

    @property
    def num_layers(self) -> int:
        return self._num_layers

    @property
    def downsample_rates(self) -> List[int]:
        return self._downsample_rates

    @property
    def hidden_sizes(self) -> List[int]:
        return self._hidden_sizes

    def forward(self, wavs: torch.FloatTensor, wavs_len: torch.LongTensor):
        wavs_list, wavs_len, original = _unpad(wavs, wavs_len)
        hidden_states = self.upstream(wavs_list)["hidden_states"]
        assert isinstance(hidden_states, (list, tuple))
        assert len(hidden_states) == self.num_layers, f"{len(hidden_states)}, {self.num_layers}"
        max_wav_len = int(max(wavs_len))
        all_hs, all_lens = [], []
        for h, stride in zip(hidden_states, self.downsample_rates):
            expected = len(range(0, max_wav_len, stride))
            h = _match_length(h, expected)
            assert h.size(1) == expected
            h_len = torch.div(original - 1, stride, rounding_mode="floor") + 1
            h = h[:, : int(max(h_len)), :]
            if self.normalize:
                h = F.layer_norm(h, h.shape[-1:])
            all_hs.append(h)
            all_lens.append(h_len)
        return all_hs, all_lens


class UpstreamFeaturizer(nn.Module):
    """``Featurizer(upstream)(*upstream(wavs, wavs_len))`` in ONE library call: the softmax-weighted sum over layers
    (optionally of layer-normed states) is accumulated by the encoder's own row kernels as each state is produced
    (``s3enc_forward_ex`` with ``featurize``), so the (NL+1, B, T, D) slab is never written and a data-parallel
    exchange moves a single (B, T, D) block.  Inference of the layer weights only (their gradient needs every state:
    use ``Featurizer`` on the slab for training).  ``forward(wavs, wavs_len) -> (hs, hs_len)``."""

    def __init__(self, upstream: S3PRLUpstream, featurizer: Featurizer):
        super().__init__()
        if len(set(upstream.downsample_rates)) != 1:
            raise AssertionError("every layer must share one stride")
        self.upstream, self.featurizer = upstream, featurizer

    def layer_weights(self) -> List[float]:
        """softmax(weights) scattered to one entry per upstream layer (0 for unselected layers)."""
        n = self.upstream.num_layers
        if n == 1:
            return [1.0]
        w = F.softmax(self.featurizer.weights.detach().float(), dim=-1).cpu().tolist()
        full = [0.0] * n
        for i, l in enumerate(self.featurizer.layer_selections):
            full[l] = w[i]
        return full

    @torch.no_grad()
    def forward(self, wavs: torch.FloatTensor, wavs_len: torch.LongTensor, n_max: Optional[int] = None):
        wavs_list, wavs_len, original = _unpad(wavs, wavs_len)
        normalize = bool(self.upstream.normalize or self.featurizer.normalize)
        expert = self.upstream.upstream
        sel = getattr(expert, "feature_selection", None)
        h = expert.encode_featurized(wavs_list, self.layer_weights(), normalize, n_max=n_max, selection=sel)
        if h.device != wavs.device:
            h = h.to(wavs.device)
        stride = self.upstream.downsample_rates[0]
        h = _match_length(h, len(range(0, int(max(wavs_len)), stride)))
        h_len = torch.div(original - 1, stride, rounding_mode="floor") + 1
        return h[:, : int(max(h_len)), :], h_len
