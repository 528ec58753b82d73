"""The consumer side of the path on the MI355X experts:

* ``S3PRLUpstream`` — the interface of ``s3prl.nn.S3PRLUpstream`` (nn/upstream.py:102-231): padded ``(wavs, wavs_len)`` in,
  ``(List[hs], List[hs_len])`` out, with the reference's frame-count contract: layer ``l`` of a batch padded to ``n`` samples
  has ``ceil(n / stride_l)`` frames (an upstream that is off by rounding is trimmed, or its last frame repeated) and utterance
  ``b`` owns the first ``ceil(len_b / stride_l)`` of them;
* ``Featurizer`` (``s3prl.nn.Featurizer``, re-exported from ``s3prl_amd.featurizer``) and ``UpstreamFeaturizer``, the two
  fused: the weighted sum runs as the encoder's epilogue, so only one ``(B, T, D)`` tensor ever leaves the library;
* ``LegacyFeaturizer`` — the OLD interface, ``s3prl.upstream.interfaces.Featurizer`` (interfaces.py:134-272), the class
  ``downstream/runner.py`` instantiates: ``feature_selection`` / ``layer_selection`` on the expert's result dict,
  ``forward(paired_wavs, paired_features) -> List[Tensor]`` of un-padded per-utterance features.

Differences from the reference, all on the cheap side: no probe forward at construction when the expert can answer from its
checkpoint config (layer count / hidden size / stride), and the per-layer ``F.layer_norm`` of ``normalize=True`` is done by the
library: as part of the fused epilogue, or (un-fused ``S3PRLUpstream``) by its row kernel through ``s3enc_op_layernorm``.
"""

from __future__ import annotations

import sys
from typing import Dict, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hub
from .featurizer import Featurizer, _WeightedSum

SAMPLE_RATE = 16000
MIN_SECOND = 0.05  # batches shorter than this are zero-extended before the encoder sees them (nn/upstream.py:18-19,183-192)
TOLERABLE_SEQLEN_DIFF = 5  # interfaces.py:17: frames an upstream may be off before LegacyFeaturizer.tolist refuses

__all__ = ["S3PRLUpstream", "Featurizer", "UpstreamFeaturizer", "LegacyFeaturizer", "randomize_weights"]


def _frames(n_samples, stride: int):
    """frames of an n-sample signal at `stride`: ceil(n / stride) — int or tensor"""
    return (n_samples + stride - 1) // stride if isinstance(n_samples, int) else torch.div(n_samples + stride - 1, stride, rounding_mode="floor")


def _fit_frames(h: torch.Tensor, frames: int) -> torch.Tensor:
    """(B, n, D) -> (B, frames, D): one gather with the frame index clamped to the last frame trims a surplus and repeats
    the last frame over a deficit.  Only rounding-level mismatches are legal (fewer than a factor of two either way)."""
    n = h.size(1)
    if n == frames:
        return h
    assert 2 * min(n, frames) > max(n, frames), f"upstream returned {n} frames where {frames} were expected"
    return h[:, torch.arange(frames, device=h.device).clamp_(max=n - 1), :]


_LN_AFFINE: Dict[tuple, tuple] = {}  # (device index, C) -> (ones, zeros): the library's row kernel takes gamma / beta


def _state_layer_norm(h: torch.Tensor) -> torch.Tensor:
    """``F.layer_norm(h, h.shape[-1:])`` of one state (nn/upstream.py:224-225).  GPU-resident fp32 constants go through the
    library's row kernel (``s3enc_op_layernorm``: one wave per row, the kernel every LayerNorm of the encoder runs on);
    anything else (CPU states of a probe forward, states that carry a graph) keeps the torch op."""
    C = h.shape[-1]
    if not (h.is_cuda and h.dtype == torch.float32 and not h.requires_grad and C % 4 == 0 and C <= 2048):
        return F.layer_norm(h, h.shape[-1:])
    from . import _lib

    x = h.contiguous()
    if x.data_ptr() % 16:
        return F.layer_norm(h, h.shape[-1:])
    key = (x.device.index, C)
    if key not in _LN_AFFINE:
        _LN_AFFINE[key] = (torch.ones(C, device=x.device), torch.zeros(C, device=x.device))
    gamma, beta = _LN_AFFINE[key]
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().s3enc_op_layernorm(_lib.DTYPES["fp32"], x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            x.numel() // C, C, 0, out.data_ptr(), None,
                                            torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "s3enc_op_layernorm")
    return out


def _split_batch(wavs: torch.Tensor, wavs_len: torch.Tensor):
    """padded (B, n) or (B, n, 1) + lengths -> (list of 1-D waveforms, lengths as encoded, lengths as given)."""
    wavs = wavs.squeeze(-1) if wavs.dim() == 3 else wavs
    given = wavs_len
    floor = int(MIN_SECOND * SAMPLE_RATE)
    short = floor - int(given.max())
    if short > 0:  # every utterance grows by the same zeros, so the relative lengths survive
        wavs = F.pad(wavs, (0, short))
        wavs_len = given + short
    return [w[: int(n)] for w, n in zip(wavs, wavs_len)], wavs_len, given


def randomize_weights(weights: Dict[str, np.ndarray], seed: Optional[int] = None) -> Dict[str, np.ndarray]:
    """What ``randomize_upstream`` does to a torch module (nn/upstream.py:27-35), on a checkpoint's tensor dict: vectors and
    scalars are re-drawn from N(mean, std) of their own values, everything with two or more dimensions from Xavier-normal
    (std = sqrt(2 / (fan_in + fan_out)), receptive field included)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, w in weights.items():
        w = np.asarray(w, dtype=np.float32)
        if w.ndim < 2:
            std = float(w.std(ddof=1)) if w.size > 1 else 0.0
            out[name] = rng.normal(float(w.mean()) if w.size else 0.0, std if np.isfinite(std) else 0.0, size=w.shape).astype(np.float32)
        else:
            field = int(np.prod(w.shape[2:])) if w.ndim > 2 else 1
            std = (2.0 / ((w.shape[0] + w.shape[1]) * field)) ** 0.5
            out[name] = rng.normal(0.0, std, size=w.shape).astype(np.float32)
    return out


class S3PRLUpstream(nn.Module):
    """``S3PRLUpstream(name, path_or_url=None, refresh=False, normalize=False, extra_conf=None, randomize=False)`` over
    ``s3prl_amd.hub``; ``forward(wavs, wavs_len) -> (all_hs, all_lens)``.  ``randomize=True`` re-draws every checkpoint
    tensor (``randomize_weights``) before the weights are packed for the GPU."""

    @classmethod
    def available_names(cls, only_registered_ckpt: bool = False) -> List[str]:
        return hub.options(only_registered_ckpt)

    def __init__(self, name: str, path_or_url: str = None, refresh: bool = False, normalize: bool = False,
                 extra_conf: dict = None, randomize: bool = False):
        super().__init__()
        conf = dict(extra_conf or {}, refresh=refresh)
        if path_or_url is not None:
            conf["ckpt"] = path_or_url
        self.upstream = getattr(hub, name)(**conf)
        if randomize:
            if not hasattr(self.upstream, "randomize_"):
                raise NotImplementedError(f"{name}: this upstream keeps no host copy of its weights to re-draw")
            self.upstream.randomize_()
        self.normalize = normalize
        self._num_layers = int(self.upstream.num_layers)
        self._hidden_sizes = list(self.upstream.hidden_sizes)
        rates = self.upstream.get_downsample_rates("hidden_states")
        self._downsample_rates = list(rates) if isinstance(rates, (tuple, list)) else [int(rates)] * self._num_layers

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
        wavs_list, encoded_len, given_len = _split_batch(wavs, wavs_len)
        states = self.upstream(wavs_list)["hidden_states"]
        assert isinstance(states, (list, tuple)) and len(states) == self.num_layers, \
            f"the upstream returned {len(states)} states, {self.num_layers} were announced"
        padded = int(encoded_len.max())
        all_hs, all_lens = [], []
        for h, stride in zip(states, self.downsample_rates):
            h_len = _frames(given_len, stride)
            h = _fit_frames(h, _frames(padded, stride))[:, : int(h_len.max()), :]
            all_hs.append(_state_layer_norm(h) if self.normalize else h)
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
        wavs_list, encoded_len, given_len = _split_batch(wavs, wavs_len)
        normalize = bool(self.upstream.normalize or self.featurizer.normalize)
        expert = self.upstream.upstream
        sel = getattr(expert, "feature_selection", None)
        h = expert.encode_featurized(wavs_list, self.layer_weights(), normalize, n_max=n_max, selection=sel)
        if h.device != wavs.device:
            h = h.to(wavs.device)
        stride = self.upstream.downsample_rates[0]
        h_len = _frames(given_len, stride)
        return _fit_frames(h, _frames(int(encoded_len.max()), stride))[:, : int(h_len.max()), :], h_len


class LegacyFeaturizer(nn.Module):
    """The interface of ``s3prl.upstream.interfaces.Featurizer`` (interfaces.py:134-272): built from an expert with a probe
    forward, picks ``feature_selection`` out of the expert's result dict (``"hidden_states"`` when the key is unknown),
    optionally one layer of it (``layer_selection``), else learns a softmax-weighted sum over the list — computed by
    ``libs3enc`` (``csrc/featurizer.hip``, forward and the layer weights' backward) when the states are on the GPU — and
    ``forward(paired_wavs, paired_features)`` returns the per-utterance features cut to ``round(len / downsample_rate)``
    frames.  ``upstream_device`` is where the probe waveform is created (the experts move CPU waveforms to the GPU)."""

    def __init__(self, upstream: nn.Module, feature_selection: str = "hidden_states", upstream_device: str = "cuda",
                 layer_selection: Optional[int] = None, normalize: bool = False, **kwargs):
        super().__init__()
        self.name = "Featurizer"
        upstream.eval()
        probe = [torch.randn(SAMPLE_RATE).to(upstream_device)]
        with torch.no_grad():
            probe_out = upstream(probe)
        if feature_selection not in probe_out:
            if "hidden_states" not in probe_out:
                print(f"[{self.name}] - Error: {feature_selection} is not a key of the upstream's result and neither is "
                      f"\"hidden_states\"; available: {list(probe_out.keys())}", file=sys.stderr)
                raise ValueError(feature_selection)
            print(f"[{self.name}] - Warning: {feature_selection} is not a key of the upstream's result; using "
                  f"\"hidden_states\"", file=sys.stderr)
            feature_selection = "hidden_states"
        self.feature_selection, self.layer_selection, self.normalize = feature_selection, layer_selection, bool(normalize)

        feature = self._select_feature(probe_out)
        if isinstance(feature, (list, tuple)):
            self.layer_num = len(feature)
            print(f"[{self.name}] - Take a list of {self.layer_num} features and weighted sum them.", file=sys.stderr)
            self.weights = nn.Parameter(torch.zeros(self.layer_num))
            feature = self._weighted_sum(list(feature), probe=True)  # (shape probe only: plain torch, like the reference)
        self.output_dim = feature.size(-1)
        if hasattr(upstream, "get_downsample_rates"):
            self.downsample_rate = upstream.get_downsample_rates(feature_selection)
        else:  # no static rate: derive it from the probe
            self.downsample_rate = round(max(len(w) for w in probe) / feature.size(1))
        print(f"[{self.name}] - The selected feature {feature_selection}'s downsample rate is {self.downsample_rate}",
              file=sys.stderr)

    def _select_feature(self, features: Dict[str, Union[torch.Tensor, list, dict]]):
        feature = features.get(self.feature_selection)
        if isinstance(feature, dict):
            feature = list(feature.values())
        if isinstance(feature, (list, tuple)):
            if len(feature) == 1:
                return feature[0]
            if isinstance(self.layer_selection, int):
                return feature[self.layer_selection]
        return feature

    def _weighted_sum(self, feature: List[torch.Tensor], probe: bool = False) -> torch.Tensor:
        assert self.layer_num == len(feature), (
            f"the upstream returned {len(feature)} states, the weights were built for {self.layer_num} (an upstream with "
            "layer drop returns a varying number of states: select one layer instead, e.g. last_hidden_state)")
        if probe:  # construction-time shape probe: on the CPU, where the freshly created weights live (as the reference does)
            feature = [f.detach().cpu() for f in feature]
        norm_weights = F.softmax(self.weights, dim=-1)
        # the library's weighted sum (and, for training, its backward for the LAYER WEIGHTS) when the states are GPU-resident
        # constants; states that carry a graph (a trainable upstream: the reference's `upstream_trainable` flow) need the
        # gradient with respect to the states too, which only the torch form below propagates
        if feature[0].is_cuda and not probe and not any(f.requires_grad for f in feature):
            return _WeightedSum.apply(norm_weights, self.normalize, *feature)
        stacked = torch.stack([f.float() for f in feature], dim=0)
        if self.normalize:
            stacked = F.layer_norm(stacked, stacked.shape[-1:])
        return torch.tensordot(norm_weights.to(device=stacked.device, dtype=stacked.dtype), stacked, dims=1)

    def tolist(self, paired_wavs: List[torch.Tensor], paired_feature: torch.Tensor) -> List[torch.Tensor]:
        assert paired_feature.dim() == 3, "(batch_size, max_seq_len, feat_dim)"
        lengths = [round(len(w) / self.downsample_rate) for w in paired_wavs]
        off = abs(paired_feature.size(1) - round(max(len(w) for w in paired_wavs) / self.downsample_rate))
        assert off < TOLERABLE_SEQLEN_DIFF, f"{off} >= {TOLERABLE_SEQLEN_DIFF}"
        return [f[:n] for f, n in zip(paired_feature, lengths)]

    def forward(self, paired_wavs: List[torch.Tensor], paired_features: Dict[str, Union[torch.Tensor, list, dict]]):
        feature = self._select_feature(paired_features)
        if isinstance(feature, (list, tuple)):
            feature = self._weighted_sum(list(feature))
        return self.tolist(paired_wavs, feature)
