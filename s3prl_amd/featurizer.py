"""``Featurizer``: the direct consumer of the upstream's ``hidden_states``, with the weighted sum over
layers running in ``libs3enc.so`` (``csrc/featurizer.hip``).

Mirror of ``s3prl.nn.Featurizer`` (s3prl/nn/upstream.py:234-349): same constructor
(``Featurizer(upstream, layer_selections=None, normalize=False)`` — ``upstream`` only needs ``num_layers``,
``hidden_sizes`` and ``downsample_rates``), same trainable ``weights`` parameter (zeros), same
``forward(all_hs, all_lens) -> (hs, hs_len)``.  The upstream is frozen on this path (inference-only HIP encoder), so
the only gradient is the one of the layer weights: the HIP backward kernel returns d out / d softmax(w) and torch
differentiates the softmax.  When the layers are views of one slab (what our experts return) they are read in place;
otherwise they are stacked first.  CUDA tensors only — there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def _as_slab(all_hs: List[torch.Tensor]):
    """(base tensor, layer stride in elements) if the layers are equally spaced row-major views of one allocation
    (what our experts return), else a row-major stack — the reference experts' hooks return transposed views of
    (T, B, D) memory (``input[0].transpose(0, 1)``, hubert/expert.py:39), which are dense but NOT row-major."""
    h0 = all_hs[0]
    if len(all_hs) > 1 and all(h.is_contiguous() and h.shape == h0.shape and h.dtype == torch.float32 for h in all_hs):
        step = (all_hs[1].data_ptr() - h0.data_ptr()) // 4
        if step >= h0.numel() and step % 4 == 0 and h0.data_ptr() % 16 == 0 and \
                all((h.data_ptr() - h0.data_ptr()) == 4 * step * i for i, h in enumerate(all_hs)):
            return h0, step, None
    stacked = torch.stack([h.float() for h in all_hs], dim=0).contiguous()  # (L, B, T, D) row-major
    return stacked[0], stacked[0].numel(), stacked


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, norm_weights: torch.Tensor, normalize: bool, *all_hs: torch.Tensor):
        lib = _lib.load()
        h0 = all_hs[0]
        if not h0.is_cuda:
            raise RuntimeError("s3prl_amd.Featurizer runs on an MI355X only (no CPU fallback)")
        base, step, keep = _as_slab(list(all_hs))
        L, D = len(all_hs), h0.shape[-1]
        rows = h0.numel() // D
        out = torch.empty(h0.shape, dtype=torch.float32, device=h0.device)  # row-major, whatever h0's strides are
        w = norm_weights.detach().float().cpu().contiguous()
        wp = (C.c_float * L)(*w.tolist())
        with torch.cuda.device(h0.device):
            stream = torch.cuda.current_stream(h0.device).cuda_stream
            _lib.check(lib.s3enc_weighted_sum(C.c_void_p(base.data_ptr()), step, L, wp, int(normalize), rows, D,
                                              C.c_void_p(out.data_ptr()), C.c_void_p(stream)), "s3enc_weighted_sum")
        ctx.normalize, ctx.step, ctx.L, ctx.D, ctx.rows = bool(normalize), step, L, D, rows
        ctx.base, ctx.keep = base, keep  # keeps the slab (or the stacked copy) alive for backward
        ctx.wdev, ctx.wdtype = norm_weights.device, norm_weights.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        lib = _lib.load()
        g = grad_out.contiguous().float()
        gw = torch.empty(ctx.L, dtype=torch.float32, device=g.device)
        # reduction partials come from torch's caching allocator: stream-ordered reuse, no hipMalloc / sync per step
        scratch = torch.empty(int(lib.s3enc_weighted_sum_backward_scratch(ctx.rows, ctx.L)), dtype=torch.float64, device=g.device)
        with torch.cuda.device(g.device):
            stream = torch.cuda.current_stream(g.device).cuda_stream
            _lib.check(lib.s3enc_weighted_sum_backward(C.c_void_p(ctx.base.data_ptr()), ctx.step, ctx.L, int(ctx.normalize),
                                                       ctx.rows, ctx.D, C.c_void_p(g.data_ptr()), C.c_void_p(gw.data_ptr()),
                                                       C.c_void_p(scratch.data_ptr()), C.c_void_p(stream)),
                       "s3enc_weighted_sum_backward")
        return (gw.to(device=ctx.wdev, dtype=ctx.wdtype), None) + (None,) * ctx.L  # frozen upstream: no grad to the layers


class Featurizer(nn.Module):
    """See the module docstring; argument meaning as in the reference (nn/upstream.py:243-252)."""

    def __init__(self, upstream, layer_selections: Optional[List[int]] = None, normalize: bool = False):
        super().__init__()
        sizes, rates = set(upstream.hidden_sizes), set(upstream.downsample_rates)
        if len(sizes) != 1 or len(rates) != 1:
            raise AssertionError("every layer must share one hidden size and one stride")
        (self._output_size,), (self._downsample_rate,) = sizes, rates
        self.normalize = bool(normalize)
        n_layers = int(upstream.num_layers)
        if n_layers > 1:  # a single layer is passed through untouched and needs no weights
            chosen = range(n_layers) if layer_selections is None else layer_selections
            assert len(chosen) <= n_layers
            self.layer_selections = sorted(chosen)
            self.weights = nn.Parameter(torch.zeros(len(self.layer_selections)))

    @property
    def output_size(self) -> int:
        """hidden size of the weighted-sum output"""
        return self._output_size

    @property
    def downsample_rate(self) -> int:
        """stride (in 16 kHz samples) of the weighted-sum output"""
        return self._downsample_rate

    def forward(self, all_hs: List[torch.Tensor], all_lens: List[torch.Tensor]):
        """``all_hs``: per-layer (B, T, D) tensors, ``all_lens``: per-layer (B,) lengths -> ((B, T, D), (B,))."""
        if len(all_hs) == 1:
            return all_hs[0], all_lens[0]
        keep = set(self.layer_selections)
        picked_hs = [h for i, h in enumerate(all_hs) if i in keep]
        picked_lens = [n for i, n in enumerate(all_lens) if i in keep]
        assert len(picked_hs) == len(picked_lens) > 1
        norm_weights = F.softmax(self.weights, dim=-1)
        return _WeightedSum.apply(norm_weights, self.normalize, *picked_hs), picked_lens[0]
