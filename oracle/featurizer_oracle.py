"""CPU restatement (numpy) of the consumer side of the path: ``S3PRLUpstream.forward``'s length logic and
``Featurizer._weighted_sum``.  TEST INFRASTRUCTURE ONLY (same rule as ``encoder_oracle.py``: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s checker legs may import it).

Parity status: **pinned** — ``tests/golden/feat_*.npz`` were produced by running the reference's own
``s3prl.nn.S3PRLUpstream`` + ``s3prl.nn.Featurizer`` (``tests/golden/make_golden.py::make_feat_case``);
``tests/test_featurizer_cpu.py`` checks this file against them.
"""

from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

MIN_SECOND, SAMPLE_RATE = 0.05, 16000  # nn/upstream.py:18-19
EPS = 1e-5


def layer_norm_last(x: np.ndarray) -> np.ndarray:
    """``F.layer_norm(h, h.shape[-1:])`` — no affine (nn/upstream.py:225-226,321-322)."""
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    return (xc / np.sqrt((xc * xc).mean(-1, keepdims=True) + EPS)).astype(x.dtype)


def encoded_lengths(lengths: Sequence[int]) -> List[int]:
    """Lengths actually fed to the upstream: batches shorter than MIN_SECOND are zero-extended (nn/upstream.py:183-192)."""
    m = max(lengths)
    extra = int(MIN_SECOND * SAMPLE_RATE) - m if m < MIN_SECOND * SAMPLE_RATE else 0
    return [int(n) + extra for n in lengths]


def match_length(h: np.ndarray, target: int) -> np.ndarray:
    """``_match_length`` (nn/upstream.py:150-164)."""
    n = h.shape[1]
    if n > target:
        assert n // target == 1
        return h[:, :target]
    if n < target:
        assert target // n == 1
        return np.concatenate([h, np.repeat(h[:, -1:], target - n, axis=1)], axis=1)
    return h


def upstream_outputs(hidden_states: Sequence[np.ndarray], lengths: Sequence[int], stride: int = 320,
                     normalize: bool = False):
    """``S3PRLUpstream.forward`` after the upstream call (nn/upstream.py:201-231): per layer match the frame count to
    ``len(range(0, max_len, stride))``, cut to the longest valid length, optional layer norm; lengths
    ``(len - 1) // stride + 1`` of the ORIGINAL lengths."""
    enc = encoded_lengths(lengths)
    expected = len(range(0, max(enc), stride))
    h_len = np.array([(int(n) - 1) // stride + 1 for n in lengths], dtype=np.int64)
    out = []
    for h in hidden_states:
        h = match_length(h, expected)[:, : int(h_len.max())]
        out.append(layer_norm_last(h) if normalize else h)
    return out, [h_len.copy() for _ in out]


def weighted_sum(all_hs: Sequence[np.ndarray], weights: np.ndarray, layer_selections: Optional[Sequence[int]] = None,
                 normalize: bool = False) -> np.ndarray:
    """``Featurizer._weighted_sum`` (nn/upstream.py:312-328): softmax over the raw weights of the selected layers,
    optional per-layer layer norm, sum."""
    sel = list(range(len(all_hs))) if layer_selections is None else sorted(layer_selections)
    hs = [all_hs[i] for i in sel]
    if normalize:
        hs = [layer_norm_last(h) for h in hs]
    w = np.asarray(weights, dtype=np.float64)
    w = np.exp(w - w.max())
    w = (w / w.sum()).astype(hs[0].dtype)
    out = np.zeros_like(hs[0])
    for wi, h in zip(w, hs):
        out = out + wi * h
    return out
