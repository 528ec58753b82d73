"""CPU restatement (numpy) of the reference ``fbank`` upstream (BASELINE configs[0]).  TEST INFRASTRUCTURE ONLY.

Reference path: ``s3prl.hub.fbank`` -> ``baseline/hubconf.py:45-50`` -> ``baseline/expert.py:23-79`` ->
``baseline/extracter.py:32-90`` with ``baseline/fbank.yaml``:
    torchaudio.compliance.kaldi.fbank(num_mel_bins=80, frame_length=25, frame_shift=10, use_log_fbank=True)
    -> 2 x torchaudio.transforms.ComputeDeltas(win_length=5)  (extracter.py:59-76)
    -> CMVN over time, unbiased std, eps 1e-10                (extracter.py:79-90)
    -> pad_sequence over the batch                            (expert.py:74-79)

Parity status: **parity unpinned**.  The arithmetic lives in torchaudio (an un-vendored dependency,
``torchaudio >=0.8.0`` in requirements/install.txt:1; not installed here, no network) and the reference's golden
vector ``sample_hidden_states/fbank.pt`` is hosted remotely (test/test_upstream.py:25-66).  This file restates the
published algorithm of ``torchaudio.compliance.kaldi.fbank`` (Kaldi's ``compute-fbank-feats`` defaults as torchaudio
documents them: povey window, pre-emphasis 0.97, remove_dc_offset, dither 0, snip_edges, round_to_power_of_two,
low_freq 20, high_freq = Nyquist, power spectrum, natural log floored at float32 eps) and of
``torchaudio.functional.compute_deltas`` (replicate-padded regression over +-2 frames, denominator 10); it is anchored
only by the reference's call sites above and by the known-answer properties in ``tests/test_fbank_cpu.py``.
"""

from __future__ import annotations

import math
from typing import List

import numpy as np

SAMPLE_RATE = 16000
EPS32 = float(np.finfo(np.float32).eps)  # torch.finfo(torch.float).eps: the floor under the log


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def povey_window(n: int) -> np.ndarray:
    """hann(periodic=False) ** 0.85 (kaldi 'povey')."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * math.pi * k / (n - 1))) ** 0.85


def mel_banks(num_bins: int, padded: int, sample_rate: float, low_freq: float = 20.0, high_freq: float = 0.0) -> np.ndarray:
    """kaldi get_mel_banks without VTLN: (num_bins, padded/2 + 1) triangular filters, equally spaced on the mel
    scale between low_freq and high_freq (<= 0: offset from Nyquist); the Nyquist column is zero."""
    nfft = padded // 2
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    width = sample_rate / padded
    lo, hi = mel_scale(low_freq), mel_scale(high_freq)
    delta = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
    mel = mel_scale(width * np.arange(nfft, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    banks = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([banks, np.zeros((num_bins, 1))], axis=1)


def frame_params(frame_length_ms: float = 25.0, frame_shift_ms: float = 10.0):
    size = int(SAMPLE_RATE * frame_length_ms * 0.001)
    shift = int(SAMPLE_RATE * frame_shift_ms * 0.001)
    padded = 1 << (size - 1).bit_length()  # round_to_power_of_two
    return size, shift, padded


def num_frames(n: int, size: int = 400, shift: int = 160) -> int:
    """snip_edges=True: only whole windows."""
    return 0 if n < size else 1 + (n - size) // shift


def kaldi_fbank(wav: np.ndarray, num_mel_bins: int = 80, frame_length: float = 25.0, frame_shift: float = 10.0,
                preemph: float = 0.97, dtype=np.float64) -> np.ndarray:
    """torchaudio.compliance.kaldi.fbank with the reference's arguments (extracter.py:52-56): (frames, num_mel_bins)."""
    size, shift, padded = frame_params(frame_length, frame_shift)
    wav = np.asarray(wav, dtype=dtype)
    m = num_frames(len(wav), size, shift)
    if m == 0:
        return np.zeros((0, num_mel_bins), dtype=dtype)
    it = wav.itemsize
    frames = np.lib.stride_tricks.as_strided(np.ascontiguousarray(wav), shape=(m, size), strides=(shift * it, it)).copy()
    frames -= frames.mean(axis=1, keepdims=True)  # remove_dc_offset
    prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)  # replicate-pad on the left
    frames = frames - dtype(preemph) * prev  # pre-emphasis
    frames = frames * povey_window(size).astype(dtype)
    spec = np.fft.rfft(frames.astype(np.float64), n=padded, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(dtype)  # use_power
    banks = mel_banks(num_mel_bins, padded, SAMPLE_RATE).astype(dtype)
    mel = power @ banks.T
    return np.log(np.maximum(mel, dtype(EPS32))).astype(dtype)  # use_log_fbank


def compute_deltas(x_tf: np.ndarray, win_length: int = 5) -> np.ndarray:
    """torchaudio.functional.compute_deltas along time on a (frames, feat) array:
    d[t] = sum_{k=-n..n} k * x[clamp(t + k)] / (n (n+1) (2n+1) / 3), n = (win_length - 1) // 2."""
    n = (win_length - 1) // 2
    denom = n * (n + 1) * (2 * n + 1) / 3.0
    T = x_tf.shape[0]
    out = np.zeros_like(x_tf)
    for k in range(-n, n + 1):
        idx = np.clip(np.arange(T) + k, 0, T - 1)
        out += k * x_tf[idx]
    return (out / denom).astype(x_tf.dtype)


def extract(wav: np.ndarray, order: int = 2, win_length: int = 5, use_cmvn: bool = True, eps: float = 1e-10,
            dtype=np.float64) -> np.ndarray:
    """``get_extracter(fbank.yaml)`` applied to one waveform (extracter.py:32-90): (frames, 80 * (order + 1))."""
    feats = [kaldi_fbank(wav, dtype=dtype)]
    for _ in range(order):
        feats.append(compute_deltas(feats[-1], win_length))
    x = np.concatenate(feats, axis=-1)
    if use_cmvn and x.shape[0] > 0:
        std = x.std(axis=0, ddof=1, keepdims=True) if x.shape[0] > 1 else np.full((1, x.shape[1]), np.nan)
        x = (x - x.mean(axis=0, keepdims=True)) / (eps + std)
    return x.astype(dtype)


def forward(wavs: List[np.ndarray], use_cmvn: bool = True, dtype=np.float64) -> np.ndarray:
    """``baseline.expert.UpstreamExpert.forward`` (expert.py:67-79): the zero-padded (B, T_max, 240) batch."""
    feats = [extract(w, use_cmvn=use_cmvn, dtype=dtype) for w in wavs]
    T = max(f.shape[0] for f in feats)
    out = np.zeros((len(feats), T, feats[0].shape[1]), dtype=dtype)
    for b, f in enumerate(feats):
        out[b, : f.shape[0]] = f
    return out
