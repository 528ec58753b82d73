"""CPU checks of the fbank restatement (oracle/fbank_oracle.py; BASELINE configs[0]).

torchaudio is not installed and the reference's golden vector is remote, so the oracle is "parity unpinned" against the
reference itself (see its header).  It is anchored here by
  * an INDEPENDENT implementation of the same published algorithm: ``transformers.audio_utils`` (the numpy fallback
    Hugging Face ships for ``torchaudio.compliance.kaldi.fbank`` in its Speech2Text / AST feature extractors),
  * known-answer properties of each stage (a pure tone peaks in the mel bin that contains it, the delta of a ramp is
    its slope, CMVN output has zero mean / unit unbiased variance, frame count and downsample rate of the reference's
    expert: upstream/baseline/expert.py:35-37, test/test_upstream.py shapes)."""

import numpy as np
import pytest

from oracle import fbank_oracle as F


def test_kaldi_fbank_matches_independent_transformers_implementation():
    au = pytest.importorskip("transformers.audio_utils")
    rng = np.random.default_rng(0)
    for n in (32000, 16000 + 137, 400):
        wav = rng.standard_normal(n)
        mel_filters = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000,
                                         sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
        window = au.window_function(400, "povey", periodic=False)
        ref = au.spectrogram(wav, window, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False,
                             preemphasis=0.97, mel_filters=mel_filters, log_mel="log", mel_floor=1.192092955078125e-07,
                             remove_dc_offset=True).T
        got = F.kaldi_fbank(wav)
        assert got.shape == ref.shape == (F.num_frames(n), 80)
        assert np.abs(got - ref).max() < 1e-5


def test_frame_count_and_stride():
    assert F.frame_params() == (400, 160, 512)
    assert F.num_frames(32000) == 198 and F.num_frames(400) == 1 and F.num_frames(399) == 0
    assert F.num_frames(160000) == 998  # ~ n / 160: the 10 ms stride get_downsample_rates reports


def test_pure_tone_peaks_in_its_mel_bin():
    t = np.arange(16000) / 16000.0
    for f0 in (300.0, 1000.0, 3000.0):
        fb = F.kaldi_fbank(np.sin(2 * np.pi * f0 * t))
        lo, hi = F.mel_scale(20.0), F.mel_scale(8000.0)
        centers = lo + (np.arange(80) + 1) * (hi - lo) / 81
        expect = int(np.argmin(np.abs(centers - F.mel_scale(f0))))
        assert abs(int(np.argmax(fb.mean(0))) - expect) <= 1


def test_deltas_of_a_ramp_and_cmvn_statistics():
    T = 50
    ramp = (3.0 * np.arange(T))[:, None] * np.ones((1, 4))
    d = F.compute_deltas(ramp)
    assert np.allclose(d[2:-2], 3.0)  # interior: exact slope
    assert np.allclose(d[0], (1 * 3 + 2 * 6) / 10.0)  # replicate padding at the left edge
    rng = np.random.default_rng(1)
    x = F.extract(rng.standard_normal(16000))
    assert x.shape == (98, 240)
    assert np.abs(x.mean(0)).max() < 1e-9 and np.abs(x.std(0, ddof=1) - 1).max() < 1e-6


def test_batch_is_zero_padded_like_pad_sequence():
    rng = np.random.default_rng(2)
    out = F.forward([rng.standard_normal(32000), rng.standard_normal(16000)])
    assert out.shape == (2, 198, 240)
    assert np.all(out[1, 98:] == 0) and np.abs(out[1, :98]).sum() > 0
