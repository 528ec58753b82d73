"""The `fbank` baseline upstream on the MI355X (s3prl_amd/csrc/fbank.hip through the C ABI and the hub entry) against
the numpy oracle (oracle/fbank_oracle.py, float64) — BASELINE configs[0]: 4 x 2 s @16 kHz random wavs."""

import numpy as np
import pytest

from oracle import fbank_oracle as F

pytestmark = pytest.mark.gpu


def _wavs(lengths, seed=1234):
    import torch

    torch.manual_seed(seed)
    return [torch.randn(n) for n in lengths]


@pytest.mark.parametrize("lengths", [[32000] * 4, [32000, 16000, 23456, 400], [160000, 48000]])
def test_fbank_hub_entry_matches_oracle(lengths):
    import torch
    import s3prl_amd.hub as hub

    expert = hub.fbank().cuda().eval()
    wavs = _wavs(lengths)
    with torch.no_grad():
        res = expert([w.cuda() for w in wavs])
    hs = res["hidden_states"]
    assert isinstance(hs, list) and len(hs) == 1 and res["last_hidden_state"] is hs[0]
    got = hs[0].cpu().numpy()
    ref = F.forward([w.numpy().astype(np.float64) for w in wavs])
    assert got.shape == ref.shape and got.shape[2] == 240
    assert expert.get_downsample_rates("hidden_states") == 160
    # consumers' length contract (nn/upstream.py:166-179): T within a few frames of n / 160
    assert abs(got.shape[1] - round(max(lengths) / 160)) < 5
    for b, n in enumerate(lengths):
        T = F.num_frames(n)
        assert np.all(got[b, T:] == 0)
        if T == 1:  # one frame: unbiased std is nan in torch, numpy and here alike
            assert np.isnan(got[b, 0]).all()
            continue
        # fp32 GEMM + log vs float64: the log of a small mel energy amplifies relative spectrum error; CMVN rescales by
        # 1/std (deltas-of-deltas have std ~0.1-0.3), hence an absolute tolerance on O(1) normalised features
        assert np.abs(got[b, :T] - ref[b, :T]).max() < 5e-3, f"utt {b}: {np.abs(got[b, :T] - ref[b, :T]).max():.2e}"
        assert np.abs(got[b, :T, :80] - ref[b, :T, :80]).max() < 5e-4


def test_fbank_no_cmvn_stages():
    """Without CMVN the three stages are visible separately: log-mel, delta, delta-delta."""
    import torch
    import s3prl_amd.hub as hub

    expert = hub.fbank_no_cmvn().cuda().eval()
    wavs = _wavs([32000, 20000], seed=5)
    got = expert([w.cuda() for w in wavs])["hidden_states"][0].cpu().numpy()
    ref = F.forward([w.numpy().astype(np.float64) for w in wavs], use_cmvn=False)
    for b, n in enumerate([32000, 20000]):
        T = F.num_frames(n)
        for lo, tol in ((0, 2e-4), (80, 2e-4), (160, 2e-4)):
            err = np.abs(got[b, :T, lo:lo + 80] - ref[b, :T, lo:lo + 80]).max()
            assert err < tol, f"utt {b} cols {lo}: {err:.2e}"


def test_fbank_misaligned_waveform_and_errors():
    import torch
    import s3prl_amd.hub as hub

    expert = hub.fbank().cuda().eval()
    base = torch.randn(32001).cuda()
    w = base[1:]  # 4-byte aligned only
    got = expert([w])["hidden_states"][0].cpu().numpy()
    ref = F.forward([w.cpu().numpy().astype(np.float64)])
    assert np.abs(got - ref).max() < 5e-3
    with pytest.raises(ValueError):
        expert([torch.randn(100).cuda()])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        expert([torch.randn(32000)])
