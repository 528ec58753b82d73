"""End-to-end parity of the HIP encoder (through libs3enc's C ABI) against
  (1) the committed reference goldens (tests/golden/*.npz, produced by running s3prl on CPU), and
  (2) the numpy oracle on the same seeded inputs, including intermediate taps to localise a failure.
Tolerances: fp32 path 1e-4 rel (Frobenius, per layer; BASELINE target is 1e-3); 16-bit paths are reported
against their own looser bounds."""

import numpy as np
import pytest

from conftest import golden_names
from oracle import encoder_oracle as O

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


def _encoder(cfg, weights, dtype="fp32"):
    from s3prl_amd.encoder import HipEncoder

    return HipEncoder(cfg, weights, dtype=dtype)


def _run(enc, wavs, n_max=None, selection=None):
    import torch

    dev = [torch.from_numpy(w).cuda() for w in wavs]
    out = enc.forward(dev, n_max=n_max, selection=selection)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", golden_names())
def test_fp32_matches_reference_golden(name, golden_loader):
    meta, cfg, weights, wavs, golden, norms = golden_loader(name)
    enc = _encoder(cfg, weights)
    hs = _run(enc, wavs, selection=meta.get("selection"))
    assert list(hs.shape[1:]) == meta["shape"] and hs.shape[0] == meta.get("n_states", cfg.encoder_layers + 1)
    assert np.isfinite(hs).all()
    ts, cs = meta["t_stride"], meta["c_stride"]
    errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
    assert max(errs) < FP32_TOL, f"{name}: per-layer rel-err {['%.2e' % e for e in errs]}"
    for l in range(len(golden)):
        assert abs(np.linalg.norm(hs[l].astype(np.float64)) - norms[l]) / norms[l] < FP32_TOL
    enc.close()


@pytest.mark.parametrize("name", ["tiny_hubert_pad", "tiny_hubert_large_pad", "tiny_wavlm_large_pad"])
def test_fp32_taps_match_oracle(name, golden_loader):
    """Stage-by-stage: conv stack, projection, positional conv — names the first kernel that drifts."""
    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    enc = _encoder(cfg, weights)
    hs = _run(enc, wavs)
    taps = {}
    ref = O.forward(cfg, weights, wavs, dtype=np.float64, taps=taps)
    n = len(cfg.conv_layers)
    for i in range(n - 3, n):  # earlier conv activations live in ping-pong buffers that later layers overwrite
        got = enc.debug_tap(f"conv{i}").reshape(taps[f"conv{i}"].shape)
        assert O.rel_err(got, taps[f"conv{i}"]) < 2e-5, f"conv{i}"
    got = enc.debug_tap("proj").reshape(taps["proj"].shape)
    assert O.rel_err(got, taps["proj"]) < 2e-5, "proj"
    for l, r in enumerate(ref):
        assert O.rel_err(hs[l], r) < 5e-5, f"hidden_states[{l}]"
    enc.close()


def test_run_to_run_determinism(golden_loader):
    """Eval-mode determinism like test/test_upstream.py:118-123 — here bit-exact."""
    _, cfg, weights, wavs, _, _ = golden_loader("tiny_hubert_pad")
    enc = _encoder(cfg, weights)
    a = _run(enc, wavs)
    b = _run(enc, wavs)
    assert np.array_equal(a, b)
    enc.close()


def test_shard_with_global_nmax_equals_full_batch(golden_loader):
    """a data-parallel shard padded to the global n_max reproduces the full-batch rows."""
    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_hubert_pad")
    enc = _encoder(cfg, weights)
    full = _run(enc, wavs)
    shard = _run(enc, wavs[2:], n_max=max(meta["lengths"]))
    assert O.rel_err(shard, full[:, 2:]) < 1e-6
    enc.close()


# full-size shapes of BASELINE configs[3] / [4]: T = 499 / 749, D = 1024, H = 16, Dg = 64, K = 1024 / 4096 GEMM tiles,
# the 876-entry relative-position window — every kernel instance those configs time is compared with the reference here
FULL_SIZE = ["hubert_large_10s", "wavlm_large_15s_pad"]


@pytest.mark.parametrize("dtype,tol", [("bf16", 3e-2), ("fp16", 4e-3)])
@pytest.mark.parametrize("name", ["tiny_hubert_pad", "tiny_wavlm_large_pad", "hubert_base_pseudo", "hubert_large_pseudo",
                                  "tiny_distiller_pad", "tiny_wav2vec2_large_fsbefore", "tiny_data2vec_pad",
                                  "data2vec_base_pseudo", "tiny_multires_pad", "tiny_multires_large_pad", "tiny_multires3_pad",
                                  "multires_hubert_base_pseudo"] + FULL_SIZE)
def test_16bit_paths_close_to_reference(name, dtype, tol, golden_loader):
    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    enc = _encoder(cfg, weights, dtype=dtype)
    hs = _run(enc, wavs, selection=meta.get("selection"))
    assert np.isfinite(hs).all()
    ts, cs = meta["t_stride"], meta["c_stride"]
    errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
    assert max(errs) < tol, f"{name}/{dtype}: per-layer rel-err {['%.2e' % e for e in errs]}"
    enc.close()


def test_extra_short_input(golden_loader):
    """0.05 s inputs (EXTRA_SHORT_SEC, test/test_upstream.py:24,192-200): 800 samples -> 2 frames."""
    _, cfg, weights, _, _, _ = golden_loader("tiny_hubert_pad")
    from s3prl_amd.synth import synth_wavs

    wavs = synth_wavs([800, 800], seed=5)
    enc = _encoder(cfg, weights)
    hs = _run(enc, wavs)
    assert hs.shape[2] == 2
    ref = O.forward(cfg, weights, wavs, dtype=np.float64)
    for l, r in enumerate(ref):
        assert O.rel_err(hs[l], r) < 5e-5
    enc.close()


def test_errors_are_reported_not_thrown(golden_loader):
    import torch
    from s3prl_amd._lib import S3EncError

    _, cfg, weights, wavs, _, _ = golden_loader("tiny_hubert_pad")
    bad = dict(weights)
    bad.pop("encoder.layers.1.fc1.weight")
    with pytest.raises(S3EncError, match="missing tensor"):
        _encoder(cfg, bad)
    enc = _encoder(cfg, weights)
    with pytest.raises(ValueError):
        enc.forward([torch.zeros(100).cuda()])  # shorter than the receptive field
    enc.close()


def test_hub_expert_contract(tmp_path, golden_loader):
    """The drop-in boundary end to end: converted-checkpoint file -> hub entry -> UpstreamExpert -> result dict
    (SURVEY §8b), compared with the reference golden."""
    import torch
    import s3prl_amd.hub as hub
    from s3prl_amd.ckpt import save_checkpoint

    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_wavlm_pad")
    path = str(tmp_path / "wavlm.pt")
    save_checkpoint(path, cfg, weights)
    expert = hub.wavlm_local(ckpt=path, refresh=False).cuda().eval()
    with torch.no_grad():
        res = expert([torch.from_numpy(w).cuda() for w in wavs])
    hs = res["hidden_states"]
    assert isinstance(hs, tuple) and len(hs) == cfg.encoder_layers + 1
    assert res["last_hidden_state"] is hs[-1] and res["hidden_state_0"] is hs[0]
    assert all(h.is_cuda and h.dtype == torch.float32 and h.shape == hs[0].shape for h in hs)
    assert expert.get_downsample_rates("hidden_states") == 320
    for h, g in zip(hs, golden):
        assert O.rel_err(h.cpu().numpy(), g) < FP32_TOL
    # length contract S3PRLUpstream / Featurizer rely on (nn/upstream.py:166-179, interfaces.py:250-261)
    assert abs(hs[0].shape[1] - round(max(meta["lengths"]) / 320)) < 5


def test_layer_events_and_gather_single_rank(golden_loader):
    """The overlap plumbing of the data-parallel path (events recorded by the library, all-gather issued on a side
    stream) on a 1-rank RCCL group: the gathered slab must equal the local one."""
    import os
    import torch
    import torch.distributed as dist
    from s3prl_amd.parallel import gather_layers

    _, cfg, weights, wavs, _, _ = golden_loader("tiny_hubert_pad")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29613")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        enc = _encoder(cfg, weights)
        events = enc.layer_events()
        dev = [torch.from_numpy(w).cuda() for w in wavs]
        plain = enc.forward(dev).clone()
        hs = enc.forward(dev)
        gathered = gather_layers(hs, overlap_events=events)
        torch.cuda.synchronize()
        assert torch.equal(gathered, plain)
        enc.close()
    finally:
        dist.destroy_process_group()


# ---- BASELINE.json's full sizes: parity through properties that do not need a full-size CPU run -------------------

@pytest.mark.parametrize("model", ["hubert_base", "wav2vec2_base"])
def test_full_size_batch_rows_match_oracle_and_properties(model):
    """32 x 10 s @16 kHz (BASELINE configs[1] / the metric's workload), fp32.  In an equal-length batch every utterance
    is independent (GroupNorm is per (utterance, channel), no padding), so
      (1) rows 0 and 31 of the 32-batch must equal the ORACLE run on those two utterances alone (seconds on CPU),
      (2) permuting the batch permutes the output rows (bit-exact: same kernels, same per-row arithmetic),
      (3) a data-parallel shard (utterances 16..31, global n_max) reproduces its rows of the full batch bit-exactly,
      (4) post-LN models: every returned frame of hidden_states[1:] has zero mean / unit variance before the affine
          -> checked through the last layer with the known gamma/beta."""
    import torch

    from oracle import torch_oracle as TO
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config(model)
    weights = synth_weights(cfg, 0)
    enc = _encoder(cfg, weights)
    B, n = 32, 160000
    gen = torch.Generator(device="cuda").manual_seed(1234)
    wavs = [torch.randn(n, device="cuda", generator=gen) for _ in range(B)]
    full = enc.forward(wavs).clone()
    torch.cuda.synchronize()
    NL, T, D = cfg.encoder_layers, 499, cfg.encoder_embed_dim
    assert tuple(full.shape) == (NL + 1, B, T, D) and torch.isfinite(full).all()
    # (1) oracle on two utterances
    ref = TO.forward(cfg, TO.prepare(cfg, weights), [wavs[0].cpu(), wavs[31].cpu()])
    for l in range(NL + 1):
        got = full[l][[0, 31]].cpu().numpy()
        assert O.rel_err(got, ref[l].numpy()) < FP32_TOL, f"layer {l}"
    # (2) permutation equivariance
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).tolist()
    permuted = enc.forward([wavs[i] for i in perm])
    torch.cuda.synchronize()
    assert torch.equal(permuted, full[:, perm])
    # (3) shard with the global n_max
    shard = enc.forward(wavs[16:], n_max=n)
    torch.cuda.synchronize()
    assert torch.equal(shard, full[:, 16:])
    # (4) LayerNorm property of the post-LN stack
    if not cfg.layer_norm_first:
        g = torch.from_numpy(weights[f"encoder.layers.{NL - 1}.final_layer_norm.weight"]).cuda()
        bta = torch.from_numpy(weights[f"encoder.layers.{NL - 1}.final_layer_norm.bias"]).cuda()
        z = (full[NL] - bta) / g
        assert z.mean(-1).abs().max() < 1e-4 and (z.var(-1, unbiased=False) - 1).abs().max() < 1e-3
    enc.close()


def test_full_size_mixed_lengths_valid_frames_match_oracle():
    """WavLM-large-style gated relative-position attention at a BASELINE configs[4]-like shape cut to one GPU-second:
    8 utterances of mixed length up to 15 s (padding mask, pre-LN, layer-norm extractor, waveform normalisation);
    the shortest and the longest utterance are checked against the oracle run on the PAIR padded to the batch n_max
    (every batch coupling goes through n_max only)."""
    import torch

    from oracle import torch_oracle as TO
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("wavlm_base_plus")
    weights = synth_weights(cfg, 0)
    enc = _encoder(cfg, weights)
    rng = np.random.default_rng(1234)
    lens = [240000] + [int(x) for x in rng.integers(16000, 240000, size=7)]
    gen = torch.Generator(device="cuda").manual_seed(7)
    wavs = [torch.randn(n, device="cuda", generator=gen) for n in lens]
    full = enc.forward(wavs)
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()
    short = int(np.argmin(lens))
    pair = [wavs[0].cpu(), wavs[short].cpu()]
    ref = TO.forward(cfg, TO.prepare(cfg, weights), pair, n_max=240000)
    for l in range(cfg.encoder_layers + 1):
        got = full[l][[0, short]].cpu().numpy()
        assert O.rel_err(got, ref[l].numpy()) < FP32_TOL, f"layer {l}"
    enc.close()


@pytest.mark.parametrize("name", ["tiny_hubert", "tiny_wavlm_large"])
def test_long_single_utterance_and_one_frame_neighbour(name):
    """Edge shapes: B = 1 with 40 s of audio (T = 1999: the WavLM relative-position table spans 3997 entries, attention
    walks 63 key tiles) and a batch that pairs it with a 400-sample utterance (1 valid frame, everything else masked)."""
    import torch
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config(name)
    weights = synth_weights(cfg, 2)
    enc = _encoder(cfg, weights)
    for lengths in ([640000], [640000, 400]):
        wavs = synth_wavs(lengths, seed=9)
        hs = _run(enc, wavs)
        assert hs.shape[2] == 1999 and np.isfinite(hs).all()
        ref = O.forward(cfg, weights, wavs, dtype=np.float64)
        for l, r in enumerate(ref):
            assert O.rel_err(hs[l], r) < 5e-5, f"{lengths} layer {l}: {O.rel_err(hs[l], r):.2e}"
    enc.close()


@pytest.mark.parametrize("dtype", ["fp32", "fp32x3", "fp16x2", "bf16"])
def test_one_utterance_alone_equals_its_rows_in_a_batch_bit_for_bit(dtype):
    """A row's rounding must not depend on the batch it sits in (shards padded to the global n_max reproduce the
    full batch): one 2 s utterance alone is M = 99 rows — below every large tile's row count — and the same utterance among five
    is M = 495; the GEMM dispatch (tile heights, the two-term fp16x2 kernel that stages both weight terms, the 128x128 fallback)
    must give it the same bits both ways, in every operand mode."""
    import torch
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("hubert_base")
    enc = _encoder(cfg, synth_weights(cfg, 0), dtype=dtype)
    gen = torch.Generator(device="cuda").manual_seed(11)
    wavs = [torch.randn(n, device="cuda", generator=gen) for n in (32000, 32000, 20000, 32000, 9000)]
    full = enc.forward(wavs).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(full.float()).all()
    for i in (0, 2, 4):
        alone = enc.forward([wavs[i]], n_max=32000)
        torch.cuda.synchronize()
        assert torch.equal(alone[:, 0], full[:, i]), f"utterance {i} differs between B = 1 and B = 5 ({dtype})"
    enc.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_forwards_of_four_handles_on_four_streams_keep_their_bits(dtype):
    """Round 6: forwards of several handles enqueued at once on four streams (a serving process with one encoder per model, or one per
    worker thread) must give every utterance the bits it gets alone.  Without the forward chain (tuning key `forward_chain`, default 1:
    a forward waits on the device for the previous forward of any handle) the 16-bit modes were measured NOT bit-stable in this
    situation — rare rows a few 16-bit ulps off, tools/two_stream_probe.py, profiles/r06c_concurrent_forwards.md — and 28 % slower."""
    import ctypes as C

    import torch
    from s3prl_amd import _lib
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("hubert_base")
    weights = synth_weights(cfg, 0)
    S, per, n = 4, 8, 160000
    encs = [_encoder(cfg, weights, dtype=dtype) for _ in range(S)]
    gen = torch.Generator(device="cuda").manual_seed(77)
    wavs = [torch.randn(n, device="cuda", generator=gen) for _ in range(S * per)]
    lib = _lib.load()
    T, D, NS = encs[0].num_output_frames(n), encs[0].embed_dim, encs[0].num_states()
    streams = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize()

    def run(out, concurrent):
        keep = []
        for s, enc in enumerate(encs):
            sub = wavs[s * per:(s + 1) * per]
            ptrs = (C.c_void_p * per)(*[w.data_ptr() for w in sub])
            lens = (C.c_int64 * per)(*[n] * per)
            opts = _lib.S3ForwardOpts(_lib.SELECTIONS[None], _lib.F32, 0, 0, None)
            st = streams[s if concurrent else 0]
            _lib.check(lib.s3enc_forward_ex(enc._h, ptrs, lens, per, n, C.byref(opts), C.c_void_p(out.data_ptr() + s * per * T * D * 4),
                                            S * per * T * D, C.c_void_p(st.cuda_stream)), "s3enc_forward_ex")
            keep.append((ptrs, lens))
        return keep

    ref = torch.empty((NS, S * per, T, D), device="cuda")
    k0 = run(ref, False)
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all()
    for trial in range(3):
        out = torch.empty_like(ref)
        k1 = [run(out, True) for _ in range(3)]  # three forwards per handle back to back: the streams' phases mix
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"{dtype}: trial {trial}: forwards of four handles on four streams differ from the same forwards one at a time"
    for e in encs:
        e.close()


@pytest.mark.parametrize("dtype,tol", [("bf16", 3e-2), ("fp16x2", 2e-3)])
def test_conv0_fast_0_is_the_same_layer_at_the_modes_rounding(dtype, tol):
    """Tuning key `conv0_fast` (round 6, fourth session): 0 runs the first conv layer of the 16-bit modes on scalar taps and libm erff —
    the instantiation with which overlapping forwards of several handles were measured bit-stable WITHOUT the forward chain
    (profiles/r06d_concurrent_forwards_exclusions.md).  The same layer, evaluated in a different order: every state within the mode's
    rounding of the default kernel's, run-to-run bit-identical, and the default comes back bit for bit when the key is reset."""
    import torch
    from s3prl_amd import _lib
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("hubert_base")
    enc = _encoder(cfg, synth_weights(cfg, 0), dtype=dtype)
    gen = torch.Generator(device="cuda").manual_seed(11)
    wavs = [torch.randn(n, device="cuda", generator=gen) for n in (48000, 40000, 16000)]
    lib = _lib.load()
    try:
        ref = enc.forward(wavs).clone()
        _lib.check(lib.s3enc_set_tuning(b"conv0_fast", 0), "s3enc_set_tuning")
        a = enc.forward(wavs).clone()
        b = enc.forward(wavs).clone()
    finally:
        _lib.check(lib.s3enc_set_tuning(b"conv0_fast", 1), "s3enc_set_tuning")
    c = enc.forward(wavs).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)
    assert torch.equal(c, ref)
    assert not torch.equal(a, ref), "conv0_fast = 0 did not change the kernel"
    for l in range(ref.shape[0]):
        rel = float((a[l] - ref[l]).norm() / ref[l].norm())
        assert rel < tol, f"{dtype}: state {l}: conv0_fast = 0 is {rel:.2e} from the default kernel"
    enc.close()


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16", "fp16x2"])
def test_race_screen_repeated_runs_are_bit_identical(dtype):
    """The GEMM kernels overlap LDS-DMA (issued from inline asm, hand-counted vmcnt) with the MFMA loop; a missing wait
    or barrier would show as run-to-run differences.  HuBERT-base shapes (every GEMM mode of the path: 128x128 fp32,
    256x256 2-stage and 128x256 ring-of-3 16-bit), 8 runs, all bit-identical."""
    import torch
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("hubert_base")
    enc = _encoder(cfg, synth_weights(cfg, 0), dtype=dtype)
    gen = torch.Generator(device="cuda").manual_seed(5)
    wavs = [torch.randn(n, device="cuda", generator=gen) for n in (64000, 64000, 51234, 16000, 64000, 33333)]
    first = enc.forward(wavs).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(first).all()
    for _ in range(7):
        again = enc.forward(wavs)
        torch.cuda.synchronize()
        assert torch.equal(again, first)
    enc.close()


def test_extract_feat_tool_dumps_reference_layouts(tmp_path):
    """tools/extract_feat.py: wav files (8 kHz int16 stereo -> 16 kHz mono on the host) -> per-utterance
    (num_layer, T_i, D) dumps like task/dump_feature.py, and the batch list like the reference's tools/extract_feat.py."""
    import sys
    import torch
    from scipy.io import wavfile

    sys.path.insert(0, str((__import__("pathlib").Path(__file__).resolve().parents[1] / "tools")))
    import extract_feat as tool
    from s3prl_amd.ckpt import save_checkpoint
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("tiny_hubert")
    weights = synth_weights(cfg, 0)
    ckpt = str(tmp_path / "tiny.pt")
    save_checkpoint(ckpt, cfg, weights)
    rng = np.random.default_rng(0)
    paths = []
    for i, n in enumerate((8000, 5000)):  # 1.0 s and 0.625 s at 8 kHz
        pcm = (rng.standard_normal((n, 2)) * 3000).astype(np.int16)
        paths.append(str(tmp_path / f"utt{i}.wav"))
        wavfile.write(paths[-1], 8000, pcm)
    out = tmp_path / "feats"
    assert tool.main(["hubert_local", "--ckpt", ckpt, "--output_dir", str(out), "--wavs", *paths, "--per-utterance"]) == 0
    wavs = [tool.load_wav_16k(p) for p in paths]
    assert [len(w) for w in wavs] == [16000, 10000]
    ref = O.forward(cfg, weights, wavs, dtype=np.float64)
    for b, p in enumerate(paths):
        feat = torch.load(str(out / f"utt{b}.pt"))
        frames = min(cfg.num_frames(16000), round(len(wavs[b]) / 320))  # the Featurizer's length rule, capped at T
        assert tuple(feat.shape) == (cfg.encoder_layers + 1, frames, cfg.encoder_embed_dim)
        for l in range(cfg.encoder_layers + 1):
            assert O.rel_err(feat[l].numpy(), ref[l][b, :frames]) < 5e-5
    assert tool.main(["hubert_local", "--ckpt", ckpt, "--output_dir", str(out)]) == 0
    hs = torch.load(str(out / "hubert_local.pt"))
    assert isinstance(hs, list) and len(hs) == cfg.encoder_layers + 1 and hs[0].shape[0] == 2


@pytest.mark.parametrize("name", ["tiny_hubert_pad", "tiny_wavlm_large_pad", "hubert_base_pseudo", "hubert_large_pseudo",
                                  "tiny_distiller_pad", "distilhubert_pseudo", "tiny_wav2vec2_large_fslayers", "tiny_data2vec_pad",
                                  "data2vec_base_pseudo", "tiny_multires_pad", "tiny_multires_large_pad", "tiny_multires_plain_pad",
                                  "multires_hubert_base_pseudo"] + FULL_SIZE)
def test_fp32x3_split_precision_mode_close_to_reference(name, golden_loader):
    """compute_dtype S3ENC_F32X3 (fp32 data flow, GEMMs as three bf16 MFMAs per product): two orders tighter than the
    1e-3 target, one order looser than the exact fp32 mode."""
    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    enc = _encoder(cfg, weights, dtype="fp32x3")
    hs = _run(enc, wavs, selection=meta.get("selection"))
    assert np.isfinite(hs).all()
    ts, cs = meta["t_stride"], meta["c_stride"]
    errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
    assert max(errs) < 1e-4, f"{name}/fp32x3: per-layer rel-err {['%.2e' % e for e in errs]}"
    enc.close()


@pytest.mark.parametrize("name", ["tiny_hubert_pad", "tiny_wavlm_large_pad", "hubert_base_pseudo", "wav2vec2_base_pseudo",
                                  "wavlm_base_plus_pseudo", "hubert_large_pseudo", "distilhubert_pseudo", "data2vec_base_pseudo",
                                  "tiny_multires_pad", "multires_hubert_base_pseudo", "hf_hubert_base_pseudo"] + FULL_SIZE)
def test_fp16x2_two_term_mode_meets_the_path_tolerance(name, golden_loader):
    """compute_dtype S3ENC_F16X2: the fp16 data flow with every GEMM weight kept as two fp16 terms (w = hi + lo, the
    contraction runs over both).  Only the activations' rounding is left: every hidden state of every fixture — the full-size
    HuBERT-large 10 s and WavLM-large 15 s ones included — must sit inside the path's own 1e-3 tolerance (the plain fp16
    mode is at 0.9-1.4e-3), and strictly below the plain fp16 mode's error on the same fixture."""
    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    ts, cs = meta["t_stride"], meta["c_stride"]
    worst = {}
    for mode in ("fp16x2", "fp16"):
        enc = _encoder(cfg, weights, dtype=mode)
        hs = _run(enc, wavs, selection=meta.get("selection"))
        assert np.isfinite(hs).all()
        worst[mode] = max(O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden)))
        enc.close()
    assert worst["fp16x2"] < 1e-3, f"{name}/fp16x2: max per-layer rel-err {worst['fp16x2']:.3e}"
    assert worst["fp16x2"] < worst["fp16"], f"{name}: fp16x2 {worst['fp16x2']:.3e} vs fp16 {worst['fp16']:.3e}"


# ---- released-checkpoint statistics (round 4) -------------------------------------------------------------------------
# Fixtures `*_pl`: synth_weights(profile="pretrained_like") — residual-stream outlier channels (x30-100 writers on pre-LN
# models), LayerNorm gains of 2-4 on a few channels, Student-t matrices, score-shifting q / k biases, a near-silent and a
# near-constant conv0 filter, int16-scale PCM with a DC offset on the models without waveform normalisation — run through the
# reference (tests/golden/make_golden.py).  The fp32 fixtures are covered by test_fp32_matches_reference_golden above (every
# golden name); here the split-precision and 16-bit modes.  The reference's own regression test runs released checkpoints
# (test/test_upstream.py:118-136); this is the closest an offline box gets.
PRETRAINED_LIKE = [n for n in golden_names() if n.endswith("_pl")]


def test_pretrained_like_fixtures_are_present():
    assert {"hubert_base_pl", "hubert_large_pl", "wavlm_large_pl", "hubert_base_10s_pl", "wavlm_large_15s_pl"} <= set(PRETRAINED_LIKE)
    # round 6: the weight-seed sweep (fp32 <= 1e-4 by test_fp32_matches_reference_golden, fp32x3 <= 1e-4 and fp16x2 < 1e-3 below)
    sweep = {f"{m}_s{s}_pl" for m in ("hubert_base", "wav2vec2_base", "hubert_large", "wavlm_large", "data2vec_base") for s in range(1, 7)}
    assert sweep | {"wavlm_large_15s_s2_pl", "wavlm_large_15s_s3_pl"} <= set(PRETRAINED_LIKE)


@pytest.mark.parametrize("name", PRETRAINED_LIKE)
def test_fp32x3_on_pretrained_like_statistics(name, golden_loader):
    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    assert meta["profile"] == "pretrained_like"
    enc = _encoder(cfg, weights, dtype="fp32x3")
    hs = _run(enc, wavs)
    assert np.isfinite(hs).all()
    ts, cs = meta["t_stride"], meta["c_stride"]
    errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
    assert max(errs) < 1e-4, f"{name}/fp32x3: per-layer rel-err {['%.2e' % e for e in errs]}"
    enc.close()


# measured bounds (profiles/r04_parity.md): see DESIGN §5 for what each mode keeps of its synthetic-statistics error
PL_16BIT_TOL = {"fp16x2": 7.5e-4, "fp16": 4e-3, "bf16": 3e-2}  # fp16x2: round 5 (6.6e-4 worst: profiles/r05_parity.md)


def _pl_16bit_cases():
    """fp16x2 on EVERY pretrained-like fixture (the seed sweep included: that is what the sweep is for); the one-term modes fp16 / bf16
    — reported, not claimed to meet the tolerance — on the fixtures of rounds 4-5 and on seeds 1-2 of the sweep (the lease's time)."""
    from conftest import golden_meta

    out = []
    for n in PRETRAINED_LIKE:
        sweep = golden_meta(n).get("seed_sweep")
        for d in ("fp16x2", "fp16", "bf16"):
            if d == "fp16x2" or not sweep or "_s1_" in n or "_s2_" in n:
                out.append((n, d))
    return out


@pytest.mark.parametrize("name,dtype", _pl_16bit_cases())
def test_16bit_modes_on_pretrained_like_statistics(name, dtype, golden_loader):
    """Outlier channels of a few hundred, LayerNorm gains and int16-scale PCM must neither overflow the fp16 range nor
    produce a NaN anywhere, and each mode stays inside its own bound."""
    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    enc = _encoder(cfg, weights, dtype=dtype)
    hs = _run(enc, wavs)
    assert np.isfinite(hs).all(), f"{name}/{dtype}: non-finite hidden states"
    ts, cs = meta["t_stride"], meta["c_stride"]
    errs = [O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))]
    # (the tiny fixtures' dimensions are below what the fp16x2 hybrids take — C, D < 128: they keep the path's 1e-3; round 6's
    # weight-seed sweep — `*_s<seed>_pl`, seeds 1-6 x five models + WavLM-large 15 s seeds 2-3 — is held to the path's tolerance
    # itself, 1e-3 on EVERY seed: profiles/r06_parity_seeds.md has the distribution)
    tol = 1e-3 if dtype == "fp16x2" and (name.startswith("tiny_") or meta.get("seed_sweep")) else PL_16BIT_TOL[dtype]
    assert max(errs) < tol, f"{name}/{dtype}: per-layer rel-err {['%.2e' % e for e in errs]}"
    enc.close()


# ---- s3enc_forward_status (ABI 6): the library reports a non-finite forward instead of leaving the caller to scan ---------
def test_forward_status_is_clean_on_healthy_forwards_and_flags_non_finite_pcm():
    import torch

    from s3prl_amd import _lib
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config("tiny_hubert")
    weights = synth_weights(cfg, 1)
    wavs = synth_wavs([4000, 2345, 3111], 11)
    bad = [w.copy() for w in wavs]
    bad[1][1000] = np.inf
    dev_bad = [torch.from_numpy(w).cuda() for w in bad]
    for mode in ("fp32", "fp32x3", "fp16x2", "fp16", "bf16"):
        enc = _encoder(cfg, weights, dtype=mode)
        assert enc.check == "deferred"
        hs = _run(enc, wavs)
        assert np.isfinite(hs).all() and enc.status() == 0, mode
        enc.check = "off"
        out = enc.forward(dev_bad)
        assert enc.status(wait=False) & ~_lib.STATUS_PENDING in (0, _lib.STATUS_NONFINITE)  # a poll never blocks
        hs2 = _run(enc, wavs)                      # a healthy forward behind the bad one (synchronises)
        assert not np.isfinite(out.cpu().numpy()).all()
        enc.forward(dev_bad)
        with pytest.raises(FloatingPointError, match="non-finite"):
            enc.check_finite()                     # waits for the forwards in flight; the word is per forward, bits accumulate
        assert enc.status() == 0                   # reading clears
        assert np.array_equal(hs2, hs), mode
        # deferred: the error surfaces on a later forward, once the host is no longer ahead of the bad one
        enc.check = "deferred"
        with pytest.raises(FloatingPointError):
            enc.forward(dev_bad)
            torch.cuda.synchronize()
            enc.forward([torch.from_numpy(w).cuda() for w in wavs])
        enc.status()
        enc.close()


def test_strict_check_raises_on_the_forward_that_overflowed_fp16_only():
    """fc1 scaled until GELU(fc1) leaves the fp16 range (65504): the fp16 data flow stores inf as fc2's operand, the residual
    stream turns non-finite and the next LayerNorm reports it; the same weights are finite in fp32 and bf16."""
    import torch

    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config("tiny_hubert")
    weights = dict(synth_weights(cfg, 1))
    for k in list(weights):
        if k.endswith(".fc1.weight"):
            weights[k] = weights[k] * np.float32(3e5)
    dev = [torch.from_numpy(w).cuda() for w in synth_wavs([4000, 2345], 11)]
    for mode, overflows in (("fp32", False), ("bf16", False), ("fp16", True), ("fp16x2", True)):
        enc = HipEncoder(cfg, weights, dtype=mode, check="strict")
        if overflows:
            with pytest.raises(FloatingPointError, match="65504"):
                enc.forward(dev)
            assert enc.status() == 0
        else:
            assert torch.isfinite(enc.forward(dev)).all(), mode
        enc.close()


def test_status_poll_never_blocks():
    import torch

    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    cfg = named_config("hubert_base")
    enc = _encoder(cfg, synth_weights(cfg, 0))
    enc.check = "off"
    dev = [torch.from_numpy(w).cuda() for w in synth_wavs([48000] * 4, 3)]
    enc.forward(dev)
    from s3prl_amd import _lib

    first = enc.status(wait=False)   # STATUS_PENDING while the forward is still running, else the (clean) mask
    assert first in (_lib.STATUS_PENDING, 0)
    for _ in range(20):              # more forwards in flight than the pinned ring has slots: nothing is lost or blocks for long
        enc.forward(dev)
    assert enc.status(wait=True) == 0
    enc.close()


@pytest.mark.parametrize("name", ["hubert_base_pseudo", "hubert_base_pl", "wav2vec2_base_pl", "hubert_large_pl", "wavlm_large_pl",
                                  "data2vec_base_pseudo", "hubert_base_10s_pl"])
def test_fp16x2_mx_second_term_keeps_the_mode_inside_its_tolerance(name, golden_loader):
    """Round 5: the lo weight term of q|k|v / fc1 / fc2 as an MX-fp4 image (tuning key gemm16_mx = 14, the default) against two fp16
    terms (0): 4.8e-5 of weight error per GEMM instead of 5e-7 must leave every fixture inside 7.5e-4 (the mode's bound on the
    pretrained-like fixtures) and within 1e-4 of the two-term result's error; run-to-run bit-identical."""
    from s3prl_amd import _lib

    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    ts, cs = meta["t_stride"], meta["c_stride"]
    lib = _lib.load()
    errs, outs = {}, {}
    try:
        for mx in (14, 0):
            _lib.check(lib.s3enc_set_tuning(b"gemm16_mx", mx))
            enc = _encoder(cfg, weights, dtype="fp16x2")
            hs = _run(enc, wavs)
            if mx:
                assert np.array_equal(hs, _run(enc, wavs)), "not run-to-run bit-identical"
            assert np.isfinite(hs).all()
            errs[mx] = max(O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden)))
            outs[mx] = hs
            enc.close()
    finally:
        _lib.check(lib.s3enc_set_tuning(b"gemm16_mx", 14))
    assert not np.array_equal(outs[0], outs[14]), "the tuning key selected nothing: both runs took the same kernels"
    assert errs[14] < 7.5e-4, f"{name}: fp16x2 with the MX second term {errs[14]:.3e}"
    assert errs[14] < errs[0] + 1e-4, f"{name}: MX second term {errs[14]:.3e} vs two fp16 terms {errs[0]:.3e}"


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp16x2"])
@pytest.mark.parametrize("name", ["hubert_base_pl", "wav2vec2_base_pl", "wavlm_base_plus_pseudo", "data2vec_base_pseudo", "hubert_large_pl",
                                  "tiny_hubert_pad"])
def test_layernorm1_fold_into_fc2_is_bit_identical(name, dtype, golden_loader):
    """Round 6 (second session), tuning key `ln1_fold` (default 1): in the 16-bit modes a post-LN layer's LayerNorm 1 writes its 16-bit
    output and the rows' (mean, rstd) only; fc2's residual epilogue reads the LayerNorm INPUT row and rebuilds the fp32 output it adds
    (one shared `ln_affine`).  Every hidden state must equal the unfolded forward bit for bit — ragged batches, every fc2 kernel variant
    (two fp16 terms, the MX second term, bf16), a pre-LN model (the key selects nothing there) and a tiny one (fc2 off the big kernel)."""
    from s3prl_amd import _lib

    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    lib = _lib.load()
    outs = {}
    try:
        for on in (1, 0):
            _lib.check(lib.s3enc_set_tuning(b"ln1_fold", on))
            enc = _encoder(cfg, weights, dtype=dtype)
            outs[on] = _run(enc, wavs)
            enc.close()
    finally:
        _lib.check(lib.s3enc_set_tuning(b"ln1_fold", 1))
    assert np.isfinite(outs[1]).all()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["wavlm_large_s1_pl", "hubert_base_s1_pl", "data2vec_base_s2_pl", "tiny_hubert_large_pad"])
def test_fp16x2_conv1_on_fp32_rows_option(name, golden_loader):
    """Round 6, tuning key `fp16x2_conv1_f32` (read at s3enc_create): conv0 writes fp32 and conv1 takes the three-term GEMM like
    conv2.. — the mode's last fp16 rounding inside the conv stack.  On the worst fixtures of the weight-seed sweep it must lower the
    error (WavLM-large seed 1: 8.1e-4 -> ~7e-4) and stay a different, finite, run-to-run bit-identical result; on a tiny fixture
    (C < 128: no hybrid at all) it must change nothing."""
    from s3prl_amd import _lib

    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    ts, cs = meta["t_stride"], meta["c_stride"]
    lib = _lib.load()
    errs, outs = {}, {}
    try:
        for on in (1, 0):
            _lib.check(lib.s3enc_set_tuning(b"fp16x2_conv1_f32", on))
            enc = _encoder(cfg, weights, dtype="fp16x2")
            hs = _run(enc, wavs)
            assert np.isfinite(hs).all() and np.array_equal(hs, _run(enc, wavs))
            errs[on] = max(O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden)))
            outs[on] = hs
            enc.close()
    finally:
        _lib.check(lib.s3enc_set_tuning(b"fp16x2_conv1_f32", 0))
    if name.startswith("tiny_"):
        assert np.array_equal(outs[0], outs[1])
        return
    assert not np.array_equal(outs[0], outs[1]), "the tuning key selected nothing"
    assert errs[1] < 1e-3 and errs[1] < errs[0], f"{name}: conv1 on fp32 rows {errs[1]:.3e} vs default {errs[0]:.3e}"
