"""Round-2 rows of the path on the GPU, through the C ABI: state selections, 16-bit state output, the Featurizer as the
encoder's epilogue (s3enc_forward_ex featurize), the S3PRLUpstream / Featurizer mirror against fixtures produced by the
reference's own s3prl.nn classes, DistilHuBERT's expert dict, variable-T WavLM batches, CPU-resident waveforms."""

import types

import numpy as np
import pytest

from oracle import encoder_oracle as O
from oracle import featurizer_oracle as FO
from test_featurizer_cpu import FEAT, load_feat

pytestmark = pytest.mark.gpu


def _enc(cfg, weights, dtype="fp32"):
    from s3prl_amd.encoder import HipEncoder

    return HipEncoder(cfg, weights, dtype=dtype)


def _dev(wavs):
    import torch

    return [torch.from_numpy(w).cuda() for w in wavs]


CASES = [("tiny_hubert_pad", None), ("tiny_hubert_large_pad", None), ("tiny_wavlm_large_pad", None), ("tiny_wavlm_pad", None),
         ("tiny_wav2vec2_pad", "fairseq_layers"), ("tiny_wav2vec2_large_pad", "fairseq_layers"),
         ("tiny_wav2vec2_pad", "fairseq_layers_before_residual"), ("tiny_wav2vec2_large_pad", "fairseq_layers_before_residual"),
         ("tiny_distiller_pad", None), ("hubert_base_pseudo", None), ("wavlm_large_pseudo", None), ("tiny_data2vec_pad", None)]


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("name,selection", CASES)
def test_featurize_epilogue_equals_weighted_sum_of_the_states(name, selection, normalize, golden_loader):
    """s3enc_forward_ex(featurize) never writes the states; its single (B, T, D) output must equal the Featurizer formula
    (nn/upstream.py:312-328) applied to the states a plain forward returns — fp64 on the host as the reference value,
    with zero weights on some layers (layer_selections)."""
    import torch

    meta, cfg, weights, wavs, _, _ = golden_loader(name)
    enc = _enc(cfg, weights)
    dev = _dev(wavs)
    states = enc.forward(dev, selection=selection)
    NS = states.shape[0]
    assert NS == enc.num_states(selection)
    rng = np.random.default_rng(NS + int(normalize))
    w = np.exp(rng.standard_normal(NS))
    if NS > 2:
        w[1] = 0.0  # an unselected layer
    w = (w / w.sum()).astype(np.float32)
    fused = enc.forward_featurized(dev, w.tolist(), normalize=normalize, selection=selection)
    torch.cuda.synchronize()
    st = states.double()
    if normalize:
        st = torch.nn.functional.layer_norm(st, (st.shape[-1],))
    ref = (torch.from_numpy(w).double().cuda().view(-1, 1, 1, 1) * st).sum(0)
    err = float((fused.double() - ref).norm() / ref.norm())
    assert err < 2e-6, f"{name}/{selection}/norm={normalize}: rel-err {err:.2e}"
    # all-zero weights: a defined result (zeros), not stale memory
    z = enc.forward_featurized(dev, [0.0] * NS, selection=selection)
    torch.cuda.synchronize()
    assert float(z.abs().max()) == 0.0
    enc.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("name,selection", CASES[:9])
def test_16bit_state_output_is_the_rounded_fp32_output(name, selection, dtype, golden_loader):
    """out_dtype = the compute dtype: the producing kernels store the same fp32 values rounded to 16 bit (RNE), so the
    16-bit slab equals the fp32 slab cast by torch — bit for bit."""
    import torch

    meta, cfg, weights, wavs, _, _ = golden_loader(name)
    enc = _enc(cfg, weights, dtype=dtype)
    dev = _dev(wavs)
    h32 = enc.forward(dev, selection=selection).clone()
    h16 = enc.forward(dev, selection=selection, out_dtype=dtype)
    torch.cuda.synchronize()
    assert h16.dtype == (torch.bfloat16 if dtype == "bf16" else torch.float16) and h16.shape == h32.shape
    assert torch.equal(h16, h32.to(h16.dtype))
    with pytest.raises(ValueError):
        _enc(cfg, weights).forward(dev, out_dtype=dtype)  # fp32 encoder: only fp32 states
    enc.close()


def test_featurize_in_16bit_modes_and_fp32x3(golden_loader):
    import torch

    meta, cfg, weights, wavs, _, _ = golden_loader("hubert_base_pseudo")
    dev = _dev(wavs)
    for dtype, tol in (("bf16", 2e-6), ("fp32x3", 2e-6)):  # vs the SAME mode's states: only the summation order differs
        enc = _enc(cfg, weights, dtype=dtype)
        states = enc.forward(dev).double()
        w = torch.softmax(torch.arange(states.shape[0], dtype=torch.float64) * 0.3, 0)
        fused = enc.forward_featurized(dev, w.tolist())
        torch.cuda.synchronize()
        ref = (w.cuda().view(-1, 1, 1, 1) * states).sum(0)
        assert float((fused.double() - ref).norm() / ref.norm()) < tol
        enc.close()


@pytest.mark.parametrize("name", FEAT)
def test_s3prl_upstream_and_featurizer_mirror_match_the_reference_classes(name, tmp_path):
    """s3prl_amd.nn.S3PRLUpstream / Featurizer / UpstreamFeaturizer against fixtures produced by the reference's own
    s3prl.nn.S3PRLUpstream + s3prl.nn.Featurizer (length matching, last-frame re-padding, MIN_SECOND extension,
    per-layer layer norm, layer selection, softmax weights)."""
    import torch

    from s3prl_amd.ckpt import save_checkpoint
    from s3prl_amd.nn import Featurizer, S3PRLUpstream, UpstreamFeaturizer

    meta, cfg, weights, wavs, z = load_feat(name)
    path = str(tmp_path / "c.pt")
    save_checkpoint(path, cfg, weights)
    up = S3PRLUpstream(f"{cfg.family}_local", path_or_url=path, normalize=meta["upstream_normalize"]).cuda().eval()
    assert up.num_layers == meta["num_layers"] and set(up.downsample_rates) == {320}
    fz = Featurizer(up, layer_selections=meta["layer_selections"], normalize=meta["featurizer_normalize"]).cuda()
    with torch.no_grad():
        fz.weights.copy_(torch.from_numpy(z["feat_weights"]))
    n = max(meta["lengths"])
    padded = torch.zeros(len(wavs), n)
    for b, w in enumerate(wavs):
        padded[b, : len(w)] = torch.from_numpy(w)
    lens = torch.tensor(meta["lengths"])
    with torch.no_grad():
        all_hs, all_lens = up(padded.cuda(), lens.cuda())
        feat, feat_len = fz(all_hs, all_lens)
        fused, fused_len = UpstreamFeaturizer(up, fz)(padded.cuda(), lens.cuda())
    assert len(all_hs) == meta["num_layers"]
    for l, h in enumerate(all_hs):
        assert tuple(h.shape) == z[f"hs{l}"].shape
        assert O.rel_err(h.cpu().numpy(), z[f"hs{l}"]) < 1e-4, f"layer {l}"
        assert np.array_equal(all_lens[l].cpu().numpy(), z["lens"][l])
    assert O.rel_err(feat.cpu().numpy(), z["feat"]) < 1e-4
    assert np.array_equal(feat_len.cpu().numpy(), z["feat_len"])
    # both normalisations on means LN twice in the reference and once in the fused kernel (LN is idempotent up to eps)
    assert tuple(fused.shape) == z["feat"].shape and O.rel_err(fused.cpu().numpy(), z["feat"]) < 1e-4
    assert np.array_equal(fused_len.cpu().numpy(), z["feat_len"])


def test_featurizer_accepts_transposed_layers():
    """The reference experts' hooks return (B, T, D) VIEWS of (T, B, D) memory (hubert/expert.py:39): dense, not row-major."""
    import torch

    from s3prl_amd.featurizer import Featurizer

    torch.manual_seed(1)
    L, B, T, D = 4, 3, 21, 128
    base = [torch.randn(T, B, D, device="cuda") for _ in range(L)]
    hs = [x.transpose(0, 1) for x in base]
    assert not hs[0].is_contiguous()
    fz = Featurizer(types.SimpleNamespace(num_layers=L, hidden_sizes=[D] * L, downsample_rates=[320] * L)).cuda()
    with torch.no_grad():
        fz.weights.copy_(torch.randn(L))
    out, _ = fz(hs, [torch.full((B,), T)] * L)
    w = torch.softmax(fz.weights.detach().double(), 0)
    ref = sum(w[l] * hs[l].double() for l in range(L))
    assert out.is_contiguous() and float((out.double() - ref).norm() / ref.norm()) < 2e-6


def test_expert_dict_cpu_wavs_and_hub_signatures(tmp_path, golden_loader):
    import torch

    import s3prl_amd.hub as hub
    from s3prl_amd.ckpt import save_checkpoint

    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_hubert_pad")
    path = str(tmp_path / "h.pt")
    save_checkpoint(path, cfg, weights)
    expert = hub.hubert_custom(path, False, False, False).eval()  # positional legacy / fairseq / refresh like the reference
    cpu = [torch.from_numpy(w) for w in wavs]
    with torch.no_grad():
        res = expert(cpu)  # CPU-resident waveforms: encoded on the GPU, returned on the CPU (S3PRLUpstream's probe)
    NL = cfg.encoder_layers
    assert res["_hidden_states_info"] == tuple(f"self.model.encoder.layers[{i}]" for i in range(NL)) + ("self.model.encoder",)
    assert all(h.device.type == "cpu" for h in res["hidden_states"])
    for h, g in zip(res["hidden_states"], golden):
        assert O.rel_err(h.numpy(), g) < 1e-4
    with pytest.raises(ValueError, match="not a fairseq checkpoint"):
        hub.hubert_custom(path, legacy=True)  # legacy=True takes the ORIGINAL fairseq layout (converted here without `fairseq`)


def test_wav2vec2_feature_selection_expert(tmp_path, golden_loader):
    import torch

    import s3prl_amd.hub as hub
    from s3prl_amd.ckpt import save_checkpoint

    for name in ("tiny_wav2vec2_fslayers", "tiny_wav2vec2_large_fsbefore"):
        meta, cfg, weights, wavs, golden, _ = golden_loader(name)
        path = str(tmp_path / f"{name}.pt")
        save_checkpoint(path, cfg, weights)
        expert = hub.wav2vec2_local(path, feature_selection=meta["selection"])
        with torch.no_grad():
            res = expert(_dev(wavs))
        assert set(res) == {"hidden_states"} and len(res["hidden_states"]) == cfg.encoder_layers
        for h, g in zip(res["hidden_states"], golden):
            assert O.rel_err(h.cpu().numpy(), g) < 1e-4


def test_distiller_expert_dict(tmp_path, golden_loader):
    import torch

    import s3prl_amd.hub as hub
    from s3prl_amd.ckpt import save_checkpoint

    meta, cfg, weights, wavs, golden, _ = golden_loader("tiny_distiller_pad")
    path = str(tmp_path / "d.pt")
    save_checkpoint(path, cfg, weights)
    expert = hub.distiller_local(path)
    with torch.no_grad():
        res = expert(_dev(wavs))
        res_np = expert(_dev(wavs), no_pred=True)
    hs = res["hidden_states"]
    assert len(hs) == 1 + cfg.encoder_layers + cfg.pred_heads and res["last_hidden_state"] is hs[-1]
    assert res["paper"] is hs[cfg.encoder_layers]
    for h, g in zip(hs, golden):
        assert O.rel_err(h.cpu().numpy(), g) < 1e-4
    assert len(res_np["hidden_states"]) == 1 + cfg.encoder_layers and res_np["last_hidden_state"] is None
    valid = [cfg.valid_frames(n, max(meta["lengths"])) for n in meta["lengths"]]
    assert res["pad_mask"].sum(1).long().tolist() == valid


def test_wavlm_batches_of_changing_length_need_no_rebuild(golden_loader):
    """One (H, 2R+1) relative-position table serves every T (the bucket saturates at max_distance): batches whose T
    changes every call — below and above R — all match the oracle, on one handle, with no host synchronisation in between."""
    import torch

    from s3prl_amd.synth import synth_wavs

    _, cfg, weights, _, _, _ = golden_loader("tiny_wavlm_large_pad")  # max_distance 64: T > 65 clamps
    enc = _enc(cfg, weights)
    outs, refs = [], []
    for i, lengths in enumerate(([4000, 2345], [40000, 31000, 1500], [16000], [52000, 800])):
        wavs = synth_wavs(lengths, seed=50 + i)
        outs.append(enc.forward(_dev(wavs)))
        refs.append(O.forward(cfg, weights, wavs, dtype=np.float64))
    torch.cuda.synchronize()
    for out, ref in zip(outs, refs):
        for l, r in enumerate(ref):
            assert O.rel_err(out[l].cpu().numpy(), r) < 5e-5
    enc.close()


def test_forward_ex_argument_errors(golden_loader):
    import ctypes as C

    import torch

    from s3prl_amd import _lib
    from s3prl_amd._lib import S3EncError

    _, cfg, weights, wavs, _, _ = golden_loader("tiny_hubert_pad")
    enc = _enc(cfg, weights)
    dev = _dev(wavs)
    with pytest.raises(ValueError):
        enc.forward_featurized(dev, [1.0])  # wrong number of weights
    out = torch.empty((cfg.encoder_layers + 1, len(dev), 12, cfg.encoder_embed_dim), device="cuda")
    ptrs = (C.c_void_p * len(dev))(*[w.data_ptr() for w in dev])
    lens = (C.c_int64 * len(dev))(*[w.numel() for w in dev])
    lib = _lib.load()
    rc = lib.s3enc_forward_ex(enc._h, ptrs, lens, len(dev), 0, None, C.c_void_p(out.data_ptr()), out[0].numel() + 2, None)
    with pytest.raises(S3EncError, match="multiple of 4"):
        _lib.check(rc)
    opts = _lib.S3ForwardOpts(7, 0, 0, 0, None)
    rc = lib.s3enc_forward_ex(enc._h, ptrs, lens, len(dev), 0, C.byref(opts), C.c_void_p(out.data_ptr()), out[0].numel(), None)
    with pytest.raises(S3EncError, match="selection"):
        _lib.check(rc)
    enc.close()


@pytest.mark.parametrize("name", ["hf_tiny_hubert_large_pad", "hf_tiny_wav2vec2_pad", "hf_hubert_base_pseudo"])
def test_huggingface_checkpoint_experts_match_the_reference_hf_experts(name, tmp_path, golden_loader):
    """hf_hubert_custom / hf_wav2vec2_custom on a Hugging Face checkpoint DIRECTORY (config.json + model.safetensors +
    preprocessor_config.json, read without transformers) against fixtures produced by the reference's hf_hubert /
    hf_wav2vec2 experts running transformers on the same directory: HF's conv-length frame mask for HuBERT too, the
    feature extractor's eps-1e-7 normalisation, result dict = {"hidden_states"} only."""
    import torch

    import s3prl_amd.hub as hub
    from hf_util import write_hf_dir
    from s3prl_amd.synth import named_config, synth_weights

    meta, cfg, weights, wavs, golden, _ = golden_loader(name)
    src_cfg = named_config(meta["config"])
    path = write_hf_dir(str(tmp_path / "ckpt"), src_cfg, synth_weights(src_cfg, meta["weight_seed"]), meta["hf"])
    expert = getattr(hub, f"hf_{meta['hf']}_custom")(path)
    with torch.no_grad():
        res = expert(_dev(wavs))
    assert set(res) == {"hidden_states"} and len(res["hidden_states"]) == cfg.encoder_layers + 1
    ts, cs = meta["t_stride"], meta["c_stride"]
    for h, g in zip(res["hidden_states"], golden):
        assert O.rel_err(h.cpu().numpy()[:, ::ts, ::cs], g) < 1e-4
    wrong = "hf_wav2vec2_custom" if meta["hf"] == "hubert" else "hf_hubert_custom"
    with pytest.raises(ValueError):
        getattr(hub, wrong)(path)
