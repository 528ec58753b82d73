"""Featurizer (weighted sum over layers, csrc/featurizer.hip) against the reference formula in torch fp64
(s3prl/nn/upstream.py:312-328): forward, the gradient of the layer weights, layer selection, the slab / non-slab paths."""

import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(all_hs, weights, normalize):
    import torch
    import torch.nn.functional as F

    st = torch.stack([h.double() for h in all_hs], dim=0)
    if normalize:
        st = F.layer_norm(st, (st.shape[-1],))
    shape = st.shape[1:]
    w = F.softmax(weights.double(), dim=-1)
    return (w.unsqueeze(-1) * st.view(len(all_hs), -1)).sum(0).view(*shape)


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("D,L,sel", [(768, 13, None), (1024, 25, [0, 3, 24]), (240, 2, None), (64, 5, [1, 2, 3])])
def test_weighted_sum_forward_and_weight_gradient(D, L, sel, normalize):
    import torch
    from s3prl_amd.featurizer import Featurizer

    torch.manual_seed(0)
    B, T = 3, 37
    slab = torch.randn(L, B, T, D, device="cuda") * 2 + 0.5
    up = types.SimpleNamespace(num_layers=L, hidden_sizes=[D] * L, downsample_rates=[320] * L)
    fz = Featurizer(up, layer_selections=sel, normalize=normalize).cuda()
    with torch.no_grad():
        fz.weights.copy_(torch.randn(len(fz.layer_selections)))
    lens = [torch.full((B,), T, dtype=torch.long)] * L
    for as_slab in (True, False):
        hs = [slab[l] if as_slab else slab[l].clone() for l in range(L)]
        fz.weights.grad = None
        out, out_len = fz(hs, lens)
        assert out.shape == (B, T, D) and out_len is lens[0]
        picked = [hs[i] for i in fz.layer_selections]
        w64 = fz.weights.detach().double().clone().requires_grad_(True)
        ref = _ref(picked, w64, normalize)
        err = float((out.detach().double() - ref.detach()).norm() / ref.detach().norm())
        assert err < 2e-6, f"forward rel-err {err:.2e}"
        g = torch.randn_like(out)
        out.backward(g)
        ref.backward(g.double())
        gerr = float((fz.weights.grad.double() - w64.grad).norm() / w64.grad.norm())
        assert gerr < 1e-5, f"weight-gradient rel-err {gerr:.2e}"


def test_single_layer_passthrough_and_expert_output():
    import torch
    from s3prl_amd.featurizer import Featurizer
    from s3prl_amd.synth import named_config, synth_weights
    from s3prl_amd.upstream.hubert.expert import UpstreamExpert

    up1 = types.SimpleNamespace(num_layers=1, hidden_sizes=[240], downsample_rates=[160])
    fz1 = Featurizer(up1)
    h = torch.randn(2, 5, 240)
    assert fz1([h], [torch.tensor([5, 5])])[0] is h
    # on the encoder's own output: the 13 hidden_states are views of one slab and are read in place
    cfg = named_config("tiny_hubert")
    expert = UpstreamExpert.from_weights(cfg, synth_weights(cfg, 0))
    wavs = [torch.randn(16000).cuda(), torch.randn(12000).cuda()]
    hs = expert(wavs)["hidden_states"]
    L, D = len(hs), hs[0].shape[-1]
    fz = Featurizer(types.SimpleNamespace(num_layers=L, hidden_sizes=[D] * L, downsample_rates=[320] * L)).cuda()
    lens = [torch.tensor([49, 37])] * L
    out, _ = fz(list(hs), lens)
    ref = _ref(list(hs), fz.weights.detach(), False)
    assert float((out.double() - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("name", ["legacyfeat_tiny_hubert_sum", "legacyfeat_tiny_hubert_norm", "legacyfeat_tiny_hubert_layer2",
                                  "legacyfeat_tiny_wavlm_last"])
def test_legacy_featurizer_on_the_hip_expert_matches_the_reference_class(name):
    """The old interface (s3prl.upstream.interfaces.Featurizer, what downstream/runner.py instantiates) on the MI355X expert:
    probe forward on the GPU, feature / layer selection on the expert's result dict, the library's weighted sum (+ the
    layer weights' gradient), per-utterance cut — against fixtures made by running the reference class on the reference expert."""
    import json
    import os

    import torch
    from conftest import GOLDEN_DIR
    from oracle import encoder_oracle as O
    import importlib

    from s3prl_amd.nn import LegacyFeaturizer
    from s3prl_amd.synth import named_config, synth_wavs, synth_weights

    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = named_config(meta["config"])
    expert_cls = importlib.import_module(f"s3prl_amd.upstream.{cfg.family}.expert").UpstreamExpert
    expert = expert_cls.from_weights(cfg, synth_weights(cfg, meta["weight_seed"]))
    fz = LegacyFeaturizer(expert, feature_selection=meta["feature_selection"], upstream_device="cuda",
                          layer_selection=meta["layer_selection"], normalize=meta["normalize"])
    assert fz.output_dim == meta["output_dim"] and fz.downsample_rate == meta["downsample_rate"]
    if "feat_weights" in z:
        with torch.no_grad():
            fz.weights.copy_(torch.from_numpy(z["feat_weights"]))
        fz = fz.cuda()
    wavs = [torch.from_numpy(w).cuda() for w in synth_wavs(meta["lengths"], meta["wav_seed"])]
    outs = fz(wavs, expert(wavs))
    for b, o in enumerate(outs):
        assert tuple(o.shape) == z[f"out{b}"].shape
        assert O.rel_err(o.detach().cpu().numpy(), z[f"out{b}"]) < 1e-4, f"{name} utterance {b}"
    if "feat_weights" in z:  # training the layer weights through the old interface: gradient from the HIP backward kernel
        sum(o.square().sum() for o in outs).backward()
        assert fz.weights.grad is not None and torch.isfinite(fz.weights.grad).all() and float(fz.weights.grad.abs().sum()) > 0


def test_legacy_featurizer_backpropagates_into_states_that_carry_a_graph():
    """interfaces.py:134-272: the reference's weighted sum is plain torch, so a trainable upstream (`upstream_trainable`) gets
    its gradient through it.  The library's weighted sum only differentiates the layer weights: states with requires_grad
    take the torch form instead of silently dropping their gradient."""
    import torch
    from s3prl_amd.nn import LegacyFeaturizer

    L, B, T, D = 4, 2, 50, 64

    class Up(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.scale = torch.nn.Parameter(torch.ones(L, device="cuda"))

        def get_downsample_rates(self, key):
            return 320

        def forward(self, wavs):
            base = torch.arange(L * B * T * D, device="cuda", dtype=torch.float32).view(L, B, T, D) / (L * B * T * D)
            return {"hidden_states": [base[l] * self.scale[l] for l in range(L)]}

    up = Up()
    fz = LegacyFeaturizer(up, "hidden_states", upstream_device="cuda").cuda()
    wavs = [torch.randn(16000, device="cuda") for _ in range(B)]
    feats = fz(wavs, up(wavs))
    torch.stack(feats).sum().backward()
    assert up.scale.grad is not None and torch.all(up.scale.grad != 0)
    assert fz.weights.grad is not None
