"""Shared body of the ``UpstreamExpert`` mirrors.

Contract kept from the reference (upstream/example/expert.py:11-77, upstream/interfaces.py:100-131):

* ``UpstreamExpert(ckpt, model_config=None, **kwargs)`` — unknown kwargs (``refresh=``, ``legacy=``, …) are accepted
  and ignored;
* ``forward(wavs: List[FloatTensor (n_i,)]) -> dict`` with ``"hidden_states"`` (tuple of NL+1 ``(B, T, D)`` fp32
  tensors on the wavs' device: the input of every Transformer layer, then the encoder output), plus
  ``"_hidden_states_info"``, ``"last_hidden_state"`` and ``"hidden_state_{i}"`` exactly as ``UpstreamBase.__call__``
  adds them (interfaces.py:119-129).  Because the dict is complete, this module registers NO hooks;
* ``get_downsample_rates(key) -> 320``.

The forward is inference-only (the HIP path has no backward): asking for gradients raises — for waveforms that require
grad at the call, and for the reference's fine-tuning flow (``upstream_trainable``: ``entry.model.train()`` + a forward
with autograd enabled, downstream/runner.py:258-262,296-301) at ``loss.backward()``: in training mode the states carry an
autograd node whose backward raises, so an expert that has no parameters is never silently "fine-tuned" as a frozen
one.  Frozen use (``.eval()`` or ``torch.no_grad()``) is unaffected, and an expert is CONSTRUCTED in eval mode (it has nothing
to train): only an explicit ``.train()`` — the runner's, or ``s3prl.nn.S3PRLUpstream.__init__``'s — arms the guard.  Waveforms on the CPU are
copied to the current GPU, encoded there, and the states are returned on the waveforms' device (that is a transfer,
not a fallback: without a GPU the call raises) — ``S3PRLUpstream.__init__`` probes every upstream with CPU pseudo
waveforms (nn/upstream.py:124-126).
"""

from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from ..ckpt import load_checkpoint
from ..config import EncoderConfig
from ..encoder import HipEncoder


class _NoBackward(torch.autograd.Function):
    """Identity whose backward refuses: marks the states of a training-mode forward (see the module docstring)."""

    @staticmethod
    def forward(ctx, states, anchor):
        return states.view_as(states)

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError(
            "s3prl_amd upstream experts are inference-only: a gradient reached the hidden states of a training-mode "
            "forward (the reference's `upstream_trainable` / fine-tuning flow), but the HIP encoder has no backward and "
            "the expert holds no parameters — it would be silently frozen.  Freeze it explicitly (`expert.freeze()`, "
            "`.eval()`, `torch.no_grad()` around the upstream, or S3PRL_AMD_TRAIN_MODE=detach) — a Featurizer / head on top "
            "still gets its gradients — or fine-tune with the reference s3prl expert")


class HipUpstreamExpert(torch.nn.Module):
    family = "hubert"

    def __init__(self, ckpt: str = None, model_config: str = None, dtype: str = None, **kwargs):
        super().__init__()
        if ckpt is None:
            raise ValueError("a converted checkpoint path is required (no network: `*_local(ckpt=...)`)")
        self.cfg, self._weights = load_checkpoint(ckpt, self.family)
        self.dtype = dtype or os.environ.get("S3PRL_AMD_DTYPE", "fp32")
        self._encoders: Dict[int, HipEncoder] = {}
        # a buffer so that .to(device) / .cuda() of the enclosing model has something to move and report
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)
        self.eval()  # nothing to train: `.train()` is the explicit request of a fine-tuning flow (see _guard_backward)

    @classmethod
    def from_weights(cls, cfg: EncoderConfig, weights, dtype: str = "fp32"):
        """Build directly from in-memory weights (benchmarks / tests)."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        assert cfg.family == cls.family
        self.cfg, self._weights, self.dtype, self._encoders = cfg, dict(weights), dtype, {}
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)
        return self.eval()

    def randomize_(self, seed: Optional[int] = None):
        """``S3PRLUpstream(randomize=True)`` (nn/upstream.py:27-35,119-120): re-draw every checkpoint tensor — vectors from
        N(mean, std) of their own values, matrices / conv kernels Xavier-normal — and drop the packed GPU copies, so the
        next forward builds the encoder from the new weights."""
        from ..nn import randomize_weights

        self._weights = randomize_weights(self._weights, seed)
        self._encoders.clear()
        return self

    # ---- what S3PRLUpstream learns from a probe forward (nn/upstream.py:124-140), without running one ----
    @property
    def num_layers(self) -> int:
        return self.cfg.num_hidden_states

    @property
    def hidden_sizes(self) -> List[int]:
        return [self.cfg.encoder_embed_dim] * self.num_layers

    def get_downsample_rates(self, key: str = None) -> int:
        return self.cfg.downsample_rate

    def _states_info(self, n: int):
        """The hook identifiers ``UpstreamBase.__call__`` reports as ``_hidden_states_info`` (interfaces.py:125;
        hubert/expert.py:36-43)."""
        NL = self.cfg.encoder_layers
        if n == NL + 1:
            return tuple(f"self.model.encoder.layers[{i}]" for i in range(NL)) + ("self.model.encoder",)
        return tuple(f"state_{i}" for i in range(n))

    def _compute_device(self, wavs: List[torch.Tensor]) -> torch.device:
        dev = wavs[0].device
        if dev.type == "cuda":
            return dev
        if dev.type == "cpu" and torch.cuda.is_available():
            return torch.device("cuda", torch.cuda.current_device())
        raise RuntimeError(
            "s3prl_amd runs the encoder on an MI355X only and no GPU is visible "
            "(there is deliberately no CPU fallback — use the reference s3prl expert on CPU)")

    def _encoder_for(self, device: torch.device) -> HipEncoder:
        if device.type != "cuda":
            raise RuntimeError(
                "s3prl_amd runs the encoder on an MI355X only; move the waveforms to the GPU "
                "(there is deliberately no CPU fallback — use the reference s3prl expert on CPU)")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._encoders:
            self._encoders[idx] = HipEncoder(self.cfg, self._weights, dtype=self.dtype, device=idx)
        return self._encoders[idx]

    def _check_inference(self, wavs):
        if torch.is_grad_enabled() and any(w.requires_grad for w in wavs):
            raise RuntimeError("s3prl_amd upstream experts are inference-only (no backward through the HIP encoder)")

    # What a TRAINING-MODE forward with autograd on does (the only situation the guard is about).  "raise" (default): the
    # states carry a node whose backward raises — a fine-tuning flow is told that nothing would be tuned.  "detach": the
    # states are plain constants and a one-time warning says so — for a parent module that was put in .train() to train a
    # Featurizer / head on top of a frozen upstream WITHOUT torch.no_grad() around it (s3prl.nn.S3PRLUpstream.__init__ leaves
    # its expert in train mode, nn/upstream.py:127; with "raise" that flow stops at loss.backward() although it needs no
    # encoder gradient).  Attribute `train_mode_policy` or env S3PRL_AMD_TRAIN_MODE.
    train_mode_policy = None
    _warned_detached = False

    def freeze(self):
        """Explicitly frozen use, whatever `.train()` calls reach this module later: the states are constants."""
        self.train_mode_policy = "detach"
        HipUpstreamExpert._warned_detached = True  # asked for: nothing to warn about
        return self

    def _guard_backward(self, states: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            policy = self.train_mode_policy or os.environ.get("S3PRL_AMD_TRAIN_MODE", "raise")
            if policy == "detach":
                if not HipUpstreamExpert._warned_detached:
                    import warnings

                    HipUpstreamExpert._warned_detached = True
                    warnings.warn("s3prl_amd: training-mode forward of an inference-only upstream — its hidden states are "
                                  "constants (no gradient reaches the encoder; it holds no parameters)", stacklevel=3)
                return states
            if policy != "raise":
                raise ValueError(f"train_mode_policy / S3PRL_AMD_TRAIN_MODE must be 'raise' or 'detach', not {policy!r}")
            return _NoBackward.apply(states, torch.zeros(0, device=states.device, requires_grad=True))
        return states

    def encode(self, wavs: List[torch.Tensor], n_max: int = None, selection: Optional[str] = None,
               out_dtype: Optional[str] = None) -> torch.Tensor:
        """(NS, B, T, D) on the compute GPU.  ``n_max``: global pad-to length for data-parallel shards."""
        self._check_inference(wavs)
        return self._guard_backward(self._encoder_for(self._compute_device(wavs)).forward(
            wavs, n_max=n_max, selection=selection, out_dtype=out_dtype))

    def encode_featurized(self, wavs: List[torch.Tensor], weights, normalize: bool = False, n_max: int = None,
                          selection: Optional[str] = None) -> torch.Tensor:
        """(B, T, D) fp32: the Featurizer's weighted sum computed as the encoder's epilogue (no per-layer slab)."""
        self._check_inference(wavs)
        return self._guard_backward(self._encoder_for(self._compute_device(wavs)).forward_featurized(
            wavs, weights, normalize, n_max=n_max, selection=selection))

    def _result(self, hs: torch.Tensor, wav_device: torch.device, full: bool = True):
        if hs.device != wav_device:
            hs = hs.to(wav_device)
        hidden_states = tuple(hs[l] for l in range(hs.shape[0]))
        if not full:
            return {"hidden_states": list(hidden_states)}
        result = {"_hidden_states_info": self._states_info(len(hidden_states)), "hidden_states": hidden_states,
                  "last_hidden_state": hidden_states[-1]}
        for i, h in enumerate(hidden_states):
            result[f"hidden_state_{i}"] = h
        return result

    def forward(self, wavs: List[torch.Tensor]):
        return self._result(self.encode(wavs), wavs[0].device)
