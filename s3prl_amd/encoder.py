"""Host-side handle on the HIP encoder: owns a ``s3enc_handle``; torch is used only for device memory
(outputs are torch-allocated so the caller owns them, SURVEY §8b "Ownership") and for the current stream."""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .config import EncoderConfig


class HipEncoder:
    """wav2vec2 / HuBERT / WavLM encoder forward on one MI355X through ``libs3enc.so``."""

    def __init__(self, cfg: EncoderConfig, weights: Dict[str, "np.ndarray"], dtype: str = "fp32",
                 device: Optional[int] = None, check: Optional[str] = None):
        """``check``: what a forward does about the library's non-finite flag (``check_finite``) — "deferred" (default: a
        non-blocking poll after every forward, so an overflow raises at the latest on the next forward), "strict" (one
        stream synchronisation per forward, raises on the forward that overflowed) or "off"; env ``S3PRL_AMD_CHECK``."""
        import os

        import torch

        cfg.validate()
        self.cfg = cfg
        self.dtype = dtype
        self.check = check or os.environ.get("S3PRL_AMD_CHECK", "deferred")
        if self.check not in ("deferred", "strict", "off"):
            raise ValueError(f"check must be 'deferred', 'strict' or 'off', not {self.check!r}")
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.S3EncError("no GPU visible: the s3prl_amd encoder has no CPU fallback (use the reference s3prl on CPU)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        ccfg = _lib.make_config(cfg, dtype)
        keep = []
        tensors = (_lib.S3Tensor * len(weights))()
        for i, (name, w) in enumerate(weights.items()):
            if hasattr(w, "detach"):
                w = w.detach().cpu().float().numpy()
            a = np.ascontiguousarray(w, dtype=np.float32)
            keep.append(a)
            tensors[i].name = name.encode()
            tensors[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
            tensors[i].ndim = min(a.ndim, 4)
            shape = list(a.shape) if a.ndim <= 4 else [int(np.prod(a.shape[:-3]))] + list(a.shape[-3:])
            for j, s in enumerate(shape):
                tensors[i].shape[j] = int(s)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.s3enc_create(C.byref(ccfg), tensors, len(weights), self.device, C.byref(h)), "s3enc_create")
        self._h = h
        self.num_layers = cfg.encoder_layers
        self.embed_dim = cfg.encoder_embed_dim
        self._forwards = 0   # forwards issued / the last one known to have been judged by check_finite (error messages name the range)
        self._checked = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s3enc_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- shape helpers (answered by the library so host and device agree) ----
    def num_frames(self, n: int) -> int:
        t = C.c_int32()
        _lib.check(self._lib.s3enc_num_frames(self._h, int(n), C.byref(t)))
        return t.value

    def num_output_frames(self, n: int) -> int:
        """Frames of the states a forward writes for a batch padded to ``n`` samples (``num_frames`` except for
        multires-HuBERT, whose states are cut to the common length of its resolutions)."""
        t = C.c_int32()
        _lib.check(self._lib.s3enc_num_output_frames(self._h, int(n), C.byref(t)))
        return t.value

    def valid_frames(self, length: int, n_max: int) -> int:
        v = C.c_int32()
        _lib.check(self._lib.s3enc_valid_frames(self._h, int(length), int(n_max), C.byref(v)))
        return v.value

    def downsample_rate(self) -> int:
        r = C.c_int32()
        _lib.check(self._lib.s3enc_downsample_rate(self._h, C.byref(r)))
        return r.value

    def num_states(self, selection=None) -> int:
        """Entries of the states list under ``selection`` (None / "fairseq_layers" / "fairseq_layers_before_residual")."""
        n = C.c_int32()
        _lib.check(self._lib.s3enc_num_states(self._h, _lib.SELECTIONS[selection], C.byref(n)), "s3enc_num_states")
        return n.value

    # ---- numerical health (s3enc_forward_status, ABI 6) ----
    def status(self, wait: bool = True) -> int:
        """OR of the status bits of every FINISHED forward since the last read (0 = healthy; reading clears).  ``wait=False``
        never blocks: ``_lib.STATUS_PENDING`` is set while forwards are still running (their bits come with a later read)."""
        st = C.c_int32()
        _lib.check(self._lib.s3enc_forward_status(self._h, int(bool(wait)), C.byref(st)), "s3enc_forward_status")
        return st.value

    def check_finite(self, wait: bool = True) -> None:
        """Raise ``FloatingPointError`` if a forward since the last check produced a non-finite LayerNorm statistic — in the
        16-bit modes the sign of an fp16 / bf16 overflow on the way to the hidden states (an fp32 run only gets there from
        non-finite PCM).  ``wait=False`` never blocks: a forward still in flight is checked by the next call."""
        issued = self._forwards
        st = self.status(wait)
        settled = wait or not (st & _lib.STATUS_PENDING)   # every forward issued so far has reported
        first = self._checked + 1
        if settled:
            self._checked = issued
        if st & _lib.STATUS_NONFINITE:
            # the word accumulates over the forwards that finished since the last read: name the range, so that a deferred
            # poll after forward N + k is not read as "forward N + k is bad" (its own output may be healthy)
            which = (f"forward #{issued}" if first >= issued else f"one of the forwards #{first}..#{issued}") + \
                    f" of this encoder ({'all of them have finished' if settled else 'the later ones are still running and are not judged yet'})"
            raise FloatingPointError(
                f"libs3enc ({self.dtype}): {which} produced non-finite activations (a row LayerNorm met an inf / NaN "
                "mean or variance) — its hidden states are not valid.  In fp16 / fp16x2 this is a range overflow "
                "(|x| > 65504): use compute dtype fp32x3 / fp32 (or bf16) for this checkpoint; in fp32 check the waveforms.  "
                "check='strict' raises on the forward that overflowed; a one-shot caller should call check_finite() after its forward")

    def _after_forward(self):
        self._forwards += 1
        # strict: the caller pays one stream synchronisation per forward and gets the error on the forward that overflowed;
        # deferred (default): a non-blocking poll of the forwards that have finished — an overflow surfaces on one of the next
        # forwards (as soon as the host is no longer ahead of it) or at check_finite()
        if self.check == "strict":
            self.check_finite(wait=True)
        elif self.check == "deferred":
            self.check_finite(wait=False)

    # ---- forward ----
    def _prepare(self, wavs, n_max):
        import torch

        B = len(wavs)
        if B == 0:
            raise ValueError("empty batch")
        dev = torch.device("cuda", self.device)
        held = []
        for w in wavs:
            if w.dim() != 1:
                raise ValueError("each wav must be a 1-D tensor of samples")
            if w.device != dev or w.dtype != torch.float32 or not w.is_contiguous():
                w = w.to(device=dev, dtype=torch.float32).contiguous()
            held.append(w)
        lengths = [int(w.numel()) for w in held]
        nm = max(lengths) if n_max is None else int(n_max)
        if self.num_frames(nm) < 1:
            raise ValueError(f"input of {nm} samples is shorter than the receptive field of the conv stack")
        T = self.num_output_frames(nm)
        if T < 1:
            raise ValueError(f"input of {nm} samples is too short for the coarsest resolution of the model")
        return dev, held, lengths, nm, T

    def forward(self, wavs: Sequence["torch.Tensor"], n_max: Optional[int] = None, out: Optional["torch.Tensor"] = None,
                selection: Optional[str] = None, out_dtype: Optional[str] = None):
        """wavs: list of 1-D fp32 CUDA tensors.  Returns a (NS, B, T, D) CUDA tensor; ``[i]`` is state i of
        ``selection`` (default: ``hidden_states``).  ``n_max``: pad-to length of the GLOBAL batch (data-parallel
        shards).  ``out_dtype``: None / "fp32", or the encoder's own 16-bit compute dtype ("bf16" / "fp16") to get
        the states as 16-bit tensors (half the bytes to write and to all-gather)."""
        import torch

        dev, held, lengths, nm, T = self._prepare(wavs, n_max)
        B, D = len(held), self.embed_dim
        NS = self.num_states(selection)
        tdt, code = torch.float32, _lib.F32
        if out_dtype not in (None, "fp32", "f32", "float32"):
            fold = lambda c: _lib.F16 if c == _lib.F16X2 else c  # the two-term split mode runs the fp16 data flow
            code, own = fold(_lib.DTYPES[out_dtype]), fold(_lib.DTYPES[self.dtype])
            if code != own or code not in (_lib.BF16, _lib.F16):
                raise ValueError(f"out_dtype {out_dtype!r}: only fp32 or the encoder's own 16-bit compute dtype ({self.dtype})")
            tdt = torch.bfloat16 if code == _lib.BF16 else torch.float16
        if out is None:
            out = torch.empty((NS, B, T, D), dtype=tdt, device=dev)
        else:
            assert out.is_contiguous() and out.dtype == tdt and tuple(out.shape) == (NS, B, T, D)
        ptrs = (C.c_void_p * B)(*[w.data_ptr() for w in held])
        lens = (C.c_int64 * B)(*lengths)
        opts = _lib.S3ForwardOpts(_lib.SELECTIONS[selection], code, 0, 0, None)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = self._lib.s3enc_forward_ex(self._h, ptrs, lens, B, nm, C.byref(opts), C.c_void_p(out.data_ptr()), B * T * D,
                                            C.c_void_p(stream))
        _lib.check(rc, "s3enc_forward")
        self._after_forward()
        return out

    def forward_featurized(self, wavs: Sequence["torch.Tensor"], weights: Sequence[float], normalize: bool = False,
                           n_max: Optional[int] = None, out: Optional["torch.Tensor"] = None,
                           selection: Optional[str] = None):
        """The Featurizer's weighted sum as the encoder's epilogue: returns ONLY
        ``sum_i weights[i] * (layer_norm(state_i) if normalize else state_i)`` as one fp32 (B, T, D) tensor — the
        states never leave the workspace, so the encoder writes 1/(NL+1) of the bytes and a data-parallel exchange
        moves one layer.  ``weights``: one float per state (softmax already applied; 0 = layer not selected)."""
        import torch

        dev, held, lengths, nm, T = self._prepare(wavs, n_max)
        B, D = len(held), self.embed_dim
        NS = self.num_states(selection)
        w = [float(x) for x in weights]
        if len(w) != NS:
            raise ValueError(f"need one weight per state ({NS}), got {len(w)}")
        if out is None:
            out = torch.empty((B, T, D), dtype=torch.float32, device=dev)
        else:
            assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, T, D)
        wp = (C.c_float * NS)(*w)
        ptrs = (C.c_void_p * B)(*[x.data_ptr() for x in held])
        lens = (C.c_int64 * B)(*lengths)
        opts = _lib.S3ForwardOpts(_lib.SELECTIONS[selection], _lib.F32, 1, int(bool(normalize)), wp)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = self._lib.s3enc_forward_ex(self._h, ptrs, lens, B, nm, C.byref(opts), C.c_void_p(out.data_ptr()), 0,
                                            C.c_void_p(stream))
        _lib.check(rc, "s3enc_forward (featurize)")
        self._after_forward()
        return out

    def layer_events(self):
        """One CUDA event per state of the default selection, recorded by every following forward when state l is
        final (created once)."""
        import torch

        if getattr(self, "_events", None) is None:
            dev = torch.device("cuda", self.device)
            evs = []
            with torch.cuda.device(dev):
                for _ in range(self.num_states()):
                    ev = torch.cuda.Event(enable_timing=False)
                    ev.record(torch.cuda.current_stream(dev))  # torch creates the hipEvent_t lazily: force it
                    evs.append(ev)
            raw = (C.c_void_p * len(evs))(*[int(ev.cuda_event) for ev in evs])
            _lib.check(self._lib.s3enc_set_layer_events(self._h, raw, len(evs)), "s3enc_set_layer_events")
            self._events = evs
        return self._events

    # ---- measurement / test hooks ----
    def profile_enable(self, on=True):
        """True / 1: HIP events around every kernel; 2: only around the GEMM launches; False / 0: off."""
        _lib.check(self._lib.s3enc_profile_enable(self._h, int(on)))

    def profile_reset(self):
        _lib.check(self._lib.s3enc_profile_reset(self._h))

    def profile_read(self) -> List[dict]:
        n = C.c_int32()
        ents = (_lib.S3ProfileEntry * 128)()
        _lib.check(self._lib.s3enc_profile_read(self._h, ents, 128, C.byref(n)))
        return [dict(name=ents[i].name.decode(), launches=int(ents[i].launches), ms=float(ents[i].ms),
                     flops=float(ents[i].flops), bytes=float(ents[i].bytes)) for i in range(n.value)]

    def debug_tap(self, name: str) -> "np.ndarray":
        n = C.c_int64()
        _lib.check(self._lib.s3enc_debug_tap(self._h, name.encode(), None, 0, C.byref(n)))
        buf = np.empty(n.value, dtype=np.float32)
        _lib.check(self._lib.s3enc_debug_tap(self._h, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), n.value,
                                             C.byref(n)))
        return buf
