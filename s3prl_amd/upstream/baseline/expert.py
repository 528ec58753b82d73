"""``UpstreamExpert`` of the ``fbank`` baseline upstream (BASELINE configs[0]) on the MI355X.

Mirrors ``s3prl/upstream/baseline/expert.py:23-79`` for the kaldi-``fbank`` configurations (``fbank.yaml``,
``fbank_no_cmvn.yaml``): ``UpstreamExpert(model_config, **kwargs)``, ``forward(wavs) -> {"last_hidden_state",
"hidden_states": [feats]}`` with ``feats`` the zero-padded ``(B, T_max, 240)`` batch, ``get_downsample_rates`` = 160.
The arithmetic (framing + DC removal + pre-emphasis + povey window + DFT as one implicit GEMM on the raw PCM, mel /
log, deltas, CMVN) runs in ``libs3enc.so`` (``s3prl_amd/csrc/fbank.hip``); there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
from typing import List

import torch
import yaml

from ... import _lib

SAMPLE_RATE = 16000


class UpstreamExpert(torch.nn.Module):
    def __init__(self, model_config, **kwargs):
        """``model_config``: path of a yaml file in the reference's baseline schema (``kaldi: {feat_type, fbank: {...}}``,
        ``delta: {order, win_length}``, ``cmvn: {use_cmvn}``) or the already-parsed dict."""
        super().__init__()
        if isinstance(model_config, dict):
            self.config = model_config
        else:
            with open(model_config, "r") as f:
                self.config = yaml.load(f, Loader=yaml.FullLoader)
        if "kaldi" not in self.config or self.config["kaldi"].get("feat_type", "fbank") != "fbank":
            raise NotImplementedError("s3prl_amd implements the kaldi `fbank` baseline only (fbank / fbank_no_cmvn)")
        fb = dict(self.config["kaldi"].get("fbank", {}))
        if not fb.pop("use_log_fbank", True):
            raise NotImplementedError("use_log_fbank=False is not implemented")
        c = _lib.S3FbankConfig()
        c.sample_rate = SAMPLE_RATE
        c.num_mel_bins = int(fb.pop("num_mel_bins", 23))
        c.frame_length_ms = float(fb.pop("frame_length", 25.0))
        c.frame_shift_ms = float(fb.pop("frame_shift", 10.0))
        c.preemphasis = float(fb.pop("preemphasis_coefficient", 0.97))
        if fb:
            raise NotImplementedError(f"unsupported kaldi.fbank options: {sorted(fb)}")
        delta = self.config.get("delta", {})
        c.delta_order = int(delta.get("order", 2))
        c.delta_win_length = int(delta.get("win_length", 5))
        cmvn = self.config.get("cmvn", {})
        c.use_cmvn = int(bool(cmvn.get("use_cmvn", False)))
        c.cmvn_eps = float(cmvn.get("eps", 1e-10))
        self._c = c
        self.output_dim = c.num_mel_bins * (c.delta_order + 1)
        self.downsample_rate = round(c.frame_shift_ms * SAMPLE_RATE / 1000)
        self._lib = _lib.load()
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)

    def get_downsample_rates(self, key: str = None) -> int:
        return self.downsample_rate

    def num_frames(self, n: int) -> int:
        t = C.c_int32()
        _lib.check(self._lib.s3enc_fbank_num_frames(C.byref(self._c), int(n), C.byref(t)))
        return t.value

    def forward(self, wavs: List[torch.Tensor]):
        if len(wavs) == 0:
            raise ValueError("empty batch")
        dev = wavs[0].device
        if dev.type != "cuda":
            raise RuntimeError("s3prl_amd runs the fbank upstream on an MI355X only (no CPU fallback); move the wavs to the GPU")
        held = [w.to(device=dev, dtype=torch.float32).contiguous() for w in wavs]
        lengths = [int(w.numel()) for w in held]
        frames = [self.num_frames(n) for n in lengths]
        if min(frames) < 1:
            raise ValueError("an utterance is shorter than one 25 ms analysis window")
        B, T = len(held), max(frames)
        out = torch.empty((B, T, self.output_dim), dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * B)(*[w.data_ptr() for w in held])
        lens = (C.c_int64 * B)(*lengths)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = self._lib.s3enc_fbank_forward(C.byref(self._c), ptrs, lens, B, C.c_void_p(out.data_ptr()), T, idx,
                                               C.c_void_p(stream))
        _lib.check(rc, "s3enc_fbank_forward")
        return {"last_hidden_state": out, "hidden_states": [out]}
