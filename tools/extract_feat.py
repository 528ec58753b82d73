#!/usr/bin/env python3
"""Dump the hidden states of an MI355X upstream to disk — the output side of the path.

Two layouts, as in the reference:
  * ``<output_dir>/<name>.pt``: the list of (B, T, D) CPU tensors of one batch — what ``tools/extract_feat.py:13-35``
    / ``utility/extract_feat.py:31-41`` of the reference write for the pseudo waveforms (their golden-vector recipe);
  * ``--per-utterance``: ``<output_dir>/<stem>.pt`` = one stacked (num_layer, T_i, D) tensor per utterance, trimmed to
    its own frame count — what ``task/dump_feature.py:25-40`` (DumpFeature) writes.
Input: ``--wavs a.wav b.wav ...`` (any sample rate, mono or multi-channel; converted to 16 kHz mono on the host with
scipy — file decoding / resampling is not part of the GPU path) or, without ``--wavs``, the reference's pseudo-waveform
recipe (``util/pseudo_data.py:52-77``: seeded lengths in [1 s, 3 s], ``torch.randn``).

    python tools/extract_feat.py hubert_local --ckpt converted/hubert_base_ls960.pt --output_dir feats --wavs a.wav
"""

import argparse
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SAMPLE_RATE = 16000


def load_wav_16k(path: str):
    """(n,) float32 at 16 kHz: PCM decode + channel mean + polyphase resampling (host side)."""
    import numpy as np
    from scipy.io import wavfile
    from scipy.signal import resample_poly

    sr, x = wavfile.read(path)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":  # 8-bit PCM
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if sr != SAMPLE_RATE:
        g = np.gcd(int(sr), SAMPLE_RATE)
        x = resample_poly(x, SAMPLE_RATE // g, int(sr) // g).astype(np.float32)
    return x


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("name", help="hub entry, e.g. hubert_local / wav2vec2_local / wavlm_local / unispeech_sat_local / fbank")
    ap.add_argument("--ckpt")
    ap.add_argument("--output_dir", default="./sample_hidden_states")
    ap.add_argument("--wavs", nargs="*", default=None)
    ap.add_argument("--per-utterance", action="store_true")
    ap.add_argument("--dtype", default=None, help="fp32 (default) / fp16 / bf16 operand mode of the encoder")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--refresh", action="store_true")
    args = ap.parse_args(argv)

    import torch

    import s3prl_amd.hub as hub
    from s3prl_amd.synth import pseudo_lengths

    out_dir = Path(args.output_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    batch_path = out_dir / f"{args.name}.pt"
    if batch_path.is_file() and not args.refresh and not args.per_utterance:
        return 0
    kwargs = {}
    if args.ckpt:
        kwargs["ckpt"] = args.ckpt
    if args.dtype:
        kwargs["dtype"] = args.dtype
    model = getattr(hub, args.name)(**kwargs).to(args.device).eval()

    if args.wavs:
        names = [Path(w).stem for w in args.wavs]
        wavs = [torch.from_numpy(load_wav_16k(w)) for w in args.wavs]
    else:
        lengths = pseudo_lengths()
        torch.manual_seed(0)
        names = [f"pseudo{i}" for i in range(len(lengths))]
        wavs = [torch.randn(n) for n in lengths]
    with torch.no_grad():
        hidden = model([w.to(args.device) for w in wavs])["hidden_states"]
    hs = [h.detach().cpu() for h in hidden]
    if args.per_utterance:
        rate = model.get_downsample_rates("hidden_states")
        for b, (name, w) in enumerate(zip(names, wavs)):
            frames = min(hs[0].shape[1], max(1, round(len(w) / rate)))
            torch.save(torch.stack([h[b, :frames] for h in hs], dim=0), str(out_dir / f"{name}.pt"))
    else:
        torch.save(hs, str(batch_path))
    return 0


if __name__ == "__main__":
    sys.exit(main())
