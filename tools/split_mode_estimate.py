"""(How the two-term mode S3ENC_F16X2 was decided, before it existed.)  Would a 2-term split mode (fp16 activations x (fp16 hi + fp16 lo) weights) meet 1e-3?  Its error is the fp16 mode's error
with the WEIGHT rounding removed — measurable today: run the fp16 mode on weights that are already fp16-representable and
compare with the exact-fp32 mode on the same weights."""
import sys, json
import numpy as np, torch
sys.path.insert(0, ".")
from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import named_config, synth_wavs, synth_weights
from oracle import encoder_oracle as O

def run(cfg_name, lengths, seed=0):
    cfg = named_config(cfg_name)
    w = synth_weights(cfg, seed)
    w16 = {k: torch.from_numpy(np.asarray(v, np.float32)).half().float().numpy() for k, v in w.items()}
    wavs = [torch.from_numpy(x).cuda() for x in synth_wavs(lengths, 5)]
    out = {}
    for tag, weights, dt in (("fp32(w)", w, "fp32"), ("fp16(w)", w, "fp16"), ("fp32(w16)", w16, "fp32"), ("fp16(w16)", w16, "fp16")):
        enc = HipEncoder(cfg, weights, dtype=dt, device=0)
        out[tag] = enc.forward(wavs).float().cpu().numpy()
        del enc
    e_full = max(O.rel_err(out["fp16(w)"][l], out["fp32(w)"][l]) for l in range(out["fp32(w)"].shape[0]))
    e_act = max(O.rel_err(out["fp16(w16)"][l], out["fp32(w16)"][l]) for l in range(out["fp32(w)"].shape[0]))
    print(f"{cfg_name} {lengths}: fp16 mode vs fp32 (same weights) {e_full:.3e};  activation-rounding-only (2-term estimate) {e_act:.3e}")

run("hubert_base", [40000, 32000])
run("wav2vec2_base", [40000, 27123])
run("wavlm_base_plus", [40000, 32000])
run("hubert_large", [40000, 32000])
run("wavlm_large", [60000, 31234])
