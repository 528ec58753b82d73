"""One line per random architecture of tests/test_fuzz_gpu.py: its shape and the max per-layer error of every operand mode
(GPU box).  usage: fuzz_report.py [n_seeds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_fuzz_gpu import _random_config
from oracle import encoder_oracle as O
from s3prl_amd.encoder import HipEncoder
from s3prl_amd.synth import synth_wavs, synth_weights
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(1000 + seed)
    cfg = _random_config(rng)
    weights = synth_weights(cfg, seed)
    rate = cfg.downsample_rate
    B = int(rng.integers(1, 5))
    lengths = [int(rng.integers(14 * rate, 40 * rate)) for _ in range(B)]
    wavs = synth_wavs(lengths, seed + 1, dc=float(rng.choice([0.0, 0.2])), scale=float(rng.choice([1.0, 0.1])))
    ref = O.forward(cfg, weights, wavs, dtype=np.float32)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    desc = f"{cfg.family} D={cfg.encoder_embed_dim} F={cfg.encoder_ffn_embed_dim} C={cfg.conv_dim} conv={[(k,s) for _,k,s in cfg.conv_layers]} pos={cfg.conv_pos}/{cfg.conv_pos_groups} depth={cfg.pos_conv_depth} prel={cfg.layer_norm_first} ext={cfg.extractor_mode} rel={cfg.relative_position_embedding} mr={cfg.label_rate_ratios}{'p' if cfg.use_plain_updownsample else ''} k={cfg.conv_adapter_kernel} T={cfg.num_frames(max(lengths))} B={B}"
    res = []
    for dt in ("fp32", "bf16", "fp32x3"):
        try:
            enc = HipEncoder(cfg, weights, dtype=dt)
            hs = enc.forward(dev).cpu().numpy()
            enc.close()
            e = max(O.rel_err(hs[l], ref[l]) for l in range(len(ref)))
            res.append(f"{dt}:{e:.1e}" + ("" if np.isfinite(hs).all() else " NONFINITE"))
        except Exception as ex:
            res.append(f"{dt}:EXC {str(ex)[-90:]}")
    print(seed, desc, "|", " ; ".join(res), flush=True)
