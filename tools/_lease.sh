set -u
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( python -m pytest tests/test_encoder_gpu.py -m gpu -q -k "conv1_on_fp32 or pretrained_like_statistics" 2>&1 | tail -6 ) > gpurun_out/r06/gputests_lease5.log 2>&1
Q="--no-cpu-baseline --no-other-modes"
python bench.py $Q --dtype fp16x2 --steps 150 --warmup 5 > gpurun_out/r06/bench_fp16x2_l5.json 2>/dev/null
python bench.py $Q --dtype fp16x2 --steps 150 --warmup 5 --tune fp16x2_conv1_f32=1 > gpurun_out/r06/bench_fp16x2_conv1f32_l5.json 2>/dev/null
python bench.py $Q --model wavlm_large --secs 15 --mixed --dtype fp16x2 --steps 30 --warmup 2 > gpurun_out/r06/bench_cfg4_fp16x2_l5.json 2>/dev/null
python bench.py $Q --model wavlm_large --secs 15 --mixed --dtype fp16x2 --steps 30 --warmup 2 --tune fp16x2_conv1_f32=1 > gpurun_out/r06/bench_cfg4_fp16x2_conv1f32_l5.json 2>/dev/null
python - <<'PY' > gpurun_out/r06/parity_seeds_conv1f32.md 2>&1
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import golden_names, load_golden
from oracle import encoder_oracle as O
from s3prl_amd import _lib
from s3prl_amd.encoder import HipEncoder
lib = _lib.load()
print("| fixture | fp16x2 default | fp16x2, fp16x2_conv1_f32 = 1 |\n|---|---:|---:|")
worst = [0, 0]
for name in [n for n in golden_names() if n.endswith("_pl") and not n.startswith("tiny_")]:
    meta, cfg, weights, wavs, golden, _ = load_golden(name)
    dev = [torch.from_numpy(w).cuda() for w in wavs]
    ts, cs = meta["t_stride"], meta["c_stride"]
    row = []
    for on in (0, 1):
        _lib.check(lib.s3enc_set_tuning(b"fp16x2_conv1_f32", on))
        enc = HipEncoder(cfg, weights, dtype="fp16x2")
        hs = enc.forward(dev).cpu().numpy()
        row.append(max(O.rel_err(hs[l][:, ::ts, ::cs], golden[l]) for l in range(len(golden))))
        enc.close()
        worst[on] = max(worst[on], row[-1])
    print(f"| `{name}` | {row[0]:.2e} | {row[1]:.2e} |", flush=True)
_lib.check(lib.s3enc_set_tuning(b"fp16x2_conv1_f32", 0))
print(f"| **worst** | **{worst[0]:.2e}** | **{worst[1]:.2e}** |")
PY
tail -5 gpurun_out/r06/gputests_lease5.log
tail -3 gpurun_out/r06/parity_seeds_conv1f32.md
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/bench_*_l5.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
        print(f.split('/')[-1], d['ms_per_step'], d['clock_ghz'], 'conv0', k.get('conv0'), 'conv1', k.get('gemm:conv1'), 'parity', d.get('parity',{}).get('max_layer_rel_err_vs_torch_oracle'))
    except Exception as e: print(f, 'unreadable', e)
PY
