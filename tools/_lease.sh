set -u
mkdir -p gpurun_out/r06b
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_encoder_gpu.py -m gpu -q -x -k "layernorm or fold or hub_expert" 2>&1 | grep -E "passed|failed|error" | tail -3 )
Q="--no-cpu-baseline --no-other-modes --no-parity"
run() { # name, args
  python bench.py $Q $2 > gpurun_out/r06b/bench_$1.json 2>/dev/null
  python - <<PY
import json
x=json.loads(open('gpurun_out/r06b/bench_$1.json').read().strip().splitlines()[-1]); k=x['kernels_ms_per_step']
print('$1', x['ms_per_step'], x['clock_ghz'], 'ln1', k.get('layernorm:ln1'), 'ln2', k.get('layernorm:ln2'), 'conv', k.get('layernorm:conv'))
PY
}
for t in 1 1; do run bf16_pre$t "--dtype bf16 --steps 200 --warmup 10 --tune ln_preload=$t"; done
for t in 1 1; do run fp32_pre$t "--steps 50 --warmup 5 --tune ln_preload=$t"; done
for t in 1; do run hl_bf16_pre$t "--model hubert_large --dtype bf16 --steps 60 --warmup 5 --tune ln_preload=$t"; done
for t in 1 2; do run bf16_plain_rows$t "--dtype bf16 --steps 200 --warmup 10 --tune ln_rows=$t"; done
