set -u
mkdir -p gpurun_out/r06b
( time timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_ops_gpu.py -m gpu -q -x -k "fold or layernorm or gemm16" 2>&1 | tail -5 ) 2>&1 | tail -8
Q="--no-cpu-baseline --no-other-modes --no-parity"
run() { # name, args
  python bench.py $Q $2 > gpurun_out/r06b/bench_$1.json 2>/dev/null
  python - <<PY
import json
x=json.loads(open('gpurun_out/r06b/bench_$1.json').read().strip().splitlines()[-1]); k=x['kernels_ms_per_step']
print('$1', x['ms_per_step'], x['clock_ghz'], 'ln1', k.get('layernorm:ln1'), 'ln2', k.get('layernorm:ln2'), 'fc2', k.get('gemm:fc2'), 'out_proj', k.get('gemm:out_proj'))
PY
}
for t in 0 1 0 1; do run bf16_fold$t "--dtype bf16 --steps 200 --warmup 10 --tune ln1_fold=$t"; done
for t in 0 1; do run fp16x2_fold$t "--dtype fp16x2 --steps 150 --warmup 10 --tune ln1_fold=$t"; done
for t in 0 1; do run wavlmbp_bf16_fold$t "--model wavlm_base_plus --dtype bf16 --steps 100 --warmup 5 --tune ln1_fold=$t"; done
