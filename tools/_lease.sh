set -u
mkdir -p gpurun_out/r06b
( time timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "layernorm" 2>&1 | tail -5 ) 2>&1 | tail -8
Q="--no-cpu-baseline --no-other-modes --no-parity"
for t in 1 2 1 2; do python bench.py $Q --dtype bf16 --steps 200 --warmup 10 --tune ln_rows=$t > gpurun_out/r06b/bench_bf16_lnrows${t}.json 2>/dev/null
python - <<PY
import json
x=json.loads(open('gpurun_out/r06b/bench_bf16_lnrows${t}.json').read().strip().splitlines()[-1]); k=x['kernels_ms_per_step']
print('bf16 ln_rows=${t}', x['ms_per_step'], x['clock_ghz'], 'ln1', k.get('layernorm:ln1'), 'ln2', k.get('layernorm:ln2'))
PY
done
for t in 1 2; do python bench.py $Q --steps 60 --warmup 5 --tune ln_rows=$t > gpurun_out/r06b/bench_fp32_lnrows${t}.json 2>/dev/null
python - <<PY
import json
x=json.loads(open('gpurun_out/r06b/bench_fp32_lnrows${t}.json').read().strip().splitlines()[-1]); k=x['kernels_ms_per_step']
print('fp32 ln_rows=${t}', x['ms_per_step'], x['clock_ghz'], 'ln1', k.get('layernorm:ln1'), 'ln2', k.get('layernorm:ln2'))
PY
done
