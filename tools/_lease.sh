set -u
mkdir -p gpurun_out/r06b
( time timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gn_stats or conv0" 2>&1 | tail -5 ) 2>&1 | tail -8
Q="--no-cpu-baseline --no-other-modes --no-parity"
for t in 0 1; do python bench.py $Q --dtype bf16 --steps 200 --warmup 10 --tune gn_lag_one_block=$t > gpurun_out/r06b/bench_bf16_gnlag$t.json 2>/dev/null; done
python bench.py $Q --steps 60 --warmup 5 > gpurun_out/r06b/bench_fp32_gnlag1.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06b/bench_*gnlag*.json')):
    x=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], x['ms_per_step'], x['clock_ghz'], 'gn_stats', x['kernels_ms_per_step'].get('gn_stats'), 'attention', x['kernels_ms_per_step'].get('attention'))
PY
