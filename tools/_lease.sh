set -u
mkdir -p gpurun_out/r06b
( time timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_encoder_gpu.py -m gpu -q -x -k "attention or 16bit or bf16 or fp16 or wavlm" 2>&1 | tail -5 ) 2>&1 | tail -8
Q="--no-cpu-baseline --no-other-modes --no-parity"
for d in bf16 fp16x2; do python bench.py $Q --dtype $d --steps 200 --warmup 10 > gpurun_out/r06b/bench_${d}_attn420.json 2>/dev/null; done
python bench.py $Q --model wavlm_large --dtype fp16x2 --secs 15 --mixed --steps 40 --warmup 5 > gpurun_out/r06b/bench_cfg4_fp16x2_attn420.json 2>gpurun_out/r06b/bench_cfg4.err
python bench.py $Q --model hubert_large --dtype bf16 --steps 60 --warmup 5 > gpurun_out/r06b/bench_cfg3_bf16_attn420.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06b/bench_*attn420.json')):
    try:
        x=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], x['ms_per_step'], x['clock_ghz'], 'attention', x['kernels_ms_per_step'].get('attention'))
    except Exception as e: print(f, 'bad', e)
PY
tail -3 gpurun_out/r06b/bench_cfg4.err
