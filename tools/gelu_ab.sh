mkdir -p gpurun_out/r03h
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
Q="--no-cpu-baseline --no-other-modes"
python bench.py $Q --steps 60 --warmup 3 > gpurun_out/r03h/b_fp32.json 2>/dev/null
python bench.py $Q --steps 60 --warmup 3 --tune gelu32=0 > gpurun_out/r03h/b_fp32_libm.json 2>/dev/null
python bench.py $Q --dtype bf16 --steps 300 > gpurun_out/r03h/b_bf16.json 2>/dev/null
python bench.py $Q --model hubert_large --dtype bf16 --steps 60 --warmup 2 > gpurun_out/r03h/b_hubert_large_bf16.json 2>/dev/null
python bench.py $Q --model wavlm_large --dtype bf16 --secs 15 --mixed --steps 40 --warmup 2 > gpurun_out/r03h/b_wavlm_bf16.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03h/b_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d.get('kernels_ms_per_step',{})
        print(f.split('/')[-1], d['ms_per_step'], d['roofline']['frac'], (d.get('parity') or {}).get('max_layer_rel_err_vs_torch_oracle'), 'conv0',k.get('conv0'),'ln',{x:k[x] for x in k if x.startswith('layernorm')})
    except Exception as e: print(f, 'ERR', e)
PY
