set -u
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_fuzz_gpu.py tests/test_multires_gpu.py -x -q -m gpu -k "fp16x2 or _pl or bit_for_bit or race or fuzz or multires" > $O/pytest_enc.log 2>&1
echo "pytest enc rc=$?" >> $O/summary.txt
timeout 900 python tools/parity_table.py hubert_base_pseudo wav2vec2_base_pseudo wavlm_base_plus_pseudo distilhubert_pseudo data2vec_base_pseudo hubert_base_pl wav2vec2_base_pl hubert_base_10s_pl hubert_large_pl wavlm_large_pl hubert_large_10s_pl > $O/parity.md 2> $O/parity.err
Q="--no-cpu-baseline --no-other-modes --no-parity"
timeout 200 python bench.py --dtype fp16x2 $Q --steps 60 --warmup 5 > $O/bench_fp16x2.json 2>/dev/null
timeout 200 python bench.py --dtype bf16 $Q --steps 100 --warmup 5 > $O/bench_bf16.json 2>/dev/null
timeout 200 python bench.py --dtype bf16 $Q --steps 100 --warmup 5 --tune gemm16_rows=0 > $O/bench_bf16_rows0.json 2>/dev/null
timeout 300 python bench.py --model hubert_large --dtype fp16x2 $Q --steps 12 --warmup 2 > $O/bench_large_fp16x2.json 2>/dev/null
tail -n 3 $O/pytest_enc.log
cat $O/parity.md | tail -12
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print('$f', d['ms_per_step'], round(d['value']), {n:k[n] for n in list(k)[:14]})"; done
