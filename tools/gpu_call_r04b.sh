set -u
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent or gemm16_big_tiles or posconv" > $O/pytest_ops.log 2>&1
echo "pytest ops rc=$?" >> $O/summary.txt
timeout 300 tools/micro/gemm16_lab cmp 7 8 9 10 > $O/gemm16_cmp.md 2>&1
timeout 300 tools/micro/gemm16_lab cmpx 7 9 10 > $O/gemm16_cmp_fp16x2.md 2>&1
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "_pl or pretrained or fp16x2 or bit_for_bit" > $O/pytest_enc.log 2>&1
echo "pytest enc rc=$?" >> $O/summary.txt
timeout 600 python tools/parity_table.py hubert_base_pseudo wav2vec2_base_pseudo hubert_base_pl wav2vec2_base_pl hubert_base_10s_pl > $O/parity_pl.md 2> $O/parity_pl.err
Q="--no-cpu-baseline --no-other-modes --no-parity"
for m in 7 9 10; do
  timeout 200 python bench.py --dtype bf16 $Q --steps 100 --warmup 5 --tune gemm16_big=$m > $O/bench_bf16_m$m.json 2>/dev/null
  timeout 200 python bench.py --dtype fp16x2 $Q --steps 60 --warmup 5 --tune gemm16_big=$m > $O/bench_fp16x2_m$m.json 2>/dev/null
done
timeout 200 python bench.py --dtype fp32 $Q --steps 40 --warmup 3 > $O/bench_fp32.json 2>/dev/null
tail -n 3 $O/pytest_ops.log $O/pytest_enc.log
cat $O/gemm16_cmp.md $O/gemm16_cmp_fp16x2.md
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print('$f', d['ms_per_step'], round(d['value']), {n:k[n] for n in list(k)[:12]})"; done
