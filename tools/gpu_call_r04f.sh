set -u
export TMPDIR=/tmp
O=gpurun_out/r04f
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" > $O/pytest_ops.log 2>&1
echo "pytest ops rc=$?" >> $O/summary.txt
timeout 200 tools/micro/gemm16_lab cmp 7 > $O/gemm16_cmp.md 2>&1
Q="--no-cpu-baseline --no-other-modes --no-parity"
timeout 200 python bench.py --dtype bf16 $Q --steps 100 --warmup 5 > $O/bench_bf16.json 2>/dev/null
timeout 200 python bench.py --dtype fp16x2 $Q --steps 60 --warmup 5 > $O/bench_fp16x2.json 2>/dev/null
timeout 200 python bench.py --dtype fp32x3 $Q --steps 40 --warmup 5 > $O/bench_fp32x3.json 2>/dev/null
timeout 300 python bench.py --model hubert_large --dtype bf16 $Q --steps 20 --warmup 2 > $O/bench_large_bf16.json 2>/dev/null
tail -n 3 $O/pytest_ops.log
cat $O/gemm16_cmp.md
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print('$f', d['ms_per_step'], round(d['value']), {n:k[n] for n in list(k)[:8]})"; done
