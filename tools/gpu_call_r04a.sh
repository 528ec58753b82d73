set -u
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent or gemm16_big_tiles or far_below" > $O/pytest_gemm16.log 2>&1
echo "pytest gemm16 rc=$?" >> $O/summary.txt
timeout 300 tools/micro/gemm16_lab cmp 7 8 9 10 > $O/gemm16_cmp.md 2>&1
timeout 300 tools/micro/gemm16_lab cmp8 7 8 9 10 > $O/gemm16_cmp_shared.md 2>&1
timeout 300 tools/micro/gemm16_lab cmpx 7 8 9 10 > $O/gemm16_cmp_fp16x2.md 2>&1
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_parallel_gpu.py tests/test_featurizer_gpu.py -x -q -m gpu -k "_pl or pretrained or cabi or legacy_featurizer_backprop" > $O/pytest_new.log 2>&1
echo "pytest new rc=$?" >> $O/summary.txt
timeout 900 python tools/parity_table.py hubert_base_pseudo hubert_base_pl wav2vec2_base_pl hubert_large_pl wavlm_large_pl hubert_base_10s_pl hubert_large_10s_pl tiny_hubert_pl tiny_wavlm_large_pl > $O/parity_pl.md 2> $O/parity_pl.err
Q="--no-cpu-baseline --no-other-modes --no-parity"
for m in 7 8 9 10; do
  timeout 200 python bench.py --dtype bf16 $Q --steps 100 --warmup 5 --tune gemm16_big=$m > $O/bench_bf16_m$m.json 2>/dev/null
  timeout 200 python bench.py --dtype fp16x2 $Q --steps 60 --warmup 5 --tune gemm16_big=$m > $O/bench_fp16x2_m$m.json 2>/dev/null
done
ls -la $O
tail -3 $O/pytest_gemm16.log $O/pytest_new.log
cat $O/gemm16_cmp.md
