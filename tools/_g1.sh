set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -k "status or strict or poll or wavlm_large_15s_pl" 2>&1 | tail -15 ) > gpurun_out/r05/t1.log 2>&1
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "persistent_tile_loop" 2>&1 | tail -8 ) > gpurun_out/r05/t2.log 2>&1
timeout 300 tools/micro/gemm16_loop_probe > gpurun_out/r05/loop_probe.md 2>&1
timeout 300 tools/micro/gemm16_lab cmp 7 107 207 307 > gpurun_out/r05/lab_cmp.md 2>&1
timeout 300 tools/micro/gemm16_lab cmp8 7 107 207 307 > gpurun_out/r05/lab_cmp8.md 2>&1
timeout 300 tools/micro/gemm16_lab cmpx 7 107 207 307 > gpurun_out/r05/lab_cmpx.md 2>&1
for pp in 0 1; do
  for dt in bf16 fp16x2; do
    timeout 300 python bench.py --dtype $dt --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-other-modes --tune gemm16_pp=$pp 2>/dev/null | tail -1 > gpurun_out/r05/bench_${dt}_pp$pp.json
  done
done
timeout 600 python tools/fp16_cliff.py hubert_large wavlm_large > gpurun_out/r05/fp16_cliff.md 2> gpurun_out/r05/fp16_cliff.err
tail -3 gpurun_out/r05/t1.log gpurun_out/r05/t2.log
cat gpurun_out/r05/lab_cmp.md
