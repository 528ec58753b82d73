set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "mx_second_term or persistent_tile_loop or f16x2" 2>&1 | tail -25 ) > gpurun_out/r05/t4.log 2>&1
( timeout 1200 python -m pytest tests/test_encoder_gpu.py -x -q -k "mx_second_term or fp16x2" 2>&1 | tail -25 ) > gpurun_out/r05/t5.log 2>&1
timeout 300 tools/micro/gemm16_lab cmpx 7 1007 > gpurun_out/r05/lab_cmpx_mx.md 2>&1
for mx in 1 0; do
  timeout 300 python bench.py --model hubert_base --batch 32 --dtype fp16x2 --steps 40 --warmup 10 --no-cpu-baseline --no-other-modes --tune gemm16_mx=$mx 2>/dev/null | tail -1 > gpurun_out/r05/bench3_hubert_base_fp16x2_mx$mx.json
  timeout 300 python bench.py --model hubert_large --batch 32 --dtype fp16x2 --steps 20 --warmup 6 --no-cpu-baseline --no-other-modes --tune gemm16_mx=$mx 2>/dev/null | tail -1 > gpurun_out/r05/bench3_hubert_large_fp16x2_mx$mx.json
done
tail -4 gpurun_out/r05/t4.log; tail -4 gpurun_out/r05/t5.log; cat gpurun_out/r05/lab_cmpx_mx.md
