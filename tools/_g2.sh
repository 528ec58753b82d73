set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( timeout 1200 python -m pytest tests/test_encoder_gpu.py -x -q -k "fp16x2 or pretrained_like or 16bit" 2>&1 | tail -15 ) > gpurun_out/r05/t3.log 2>&1
timeout 300 tools/micro/gemm16_loop_probe random > gpurun_out/r05/loop_probe_random.md 2>&1
timeout 300 tools/micro/gemm16_loop_probe > gpurun_out/r05/loop_probe_const.md 2>&1
for cfg in "hubert_base 32" "hubert_large 32" ; do
  set -- $cfg
  timeout 300 python bench.py --model $1 --batch $2 --dtype fp16x2 --steps 30 --warmup 8 --no-cpu-baseline --no-other-modes 2>/dev/null | tail -1 > gpurun_out/r05/bench2_$1_fp16x2.json
done
timeout 300 python bench.py --model wavlm_large --batch 32 --mixed --dtype fp16x2 --steps 20 --warmup 6 --no-cpu-baseline --no-other-modes 2>/dev/null | tail -1 > gpurun_out/r05/bench2_wavlm_large_mixed_fp16x2.json
timeout 900 python tools/fp16_cliff.py hubert_large wavlm_large > gpurun_out/r05/fp16_cliff2.md 2> gpurun_out/r05/fp16_cliff2.err
timeout 900 python tools/parity_table.py > gpurun_out/r05/parity.md 2> gpurun_out/r05/parity.err
cat gpurun_out/r05/t3.log | tail -5
