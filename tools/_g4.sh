set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python tools/mx_mask_sweep.py > gpurun_out/r05/mx_mask.md 2> gpurun_out/r05/mx_mask.err
timeout 300 tools/micro/gemm16_lab cmpx 7 1007 > gpurun_out/r05/lab_cmpx_mx2.md 2>&1
for mx in 0 1 2 4 8 15; do
  timeout 300 python bench.py --model hubert_base --batch 32 --dtype fp16x2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-other-modes --tune gemm16_mx=$mx 2>/dev/null | tail -1 > gpurun_out/r05/bench4_hubert_base_mx$mx.json
done
for mx in 0 1 2 15 31; do
  timeout 300 python bench.py --model hubert_large --batch 32 --dtype fp16x2 --steps 20 --warmup 6 --no-cpu-baseline --no-parity --no-other-modes --tune gemm16_mx=$mx 2>/dev/null | tail -1 > gpurun_out/r05/bench4_hubert_large_mx$mx.json
done
cat gpurun_out/r05/mx_mask.md
