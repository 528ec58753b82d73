set -u
mkdir -p gpurun_out/r06
tools/micro/gemm16_lab cmp 7 20007 50007 100007 > gpurun_out/r06/gemm16_lab_skew.md 2>&1
cat gpurun_out/r06/gemm16_lab_skew.md
