set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/conv0_partner2.log
: > $L
for t in "x=0" "gemm16_rows=0" "gemm16_big=6" "gemm16_big=1" "gemm16_big=9" "conv0_fast=0"; do
  if [ "$t" = "x=0" ]; then T=""; else T="--tune $t"; fi
  timeout 300 python tools/conv0_partner_probe.py --partners fc1 qkv $T 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $L
done
