set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/stream_order.log
for a in "1 300 64 1" "4 300 64 1" "8 300 64 1" "8 1000 16 1" "8 300 128 1" "16 300 32 1" "8 2000 8 1"; do
  timeout 120 tools/micro/stream_order_probe $a 2>&1 | tee -a $L
done
