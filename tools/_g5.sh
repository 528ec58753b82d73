set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
Q="--no-cpu-baseline --no-parity --no-other-modes --no-profile"
for dt in bf16 fp32; do
  steps=60; [ $dt = fp32 ] && steps=15
  for cfg in "0 0" "8 0" "16 0" "32 0" "16 16" "32 32" "0 16"; do
    set -- $cfg
    timeout 300 python bench.py --dtype $dt --steps $steps --warmup 5 $Q --steal-cus $1 --tune reserve_cus=$2 2>/dev/null | tail -1 > gpurun_out/r05/steal_${dt}_s$1_r$2.json
  done
done
( timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r05/t_all.log 2>&1
tail -5 gpurun_out/r05/t_all.log
