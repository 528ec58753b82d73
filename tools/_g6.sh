set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r05/t_all2.log 2>&1
tail -3 gpurun_out/r05/t_all2.log
timeout 300 python tools/occupy_check.py > gpurun_out/r05/occupy_check.md 2>&1
PROFILE_ONLY="fp32 bf16 fp16x2" timeout 2400 bash tools/round_profiles.sh r05 > gpurun_out/r05/round_profiles.log 2>&1
ls gpurun_out/r05 | wc -l
