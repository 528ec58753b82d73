set -u
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r06/gputests_full.log 2>&1
tail -4 gpurun_out/r06/gputests_full.log
bash tools/round_profiles.sh r06 > gpurun_out/r06/round_profiles.log 2>&1
tail -5 gpurun_out/r06/round_profiles.log
