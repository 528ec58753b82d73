set -u
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x > gpurun_out/r06/gputests_full_raw.log 2>&1 ) 2> gpurun_out/r06/gputests_time.log
{ grep -E "passed|failed|error" gpurun_out/r06/gputests_full_raw.log | tail -3; cat gpurun_out/r06/gputests_time.log; } > gpurun_out/r06/gputests_final.log
cat gpurun_out/r06/gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time bash tools/round_profiles.sh r06 ) 2>&1 | tail -5
python tools/parity_seeds.py > gpurun_out/r06/parity_seeds.md 2> gpurun_out/r06/parity_seeds.err
tail -12 gpurun_out/r06/parity_seeds.md
cut -c1-300 gpurun_out/r06/bench_fp32.json
