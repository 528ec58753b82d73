set -u
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x > gpurun_out/r06/gputests_full_raw.log 2>&1 ) 2> gpurun_out/r06/gputests_time.log
{ grep -E "passed|failed|error" gpurun_out/r06/gputests_full_raw.log | tail -3; cat gpurun_out/r06/gputests_time.log; } > gpurun_out/r06/gputests_final.log
cat gpurun_out/r06/gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r06/bench_default_final.json 2> gpurun_out/r06/bench_default_final.err
cut -c1-400 gpurun_out/r06/bench_default_final.json
