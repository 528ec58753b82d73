set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
( time timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r06d/gputests_full_raw.log 2>&1 ) 2> gpurun_out/r06d/gputests_time.log
{ grep -E "passed|failed|error" gpurun_out/r06d/gputests_full_raw.log | tail -3; cat gpurun_out/r06d/gputests_time.log; } > gpurun_out/r06d/gputests_final.log
cat gpurun_out/r06d/gputests_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r06d/smoke.log
timeout 900 python bench.py > gpurun_out/r06d/bench_default_final.json 2> gpurun_out/r06d/bench_default_final.err
cut -c1-300 gpurun_out/r06d/bench_default_final.json
