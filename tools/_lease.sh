set -u
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r06/gputests_final.log 2>&1
tail -6 gpurun_out/r06/gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/r06/bench_default_final.json 2> gpurun_out/r06/bench_default_final.err
cut -c1-400 gpurun_out/r06/bench_default_final.json
