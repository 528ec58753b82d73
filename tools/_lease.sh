set -u
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > gpurun_out/r06/gputests_lease1.log 2>&1
tools/micro/attn_lab > gpurun_out/r06/attn_lab.md 2>&1
python tools/parity_seeds.py > gpurun_out/r06/parity_seeds.md 2> gpurun_out/r06/parity_seeds.err
python bench.py > gpurun_out/r06/bench_default_lease1.json 2> gpurun_out/r06/bench_default_lease1.err
tail -3 gpurun_out/r06/gputests_lease1.log; head -40 gpurun_out/r06/attn_lab.md; head -20 gpurun_out/r06/parity_seeds.md; cut -c1-600 gpurun_out/r06/bench_default_lease1.json
