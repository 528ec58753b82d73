set -u
mkdir -p gpurun_out/r06
( time python -m pytest tests/test_ops_gpu.py -m gpu -q -k "persistent or attention" 2>&1 | tail -8 ) > gpurun_out/r06/gputests_lease9.log 2>&1
cat gpurun_out/r06/gputests_lease9.log
python -c "import __graft_entry__ as g; g.smoke()"
