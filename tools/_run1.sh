set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150 ) > gpurun_out/r02a/pytest.log 2>&1
tail -5 gpurun_out/r02a/pytest.log
timeout 600 python bench.py --steps 30 --warmup 3 > gpurun_out/r02a/bench_fp32.json 2> gpurun_out/r02a/bench_fp32.err
tail -c 600 gpurun_out/r02a/bench_fp32.json; tail -3 gpurun_out/r02a/bench_fp32.err
timeout 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 --no-profile > gpurun_out/r02a/bench_gloo2.json 2> gpurun_out/r02a/bench_gloo2.err
tail -c 400 gpurun_out/r02a/bench_gloo2.json; tail -3 gpurun_out/r02a/bench_gloo2.err
