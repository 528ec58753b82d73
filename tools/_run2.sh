set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm16_big or conv0 or proj or taps" 2>&1 | tail -30 ) > gpurun_out/r02b/pytest.log 2>&1
( timeout 300 python -m pytest tests/test_encoder_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "taps" 2>&1 | tail -8 ) >> gpurun_out/r02b/pytest.log 2>&1
tail -12 gpurun_out/r02b/pytest.log
timeout 900 python tools/gemm_bench.py bf16 --variants 1,4,6,7 --rounds 3 > gpurun_out/r02b/gemm_bench.log 2>&1
cat gpurun_out/r02b/gemm_bench.log
