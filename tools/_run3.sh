set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm16_big" 2>&1 | tail -15 ) > gpurun_out/r02c/pytest.log 2>&1
tail -5 gpurun_out/r02c/pytest.log
timeout 900 python tools/gemm_bench.py bf16 --variants 1,7,8,9,101,102,104,103 --rounds 3 --shapes conv1,qkv,fc1,fc2,sq4k,sq8k,L_fc1,L_fc2 > gpurun_out/r02c/gemm_bench.log 2>&1
cat gpurun_out/r02c/gemm_bench.log
