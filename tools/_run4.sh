set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_x3 or gemm16_big" 2>&1 | tail -15 ) > gpurun_out/r02d/pytest.log 2>&1
tail -5 gpurun_out/r02d/pytest.log
timeout 900 python tools/gemm_bench.py fp32x3 --rounds 3 --shapes conv1,qkv,out_proj,fc1,fc2,sq4k,L_fc1,L_fc2 > gpurun_out/r02d/gemm_bench_x3.log 2>&1
cat gpurun_out/r02d/gemm_bench_x3.log
