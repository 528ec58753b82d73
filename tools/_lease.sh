set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/conv0_partner3.log
cp s3prl_amd/libs3enc.so /tmp/libs3enc_product.so
for v in ${VARIANTS:-M}; do
  cp s3prl_amd/csrc/build/dbg/libs3enc_$v.so s3prl_amd/libs3enc.so
  echo "== lib $v" | tee -a $L
  timeout 300 python tools/conv0_partner_probe.py --partners fc1 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $L
  timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" 2>&1 | tail -2 | tee -a $L
done
cp /tmp/libs3enc_product.so s3prl_amd/libs3enc.so
