set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/conv0_partner4.log
cp s3prl_amd/libs3enc.so /tmp/libs3enc_product.so
cp s3prl_amd/csrc/build/dbg/libs3enc_O.so s3prl_amd/libs3enc.so
echo "== lib O (32x32x8 twice in gemm.hip)" | tee -a $L
for gv in 1 3; do
timeout 300 python tools/conv0_partner_probe.py --partners fc1 qkv --tune gemm16_big=0 --tune gemm_variant=$gv 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee -a $L
done
cp /tmp/libs3enc_product.so s3prl_amd/libs3enc.so
