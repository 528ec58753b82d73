set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/fences2.log
: > $L
cnt() { grep "concurrent\|second run" | python -c "
import sys, json
print([json.loads(l)['differing (layer, utterance) pairs'] for l in sys.stdin])"; }
probe() { timeout 600 python tools/two_stream_probe.py --dtype bf16 --splits 1 4 8 --steps 5 --diagnose --tune forward_chain=0 2>&1 | cnt; }
cp s3prl_amd/libs3enc.so /tmp/libs3enc_product.so
for v in product both2 product both2 both2; do
  if [ $v = product ]; then cp /tmp/libs3enc_product.so s3prl_amd/libs3enc.so; else cp s3prl_amd/csrc/build/dbg/libs3enc_$v.so s3prl_amd/libs3enc.so; fi
  echo "== lib $v" | tee -a $L
  probe 2>&1 | tee -a $L
done
cp /tmp/libs3enc_product.so s3prl_amd/libs3enc.so
