set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/conv0_fast_key.log
cnt() { grep "concurrent\|second run" | python -c "
import sys, json
print([json.loads(l)['differing (layer, utterance) pairs'] for l in sys.stdin])"; }
for dt in fp32x3 fp16x2 fp32; do
for k in 1 0 0; do
  echo "== $dt conv0_fast=$k forward_chain=0" | tee -a $L
  timeout 600 python tools/two_stream_probe.py --dtype $dt --splits 1 4 8 --steps 5 --diagnose --tune forward_chain=0 --tune conv0_fast=$k 2>&1 | cnt | tee -a $L
done
done
for m in hubert_large wavlm_large; do
for k in 1 0 0; do
  echo "== $m bf16 conv0_fast=$k forward_chain=0" | tee -a $L
  timeout 900 python tools/two_stream_probe.py --model $m --dtype bf16 --batch 16 --splits 1 4 8 --steps 3 --diagnose --tune forward_chain=0 --tune conv0_fast=$k 2>&1 | cnt | tee -a $L
done
done
