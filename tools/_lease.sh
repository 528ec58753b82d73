set -u
export TMPDIR=/tmp
cnt() { grep "concurrent\|second run" | python -c "
import sys, json
print([json.loads(l)['differing (layer, utterance) pairs'] for l in sys.stdin])"; }
cp s3prl_amd/libs3enc.so /tmp/base.so
for v in base sc1 sys base sc1; do
  if [ $v = base ]; then cp /tmp/base.so s3prl_amd/libs3enc.so; else cp gpurun_variants/libs3enc_$v.so s3prl_amd/libs3enc.so; fi
  echo "== $v bf16 (chain off)"; timeout 600 python tools/two_stream_probe.py --dtype bf16 --splits 1 4 8 --steps 5 --diagnose --tune forward_chain=0 2>&1 | cnt
done
for v in base sc1; do
  if [ $v = base ]; then cp /tmp/base.so s3prl_amd/libs3enc.so; else cp gpurun_variants/libs3enc_$v.so s3prl_amd/libs3enc.so; fi
  python bench.py --no-cpu-baseline --no-other-modes --no-parity --dtype bf16 --steps 300 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v bf16', d['ms_per_step'], d['clock_ghz'])"
done
cp /tmp/base.so s3prl_amd/libs3enc.so
