set -u
export TMPDIR=/tmp
cnt() { grep "concurrent\|second run" | python -c "
import sys, json
print([json.loads(l)['differing (layer, utterance) pairs'] for l in sys.stdin])"; }
for v in 0 1 2 0 1; do
echo "== evfence $v"; S3ENC_DEBUG_EVFENCE=$v timeout 600 python tools/two_stream_probe.py --dtype bf16 --splits 1 4 8 --steps 5 --diagnose --tune forward_chain=0 2>&1 | cnt
done
