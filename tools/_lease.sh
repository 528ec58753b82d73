set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_encoder_gpu.py -q -x -k "four_handles or race_screen or one_utterance_alone" 2>&1 | tail -4
cnt() { grep "concurrent\|second run" | python -c "
import sys, json
print([json.loads(l)['differing (layer, utterance) pairs'] for l in sys.stdin])"; }
for d in bf16 fp16 fp16x2; do
echo "== $d chain on (default)"; timeout 600 python tools/two_stream_probe.py --dtype $d --splits 1 4 8 --steps 5 --diagnose 2>&1 | tee gpurun_out/r06c/chain_on_$d.log | cnt
echo "== $d chain off"; timeout 600 python tools/two_stream_probe.py --dtype $d --splits 1 4 8 --steps 5 --diagnose --tune forward_chain=0 2>&1 | tee gpurun_out/r06c/chain_off_$d.log | cnt
done
grep ms_per_step gpurun_out/r06c/chain_on_bf16.log gpurun_out/r06c/chain_off_bf16.log | cut -c1-200
Q="--no-cpu-baseline --no-other-modes --no-parity"
for c in 1 0 1 0; do
python bench.py $Q --dtype bf16 --steps 300 --warmup 5 --tune forward_chain=$c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bf16 forward_chain=$c', d['ms_per_step'], d['clock_ghz'])"
done
for c in 1 0; do
python bench.py $Q --steps 60 --warmup 3 --tune forward_chain=$c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fp32 forward_chain=$c', d['ms_per_step'], d['clock_ghz'])"
done
