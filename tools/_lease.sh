set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/after_v255.log
: > $L
cnt() { grep "concurrent\|second run" | python -c "
import sys, json
print([json.loads(l)['differing (layer, utterance) pairs'] for l in sys.stdin])"; }
timeout 300 python tools/conv0_partner_probe.py --partners fc1 qkv all 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee -a $L
for dt in bf16 fp16 fp32x3 fp16x2; do for rep in 1 2; do
  echo "== hubert_base $dt forward_chain=0" | tee -a $L
  timeout 600 python tools/two_stream_probe.py --dtype $dt --splits 1 4 8 --steps 5 --diagnose --tune forward_chain=0 2>&1 | cnt | tee -a $L
done; done
for m in hubert_large wavlm_large; do for rep in 1 2; do
  echo "== $m bf16 forward_chain=0" | tee -a $L
  timeout 900 python tools/two_stream_probe.py --model $m --dtype bf16 --batch 16 --splits 1 4 8 --steps 3 --diagnose --tune forward_chain=0 2>&1 | cnt | tee -a $L
done; done
( time timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r06d/gputests_full_raw.log 2>&1 ) 2> gpurun_out/r06d/gputests_time.log
{ grep -E "passed|failed|error" gpurun_out/r06d/gputests_full_raw.log | tail -3; cat gpurun_out/r06d/gputests_time.log; } > gpurun_out/r06d/gputests_final.log
cat gpurun_out/r06d/gputests_final.log | tee -a $L
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $L
Q="--no-cpu-baseline --no-other-modes"
for cfg in "fp32 gemm hubert_base fp32" "bf16 gemm16 hubert_base bf16" "fp16x2 gemm16 hubert_base fp16x2"; do
  set -- $cfg
  name=$1; sub=$2; model=$3; dtype=$4
  PMC_GROUPS="fetch write" tools/pmc.sh r06d_$name python bench.py --model $model --dtype $dtype --batch 32 --secs 10 --steps 2 --warmup 1 $Q --no-parity > /dev/null 2>&1
  python tools/pmc_to_traffic.py gpurun_out/pmc_r06d_$name $model $dtype 32 10 profiles/traffic.json $sub > gpurun_out/r06d/traffic_$name.json 2> gpurun_out/r06d/traffic_$name.err
  cp gpurun_out/pmc_r06d_$name.md gpurun_out/r06d/pmc_$name.md 2>/dev/null
  rm -rf gpurun_out/pmc_r06d_$name gpurun_out/pmc_r06d_$name.md
done
cp profiles/traffic.json gpurun_out/r06d/traffic.json
timeout 900 python bench.py > gpurun_out/r06d/bench_default_final.json 2> gpurun_out/r06d/bench_default_final.err
python -c "
import json; d=json.loads(open('gpurun_out/r06d/bench_default_final.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['clock_ghz'], d['roofline']['frac'], d['roofline']['traffic'], {k: v['ms_per_step'] for k, v in d['other_modes'].items()})" | tee -a $L
for d in bf16 fp16x2; do python bench.py --dtype $d $Q --steps 100 --warmup 3 > gpurun_out/r06d/bench_$d.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06d/bench_$d.json').read().strip().splitlines()[-1]); print('$d', d['ms_per_step'], d['clock_ghz'], d['roofline'].get('frac'), d['roofline'].get('traffic'))" | tee -a $L; done
