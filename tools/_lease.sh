set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
Q="--no-cpu-baseline --no-other-modes"
for cfg in "fp32 gemm hubert_base fp32" "bf16 gemm16 hubert_base bf16" "fp16x2 gemm16 hubert_base fp16x2"; do
  set -- $cfg
  name=$1; sub=$2; model=$3; dtype=$4
  ( time PMC_GROUPS="fetch write" tools/pmc.sh r06d_$name python bench.py --model $model --dtype $dtype --batch 32 --secs 10 --steps 2 --warmup 1 $Q --no-parity ) > /dev/null 2> gpurun_out/r06d/pmc_time_$name.log
  python tools/pmc_to_traffic.py gpurun_out/pmc_r06d_$name $model $dtype 32 10 profiles/traffic.json $sub > gpurun_out/r06d/traffic_$name.json 2> gpurun_out/r06d/traffic_$name.err
  cp gpurun_out/pmc_r06d_$name.md gpurun_out/r06d/pmc_$name.md 2>/dev/null
  rm -rf gpurun_out/pmc_r06d_$name gpurun_out/pmc_r06d_$name.md
  tail -3 gpurun_out/r06d/pmc_time_$name.log | head -1
done
cp profiles/traffic.json gpurun_out/r06d/traffic.json
timeout 900 python bench.py > gpurun_out/r06d/bench_default_final.json 2> gpurun_out/r06d/bench_default_final.err
python -c "
import json; d=json.loads(open('gpurun_out/r06d/bench_default_final.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['clock_ghz'], d['roofline'])"
for d in bf16 fp16x2; do python bench.py --dtype $d $Q --steps 100 --warmup 3 > gpurun_out/r06d/bench_$d.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06d/bench_$d.json').read().strip().splitlines()[-1]); print('$d', d['ms_per_step'], d['clock_ghz'], d['roofline'].get('traffic'))"; done
timeout 600 python tools/conv0_partner_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d/conv0_partner.log | cut -c1-400
