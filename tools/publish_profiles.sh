#!/bin/bash
# Copy the artefacts tools/round_profiles.sh produced (gpurun_out/<tag>, gpurun_out/pmc_<tag>_*) into profiles/.
tag=${1:-r01}
for f in gpurun_out/$tag/bench_*.json; do cp $f profiles/${tag}_$(basename $f); done
for dt in fp32 bf16; do
  extra=""; [ $dt = bf16 ] && extra=" --dtype bf16"
  { echo "# Round ${tag#r} — rocprofv3 --kernel-trace --stats of \`python bench.py$extra --steps 5 --warmup 1 --no-cpu-baseline\` (HuBERT-base 32x10 s, 1x MI355X)"; echo;
    echo "9 forwards (1 warm-up + 5 timed + 3 breakdown steps); summary produced by tools/rocprof_summary.py from the rocpd database (tools/round_profiles.sh)."
    echo "bench.py's own HIP-event figure for the dominant kernel in the un-profiled run is in profiles/${tag}_bench_$dt.json (roofline.avg_launch_ms)."; echo;
    cat gpurun_out/$tag/kernel_stats_$dt.md; } > profiles/${tag}_kernel_stats_$dt.md
  cp gpurun_out/pmc_${tag}_$dt.md profiles/${tag}_pmc_bench_$dt.md
done
python tools/pmc_to_traffic.py gpurun_out/pmc_${tag}_fp32 hubert_base fp32 32 10 profiles/traffic.json > /dev/null
python tools/pmc_to_traffic.py gpurun_out/pmc_${tag}_bf16 hubert_base bf16 32 10 profiles/traffic.json gemm16 > /dev/null
python - <<EOF
import json,glob
for f in sorted(glob.glob('profiles/${tag}_bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1][10:-5], d['ms_per_step'], d['value'], d['path_tflops'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'), d.get('cpu_baseline',{}).get('value'))
EOF
