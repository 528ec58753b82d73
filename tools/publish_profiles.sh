#!/bin/bash
# Copy the artefacts tools/round_profiles.sh merged back (gpurun_out/<tag>/) into profiles/<tag>_*, the tracked copies.
# usage: tools/publish_profiles.sh r03
tag=${1:-r04}
src=gpurun_out/$tag
[ -d $src ] || { echo "no $src" >&2; exit 1; }
for f in $src/bench_*.json; do [ -s $f ] && grep "^{" $f | tail -1 > profiles/${tag}_$(basename $f); done
for f in $src/kernel_stats_*.md; do
  [ -s $f ] || continue
  name=$(basename $f .md); name=${name#kernel_stats_}
  { echo "# Round ${tag#r} — rocprofv3 --kernel-trace --stats, configuration \`$name\` (1x MI355X)"; echo
    echo "Command: the bench.py line of \`profile_cfg $name\` in tools/round_profiles.sh with \`--steps 5 --warmup 1 --no-parity\`;"
    echo "summary by tools/rocprof_summary.py from the rocpd database.  bench.py's own HIP-event figure for the dominant kernel in the"
    echo "un-profiled run is in profiles/${tag}_bench_$name.json (roofline.avg_launch_ms)."; echo
    cat $f; } > profiles/${tag}_kernel_stats_$name.md
done
for f in $src/pmc_*.md; do [ -s $f ] && cp $f profiles/${tag}_$(basename $f); done
for f in gemm32_lab_fp32 gemm32_lab_x3 attn_lab attn_lab_product gemm16_lab gemm16_lab_persistent gemm16_lab_modes gemm16_lab_modes_fp16x2 gemm16_lab_shared_panels parity; do [ -s $src/$f.md ] && cp $src/$f.md profiles/${tag}_$f.md; done
[ -s $src/traffic.json ] && cp $src/traffic.json profiles/traffic.json
# stamp the records measured on THIS tree's kernels with the commit they belong to (the GPU box has no .git; bench.py matches on
# csrc_sha16 and quotes the commit)
python - <<PYEOF
import json, subprocess, sys
sys.path.insert(0, '.')
import bench
have = bench.csrc_sha16()
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(['git', 'status', '--porcelain', 's3prl_amd/csrc'], capture_output=True, text=True).stdout.strip())
recs = json.load(open('profiles/traffic.json'))
for r in recs:
    if r.get('csrc_sha16') == have and not r.get('commit'):
        r['commit'] = head + ('+uncommitted csrc changes' if dirty else '')
json.dump(recs, open('profiles/traffic.json', 'w'), indent=1)
print('traffic.json:', sum(r.get('csrc_sha16') == have for r in recs), 'of', len(recs), 'records match this tree (csrc', have + ')')
PYEOF
python - <<EOF
import json,glob
for f in sorted(glob.glob('profiles/${tag}_bench_*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(f, 'unreadable', e); continue
    r=d.get('roofline') or {}
    print(f.split('/')[-1][10:-5], d['ms_per_step'], d['value'], d.get('path_tflops'), r.get('achieved'), r.get('frac'), r.get('avg_launch_ms'), r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
EOF
