#!/bin/bash
# Re-create the measured artefacts of a round on the GPU box (run through gpurun from the repo root):
#   per BASELINE config: PMC passes (-> profiles/traffic.json record) -> bench (its JSON then carries roofline.traffic) ->
#   rocprofv3 kernel-trace summary; the operand modes of the headline workload; the micro labs; the parity table.
# Raw rocprof output stays on the box (only the summaries are merged back: gpurun_out is capped at 64 MiB).
# usage: tools/round_profiles.sh r03        (most important artefacts first: a cut-off run still leaves them)
#        PROFILE_ONLY="fp32 bf16" tools/round_profiles.sh r03   (PMC + rocprof only for the named configurations, bench lines for all)
set -u
tag=${1:-r03}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
Q="--no-cpu-baseline --no-other-modes"

# one configuration: name, kernel substring of its dominant GEMM, model, dtype, batch, secs, extra bench flags, steps
profile_cfg() {
  local name=$1 sub=$2 model=$3 dtype=$4 batch=$5 secs=$6 extra=$7 steps=$8
  local args="--model $model --dtype $dtype --batch $batch --secs $secs $extra"
  if [ -n "${PROFILE_ONLY:-}" ] && ! echo " $PROFILE_ONLY " | grep -q " $name "; then
    # bench line only: its roofline.traffic comes from the record an earlier full run left in profiles/traffic.json
    python bench.py $args --steps $steps --warmup 2 $Q > $out/bench_$name.json 2> $out/bench_$name.err
    return
  fi
  PMC_GROUPS="sq1 sq2 tcc fetch write" tools/pmc.sh ${tag}_$name python bench.py $args --steps 2 --warmup 1 $Q --no-parity > /dev/null 2>&1
  python tools/pmc_to_traffic.py gpurun_out/pmc_${tag}_$name $model $dtype $batch $secs profiles/traffic.json $sub > $out/traffic_$name.json 2>/dev/null
  cp gpurun_out/pmc_${tag}_$name.md $out/pmc_$name.md 2>/dev/null
  rm -rf gpurun_out/pmc_${tag}_$name gpurun_out/pmc_${tag}_$name.md
  python bench.py $args --steps $steps --warmup 2 $Q > $out/bench_$name.json 2> $out/bench_$name.err
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python bench.py $args --steps 5 --warmup 1 $Q --no-parity > /dev/null 2>&1
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" $out/kernel_stats_$name.md > /dev/null
  rm -rf /tmp/prof_$name
  cp profiles/traffic.json $out/traffic.json
}

# 1. the metric's workload (HuBERT-base 32 x 10 s, fp32): PMC + traffic first so that the default bench line carries it
profile_cfg fp32 gemm hubert_base fp32 32 10 "" 60
python bench.py > $out/bench_fp32.json 2> $out/bench_fp32.err          # the default run: >= 5 s timed, CPU baseline, other modes
# 2. BASELINE configs[2] / [3] / [4] exactly as named
profile_cfg cfg2_hubert_base_b64_bf16 gemm16 hubert_base bf16 64 10 "" 100
profile_cfg cfg3_hubert_large_bf16 gemm16 hubert_large bf16 32 10 "" 60
profile_cfg cfg4_wavlm_large_mixed_bf16 gemm16 wavlm_large bf16 32 15 "--mixed" 40
# 3. the operand modes of the headline workload
profile_cfg bf16 gemm16 hubert_base bf16 32 10 "" 300
python bench.py --dtype fp16x2 $Q > $out/bench_fp16x2.json 2>/dev/null
python bench.py --dtype fp32x3 $Q > $out/bench_fp32x3.json 2>/dev/null
python bench.py --dtype fp16 $Q --steps 200 > $out/bench_fp16.json 2>/dev/null
# 4. the other modes of configs[1], [3], [4]
python bench.py --model wav2vec2_base $Q --steps 30 --warmup 2 > $out/bench_cfg1_wav2vec2_base_fp32.json 2>/dev/null
for d in fp16x2 fp32x3 fp32; do
  python bench.py --model hubert_large --dtype $d $Q --steps 12 --warmup 1 > $out/bench_cfg3_hubert_large_$d.json 2>/dev/null
  python bench.py --model wavlm_large --dtype $d --secs 15 --mixed $Q --steps 8 --warmup 1 > $out/bench_cfg4_wavlm_large_mixed_$d.json 2>/dev/null
done
# 5. micro labs (standalone binaries, seconds each)
for b in gemm32_lab attn_lab gemm16_lab gemm16_loop_probe; do [ -x tools/micro/$b ] || echo "tools/micro/$b is not built" >&2; done
tools/micro/gemm32_lab 3 > $out/gemm32_lab_fp32.md 2>&1
tools/micro/gemm32_lab 3 - x3 > $out/gemm32_lab_x3.md 2>&1
tools/micro/attn_lab > $out/attn_lab.md 2>&1
tools/micro/gemm16_lab > $out/gemm16_lab.md 2>&1
tools/micro/gemm16_lab cmp 1 7 > $out/gemm16_lab_persistent.md 2>&1      # one tile per workgroup against the persistent tile loop
tools/micro/gemm16_loop_probe > $out/gemm16_loop_probe.md 2>&1
# 6. parity of every mode against the reference's own outputs
python tools/parity_table.py > $out/parity.md 2> $out/parity.err
# 7. a sibling model and the N-rank path, functionally (two ranks share this box's single GPU over a gloo rendezvous)
python bench.py --model multires_hubert_base $Q --steps 40 --warmup 3 > $out/bench_multires_hubert_base_fp32.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --gather layers --steps 5 --warmup 1 --no-profile > $out/bench_2rank_gloo_layers.json 2>/dev/null
ls $out
