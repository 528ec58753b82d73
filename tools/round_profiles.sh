#!/bin/bash
# Re-create the measured artefacts of a round on the GPU box (run through gpurun from the repo root):
#   per BASELINE config: PMC passes (-> profiles/traffic.json record) -> bench (its JSON then carries roofline.traffic) ->
#   rocprofv3 kernel-trace summary; the operand modes of the headline workload; the micro labs; the parity table.
# Raw rocprof output stays on the box (only the summaries are merged back: gpurun_out is capped at 64 MiB).
# usage: tools/round_profiles.sh r05        (most important artefacts first: a cut-off run still leaves them)
#        PROFILE_ONLY="fp32 bf16" tools/round_profiles.sh r04   (PMC + rocprof only for the named configurations, bench lines for all)
set -u
tag=${1:-r06}
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
# 2. the in-tolerance throughput mode of the same workload, then BASELINE configs[2] / [3] / [4]: every config's fp16x2 line
#    (parity < 1e-3) first, its bf16 line (the dtype BASELINE names; parity 1e-2) second
profile_cfg fp16x2 gemm16 hubert_base fp16x2 32 10 "" 200
profile_cfg bf16 gemm16 hubert_base bf16 32 10 "" 300
profile_cfg cfg3_hubert_large_fp16x2 gemm16 hubert_large fp16x2 32 10 "" 40
profile_cfg cfg4_wavlm_large_mixed_fp16x2 gemm16 wavlm_large fp16x2 32 15 "--mixed" 30
PROFILE_ONLY_SAVE=${PROFILE_ONLY:-}
python bench.py --model hubert_base --dtype fp16x2 --batch 64 $Q --steps 100 --warmup 3 > $out/bench_cfg2_hubert_base_b64_fp16x2.json 2>/dev/null
python bench.py --model hubert_base --dtype bf16 --batch 64 $Q --steps 100 --warmup 3 > $out/bench_cfg2_hubert_base_b64_bf16.json 2>/dev/null
python bench.py --model hubert_large --dtype bf16 $Q --steps 60 --warmup 2 > $out/bench_cfg3_hubert_large_bf16.json 2>/dev/null
python bench.py --model wavlm_large --dtype bf16 --secs 15 --mixed $Q --steps 40 --warmup 2 > $out/bench_cfg4_wavlm_large_mixed_bf16.json 2>/dev/null
# 3. the other operand modes of the headline workload and of configs[1], [3], [4]
python bench.py --dtype fp32x3 $Q > $out/bench_fp32x3.json 2>/dev/null
python bench.py --dtype fp16 $Q --steps 200 > $out/bench_fp16.json 2>/dev/null
python bench.py --model wav2vec2_base $Q --steps 30 --warmup 2 > $out/bench_cfg1_wav2vec2_base_fp32.json 2>/dev/null
for d in fp32x3 fp32; do
  python bench.py --model hubert_large --dtype $d $Q --steps 12 --warmup 1 > $out/bench_cfg3_hubert_large_$d.json 2>/dev/null
  python bench.py --model wavlm_large --dtype $d --secs 15 --mixed $Q --steps 8 --warmup 1 > $out/bench_cfg4_wavlm_large_mixed_$d.json 2>/dev/null
done
# 4. micro labs (standalone binaries, seconds each)
for b in gemm32_lab attn_lab attn_lab_product attn_lab_x0 gemm16_lab gemm16_loop_probe mx_probe; do [ -x tools/micro/$b ] || echo "tools/micro/$b is not built" >&2; done
tools/micro/gemm16_lab cmp 7 8 9 10 > $out/gemm16_lab_modes.md 2>&1        # persistent loop / + store overlap / row-per-lane epilogue / both
tools/micro/gemm16_lab cmpx 7 9 1007 > $out/gemm16_lab_modes_fp16x2.md 2>&1   # (1007: the MX second weight term, forced on every shape)
tools/micro/gemm16_lab cmp8 7 > $out/gemm16_lab_shared_panels.md 2>&1       # every operand L2-resident: what the memory side costs
tools/micro/attn_lab > $out/attn_lab.md 2>&1
{ echo '## the kernels of the library (no probes compiled in)'; QUICK=1 SKIP_F32=1 tools/micro/attn_lab_product; echo; echo '## S3_ATTN_EXP = 0 (the kernels before the second session of round 6)'; QUICK=1 SKIP_F32=1 tools/micro/attn_lab_x0; } > $out/attn_lab_product.md 2>&1
tools/micro/mx_probe > $out/mx_probe.md 2>&1
# 5. parity of every mode against the reference's own outputs (synthetic and pretrained-like statistics)
python tools/parity_table.py > $out/parity.md 2> $out/parity.err
# 6. a sibling model and the N-rank path, functionally (two ranks share this box's single GPU over a gloo rendezvous)
python bench.py --model multires_hubert_base $Q --steps 40 --warmup 3 > $out/bench_multires_hubert_base_fp32.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --gather layers --steps 5 --warmup 1 --no-profile > $out/bench_2rank_gloo_layers.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --dtype fp16x2 --exchange-algo direct --steps 5 --warmup 1 --no-profile > $out/bench_2rank_gloo_fp16x2_direct.json 2>/dev/null
ls $out
