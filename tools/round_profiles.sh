#!/bin/bash
# Re-create the measured artefacts of a round on the GPU box (run through gpurun from the repo root):
#   benches of the BASELINE configs (each with its parity leg), the rocprofv3 kernel-trace summary of the headline
#   command, the PMC passes, the parity table.   usage: tools/round_profiles.sh r02
set -u
tag=${1:-r02}
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_fp32.json 2> $out/bench_fp32.err
python bench.py --dtype bf16 --no-cpu-baseline > $out/bench_bf16.json 2>/dev/null
python bench.py --dtype fp16 --no-cpu-baseline --steps 200 > $out/bench_fp16.json 2>/dev/null
python bench.py --dtype fp32x3 --no-cpu-baseline > $out/bench_fp32x3.json 2>/dev/null
python bench.py --model wav2vec2_base --no-cpu-baseline --steps 30 --warmup 2 > $out/bench_cfg1_wav2vec2_base_fp32.json 2>/dev/null
python bench.py --model hubert_base --dtype bf16 --batch 64 --no-cpu-baseline --steps 100 --warmup 2 > $out/bench_cfg2_hubert_base_b64_bf16.json 2>/dev/null
python bench.py --model hubert_large --dtype bf16 --no-cpu-baseline --steps 60 --warmup 2 > $out/bench_cfg3_hubert_large_bf16.json 2>/dev/null
python bench.py --model hubert_large --dtype fp32x3 --no-cpu-baseline --steps 30 --warmup 2 > $out/bench_cfg3_hubert_large_fp32x3.json 2>/dev/null
python bench.py --model hubert_large --dtype fp32 --no-cpu-baseline --steps 10 --warmup 1 > $out/bench_cfg3_hubert_large_fp32.json 2>/dev/null
python bench.py --model wavlm_large --dtype bf16 --secs 15 --mixed --no-cpu-baseline --steps 40 --warmup 2 > $out/bench_cfg4_wavlm_large_mixed_bf16.json 2>/dev/null
python bench.py --model wavlm_large --dtype fp32x3 --secs 15 --mixed --no-cpu-baseline --steps 15 --warmup 1 > $out/bench_cfg4_wavlm_large_mixed_fp32x3.json 2>/dev/null
python bench.py --model wavlm_large --dtype fp32 --secs 15 --mixed --no-cpu-baseline --steps 6 --warmup 1 > $out/bench_cfg4_wavlm_large_mixed_fp32.json 2>/dev/null
# the N-rank path, functionally, on this box's single GPU (ranks share the device over a gloo rendezvous): every exchange mode
for g in layers featurized; do
  python bench.py --gpus 2 --backend gloo --gather $g --steps 5 --warmup 1 --no-profile > $out/bench_2rank_gloo_$g.json 2>/dev/null
done
python bench.py --gpus 2 --backend gloo --gather layers16 --dtype bf16 --steps 5 --warmup 1 --no-profile > $out/bench_2rank_gloo_layers16.json 2>/dev/null
for d in fp32 fp32x3 bf16; do
  python bench.py --model multires_hubert_base --dtype $d --no-cpu-baseline --steps 40 --warmup 3 > $out/bench_multires_hubert_base_$d.json 2>/dev/null
done
python tools/gemm_yardstick.py > $out/gemm_yardstick.md 2>/dev/null          # vendor BLAS beside the library's GEMMs (yardstick only)
for m in mfma_peak gemm_loop_probe; do                                        # matrix-pipe ceilings and the fp32 loop's ingredients
  [ -x tools/micro/$m ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/$m.hip -o tools/micro/$m 2>/dev/null
  tools/micro/$m > $out/$m.md 2>/dev/null
done
python tools/parity_table.py > $out/parity.md 2> $out/parity.err
# rocprofv3 kernel trace of the headline command (its own run: never combined with PMC passes)
rocprofv3 --kernel-trace --stats -d $out/prof_fp32 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity > $out/prof_fp32.log 2>&1
db=$(find $out/prof_fp32 -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" $out/kernel_stats_fp32.md > /dev/null
rocprofv3 --kernel-trace --stats -d $out/prof_bf16 -- python bench.py --dtype bf16 --steps 5 --warmup 1 --no-cpu-baseline --no-parity > $out/prof_bf16.log 2>&1
db=$(find $out/prof_bf16 -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" $out/kernel_stats_bf16.md > /dev/null
# PMC passes (traffic, MFMA-busy, waits) of the same command
tools/pmc.sh ${tag}_fp32 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
tools/pmc.sh ${tag}_bf16 python bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
rm -rf $out/prof_fp32 $out/prof_bf16   # keep the summaries, not the traces
ls $out
