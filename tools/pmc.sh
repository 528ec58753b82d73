#!/bin/bash
# Collect rocprofv3 PMC counters for a command, one pass per counter group (counter passes are never combined with
# sys/runtime tracing; see MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: tools/pmc.sh <tag> <command ...>
# Writes gpurun_out/pmc_<tag>/<group>/... and a summary gpurun_out/pmc_<tag>.md
set -u
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
declare -A groups=(
  [sq1]="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
  [sq2]="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
  [tcc]="TCC_HIT_sum TCC_MISS_sum"
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE"
)
for g in ${PMC_GROUPS:-sq1 sq2 tcc fetch write}; do
  rocprofv3 --pmc ${groups[$g]} --kernel-trace --output-format csv -d "$out/$g" -- "$@" > "$out/$g.log" 2>&1 || echo "group $g failed (see $out/$g.log)"
done
python "$root/tools/pmc_summary.py" "$out" > "$root/gpurun_out/pmc_$tag.md" 2> "$out/summary.err"
cat "$root/gpurun_out/pmc_$tag.md"
