set -u
cd $GRAFT_REPO_ROOT
export PMC_GROUPS="sq1 sq2"
tools/pmc.sh g16_v1 python tools/gemm_bench.py bf16 --variants 1 --shapes sq8k,fc1 --rounds 1 --reps 3 > /dev/null 2>&1
tools/pmc.sh g16_v7 python tools/gemm_bench.py bf16 --variants 7 --shapes sq8k,fc1 --rounds 1 --reps 3 > /dev/null 2>&1
tools/pmc.sh g16_v102 python tools/gemm_bench.py bf16 --variants 102 --shapes sq8k --rounds 1 --reps 3 > /dev/null 2>&1
tools/pmc.sh x3_v1 python tools/gemm_bench.py fp32x3 --variants 1 --shapes sq4k --rounds 1 --reps 3 > /dev/null 2>&1
for t in g16_v1 g16_v7 g16_v102 x3_v1; do echo "== $t"; grep -i "gemm" gpurun_out/pmc_$t.md | head -4; done
