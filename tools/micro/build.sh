#!/bin/bash
# Build the standalone measurement binaries of tools/micro (cross-compiles for gfx950 without a GPU; run from the repo root, after
# `python -c "import __graft_entry__ as g; g.build()"` — gemm32_lab links libs3enc.so).  The binaries are git-ignored but travel to
# the GPU box with the snapshot; tools/round_profiles.sh runs them.
set -e
cd "$(dirname "$0")/../.."
H="hipcc -O3 -std=c++17 --offload-arch=gfx950"
$H -DS3_ATTN_PROBE -Is3prl_amd/csrc tools/micro/attn_lab.hip -o tools/micro/attn_lab                    # (the timing probes compiled in: +20 registers)
$H -Is3prl_amd/csrc tools/micro/attn_lab.hip -o tools/micro/attn_lab_product                              # the library's code: QUICK=1 SKIP_F32=1
$H -DS3_ATTN_EXP=0 -Is3prl_amd/csrc tools/micro/attn_lab.hip -o tools/micro/attn_lab_x0                   # (the kernels before the second session of round 6)
$H -DS3_GEMM_PROBE -DS3_GEMM_PP_LAB -Is3prl_amd/csrc tools/micro/gemm16_lab.hip -o tools/micro/gemm16_lab
$H tools/micro/gemm16_loop_probe.hip -o tools/micro/gemm16_loop_probe
$H tools/micro/gemm_loop_probe.hip -o tools/micro/gemm_loop_probe
$H tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
$H tools/micro/mx_probe.hip -o tools/micro/mx_probe
$H tools/micro/mx_gemm_lab.hip -o tools/micro/mx_gemm_lab
hipcc -O2 --offload-arch=gfx950 tools/micro/gemm32_lab.cpp -Iinclude -Ls3prl_amd -ls3enc -Wl,-rpath,'$ORIGIN/../../s3prl_amd' -o tools/micro/gemm32_lab
hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/micro/lds_occupy.hip -o tools/micro/liblds_occupy.so    # tools/lds_base_probe.py
$H tools/micro/stream_order_probe.hip -o tools/micro/stream_order_probe
ls -la tools/micro | grep -v "\.hip\|\.cpp\|\.sh"
