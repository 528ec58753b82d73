set -u
mkdir -p gpurun_out/r06
for e in 8 16 17 24 25; do QUICK=1 tools/micro/attn_lab_x$e > gpurun_out/r06/attn_lab_x$e.md 2>&1; done
QUICK=1 tools/micro/attn_lab > gpurun_out/r06/attn_lab_x0.md 2>&1
for e in 0 8 16 17 24 25; do echo "== x$e"; grep "product" gpurun_out/r06/attn_lab_x$e.md; grep -c "0 of" gpurun_out/r06/attn_lab_x$e.md; done
