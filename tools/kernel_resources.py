#!/usr/bin/env python3
"""Register / LDS / spill summary of every kernel of one HIP source (the compiler's view; no GPU needed).
usage: tools/kernel_resources.py s3prl_amd/csrc/gemm16.hip [extra hipcc flags]"""
import os, re, subprocess, sys

src, extra = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + os.path.join(root, "s3prl_amd", "csrc"),
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
keys = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key in keys:
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key] = m.group(1)
if not rows:
    print(out[-3000:])
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"s3::\(anonymous namespace\)::|s3::", "", name).replace("void ", "")[:100]
    g = lambda k: r.get(k, "?")
    print(f"{name:100s} vgpr {g('VGPRs'):>3} agpr {g('AGPRs'):>3} sgpr {g('TotalSGPRs'):>3} vspill {g('VGPRs Spill'):>3} "
          f"sspill {g('SGPRs Spill'):>3} scratch {g('ScratchSize [bytes/lane]'):>4} occ {g('Occupancy [waves/SIMD]')}")
