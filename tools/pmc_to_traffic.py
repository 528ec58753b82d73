#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh (run on `bench.py`) into the record bench.py quotes as
roofline.traffic: HBM-side bytes per launch of the GEMM kernel = (2 x FETCH_SIZE + WRITE_SIZE) KiB, averaged over
all its launches in the profiled run (the x2: gfx950 tallies 128-byte read requests at 64 B, MI355X_MICROARCH.md §HBM;
Infinity-Cache hits are included in FETCH_SIZE, so this is an upper bound on DRAM bytes).
usage: pmc_to_traffic.py <pmc dir> <model> <dtype> <batch> <secs> <out.json> [kernel-substring]"""
import csv, glob, json, os, sys

root, model, dtype, batch, secs, out = sys.argv[1:7]
sub = sys.argv[7] if len(sys.argv) > 7 else "gemm"
tot = {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]}
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in tot and sub in r["Kernel_Name"] and "s3::" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]
            t[0] += 1
            t[1] += float(r["Counter_Value"])
fetch = 2.0 * 1024.0 * tot["FETCH_SIZE"][1] / max(1, tot["FETCH_SIZE"][0])
write = 1024.0 * tot["WRITE_SIZE"][1] / max(1, tot["WRITE_SIZE"][0])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # csrc_sha16: the identity of the kernels these counters were measured on (bench.py refuses a stale record)

rec = {"model": model, "dtype": dtype, "batch": int(batch), "secs": float(secs), "csrc_sha16": bench.csrc_sha16(), "device_code_md5": bench.device_code_md5(),
       "gemm_bytes_per_launch": round(fetch + write), "fetch_bytes_per_launch": round(fetch),
       "write_bytes_per_launch": round(write), "launches_profiled": tot["FETCH_SIZE"][0],
       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --model {model} --dtype {dtype}` "
                 f"(tools/pmc.sh; summary in profiles/), FETCH_SIZE doubled per MI355X_MICROARCH.md, includes Infinity-Cache hits"}
recs = []
if os.path.exists(out):
    recs = [r for r in json.load(open(out)) if (r["model"], r["dtype"], r["batch"], r["secs"]) != (model, dtype, int(batch), float(secs))]
recs.append(rec)
json.dump(recs, open(out, "w"), indent=1)
print(json.dumps(rec))
