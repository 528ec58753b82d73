#!/bin/bash
# The first 8-GPU lease, made decisive (run on a node with >= 8 MI355X; 24 bench runs of 15-30 s: ~10 minutes):
#   weak scaling of the metric's workload (HuBERT-base, 32 x 10 s per GPU) in the exact mode (fp32) and the in-tolerance throughput
#   mode (fp16x2), strong scaling of cfg4 (WavLM-large, 256 x <= 15 s mixed over all GPUs) in fp16x2, at N = 1, 2, 4, 8, with BOTH
#   forms of the exchange (ring = one RCCL all-gather per state; direct = all-pairs send / receive, one peer per xGMI link) and the
#   two ways of issuing it (torch.distributed / the library's own s3enc_comm_* entry points).
# Every line is bench.py's JSON (value = whole-job frames/s, comm.exposed_ms_per_step = what the compute does not hide); the
# summary at the end is the scaling table (efficiency is computed here only for reading convenience — the driver computes its own).
# usage: tools/scale_curve.sh [tag] [max_gpus]
set -u
tag=${1:-r05}
maxn=${2:-8}
out=gpurun_out/${tag}_scale
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--no-cpu-baseline --no-other-modes --no-parity"
ns="1"; for n in 2 4 8; do [ $n -le $maxn ] && ns="$ns $n"; done

run() {  # name, N, bench flags...
  local name=$1 n=$2; shift 2
  timeout 300 python bench.py --gpus $n $Q "$@" > $out/$name.json 2> $out/$name.err || echo "FAILED: $name (see $out/$name.err)" >&2
}

for n in $ns; do
  if [ $n -eq 1 ]; then
    run weak_fp32_n1 1 --dtype fp32 --steps 40 --warmup 5
    run weak_fp16x2_n1 1 --dtype fp16x2 --steps 100 --warmup 5
    run strong_fp16x2_n1 1 --model wavlm_large --secs 15 --mixed --scaling strong --global-batch 256 --dtype fp16x2 --steps 3 --warmup 1
    continue
  fi
  for algo in ring direct; do
    run weak_fp32_n${n}_${algo} $n --dtype fp32 --steps 40 --warmup 5 --exchange-algo $algo
    run weak_fp16x2_n${n}_${algo} $n --dtype fp16x2 --steps 100 --warmup 5 --exchange-algo $algo
    run strong_fp16x2_n${n}_${algo} $n --model wavlm_large --secs 15 --mixed --scaling strong --global-batch 256 --dtype fp16x2 \
        --steps $((3 * n)) --warmup 2 --exchange-algo $algo
  done
  [ $n -eq $maxn ] || continue
  # at the full node only: the exchange behind the C ABI (what a non-Python binder runs), the featurized form (13x fewer bytes), none
  run weak_fp16x2_n${n}_direct_cabi $n --dtype fp16x2 --steps 100 --warmup 5 --exchange-algo direct --exchange-via cabi
  # round 6: the copy-engine form (S3ENC_EXCHANGE_COPY: IPC-mapped slabs, one hipMemcpyAsync per state and peer, no CU, no RCCL)
  run weak_fp16x2_n${n}_copy $n --dtype fp16x2 --steps 100 --warmup 5 --exchange-algo copy
  run weak_fp32_n${n}_copy $n --dtype fp32 --steps 40 --warmup 5 --exchange-algo copy
  run weak_fp16x2_n${n}_featurized $n --dtype fp16x2 --steps 100 --warmup 5 --gather featurized
  run weak_fp16x2_n${n}_none $n --dtype fp16x2 --steps 100 --warmup 5 --gather none
  # the CU side of the exchange against its one-GPU proxy (profiles/r05_cu_contention.md: +2 % for foreign workgroups, +20 % for a
  # smaller persistent grid): the same run with 16 CUs left out of the 16-bit GEMM's grid
  run weak_fp16x2_n${n}_direct_reserve16 $n --dtype fp16x2 --steps 100 --warmup 5 --exchange-algo direct --tune reserve_cus=16
done

python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows[os.path.basename(f)[:-5]] = d
base = {k[:-3]: v["value"] for k, v in rows.items() if k.endswith("_n1")}
print("| run | N | frames/s | ms/step | x N=1 | exposed comm ms | bytes in / GPU / step |")
print("|---|---:|---:|---:|---:|---:|---:|")
for k, d in rows.items():
    b = base.get("_".join(k.split("_")[:2]))
    c = d.get("comm") or {}
    print(f"| {k} | {d['n_gpus']} | {d['value']:.0f} | {d['ms_per_step']} | {d['value'] / b:.2f} | "
          f"{c.get('exposed_ms_per_step', '')} | {c.get('bytes_received_per_gpu_per_step', '')} |" if b else f"| {k} | {d['n_gpus']} | {d['value']:.0f} | {d['ms_per_step']} | | | |")
PY
