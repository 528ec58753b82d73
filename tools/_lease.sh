set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
L=gpurun_out/r06d/runtime_flags.log
: > $L
cnt() { grep "concurrent\|second run" | python -c "
import sys, json
print([json.loads(l)['differing (layer, utterance) pairs'] for l in sys.stdin])"; }
probe() { timeout 600 python tools/two_stream_probe.py --dtype bf16 --splits 1 4 8 --steps 5 --diagnose --tune forward_chain=0 2>&1 | cnt; }
for e in X=1 DEBUG_CLR_SKIP_RELEASE_SCOPE=0 DEBUG_CLR_SKIP_RELEASE_SCOPE=1 DEBUG_HIP_DYNAMIC_QUEUES=0 DEBUG_HIP_DYNAMIC_QUEUES=1 AMD_DIRECT_DISPATCH=0 DEBUG_HIP_FORCE_ASYNC_QUEUE=1 ROC_SYSTEM_SCOPE_SIGNAL=0 DEBUG_CLR_MAX_BATCH_SIZE=1 GPU_STREAMOPS_CP_WAIT=0 HIP_FORCE_QUEUE_PROFILING=1 X=2; do
  for rep in 1 2; do
  echo "== env $e" | tee -a $L
  env $e bash -c "$(declare -f cnt probe); probe" 2>&1 | tee -a $L
  done
done
