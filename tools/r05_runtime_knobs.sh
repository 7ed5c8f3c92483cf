#!/bin/bash
# round 5: do the HIP runtime's own switches for fences / graph packets / queues move the captured decode step (cfg3)?
# Each line: the variable, then bench.py's ms_per_step of the three timed blocks.  -> gpurun_out/r05_runtime_knobs.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LOG=gpurun_out/r05_runtime_knobs.log
: > $LOG
run() {
  echo "== $*" >> $LOG
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 |
    python -c 'import sys,json
l=sys.stdin.readline()
try:
    d=json.loads(l); print("   ms_per_step", d["ms_per_step"], d["ms_per_step_blocks"])
except Exception as e:
    print("   FAILED:", l[:300])' >> $LOG
}
run X_BASELINE=1
run AMD_OPT_FLUSH=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=16
run DEBUG_HIP_GRAPH_BATCH_SIZE=4096
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run GPU_MAX_HW_QUEUES=1
run AMD_DIRECT_DISPATCH=0
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run X_BASELINE=2
cat $LOG
