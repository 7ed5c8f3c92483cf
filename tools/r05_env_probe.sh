#!/bin/bash
# round 5: does the placement of kernel arguments (HIP_FORCE_DEV_KERNARG) move the captured decode step?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in unset 0 1; do
  if [ "$v" = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> gpurun_out/r05_env_probe.log
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 >> gpurun_out/r05_env_probe.log
done
unset HIP_FORCE_DEV_KERNARG
python -m pytest tests/test_fragments_gpu.py -q -m gpu -x -k "row_class" 2>&1 | tail -3 >> gpurun_out/r05_env_probe.log
cat gpurun_out/r05_env_probe.log
