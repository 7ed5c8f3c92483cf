#!/bin/bash
# round 5, GPU session 1: kernel-argument placement probe + the wide GEMM ramp/drain experiments (tools/floor/wide.hip r05a/r05t)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( WIDE_SUITE=r05a timeout 600 tools/floor/wide 32 > gpurun_out/r05_wide_a.log 2>&1
  WIDE_SUITE=r05t WIDE_TRACE=1 timeout 300 tools/floor/wide 32 > gpurun_out/r05_wide_t.log 2>&1 )
bash tools/r05_env_probe.sh > /dev/null 2>&1
tail -5 gpurun_out/r05_env_probe.log
grep -c us gpurun_out/r05_wide_a.log
