#!/bin/bash
# round 5, GPU session 4: what bounds the wide GEMM's loop (ablation modes, wide.hip r05d) + the lean arithmetic re-checked
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WIDE_SUITE=r05d timeout 600 tools/floor/wide 32 > gpurun_out/r05_wide_d.log 2>&1
WIDE_SUITE=r05c timeout 600 tools/floor/wide 32 > gpurun_out/r05_wide_c.log 2>&1
cat gpurun_out/r05_wide_d.log gpurun_out/r05_wide_c.log | cut -c1-150
