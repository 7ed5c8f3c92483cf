#!/bin/bash
# round 5, GPU session 2: weighted k split / prologue barrier / SiLU epilogue forms of the wide GEMM (tools/floor/wide.hip r05b)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WIDE_SUITE=r05b timeout 600 tools/floor/wide 32 > gpurun_out/r05_wide_b.log 2>&1
WIDE_SUITE=r05t WIDE_T2=1 WIDE_TRACE=1 timeout 300 tools/floor/wide 32 > gpurun_out/r05_wide_t2.log 2>&1
cat gpurun_out/r05_wide_b.log
