#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WIDE_SUITE=r05c timeout 600 tools/floor/wide 32 > gpurun_out/r05_wide_c.log 2>&1
cat gpurun_out/r05_wide_c.log | cut -c1-150
