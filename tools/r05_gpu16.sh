#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 tools/floor/attn_unit > gpurun_out/r05_attn_timeline.log 2>&1
cat gpurun_out/r05_attn_timeline.log
