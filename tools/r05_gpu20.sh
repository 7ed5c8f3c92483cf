#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/dense_frag_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dense_frag.log
