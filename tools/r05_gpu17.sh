#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fragments_gpu.py -q -m gpu -x -k "attn or attention or decode" 2>&1 | tail -3
python tools/fuzz_attention.py 120 5 2>&1 | tail -3
timeout 600 tools/floor/attn_unit > gpurun_out/r05_attn_timeline2.log 2>&1
grep -A9 "^cfg" gpurun_out/r05_attn_timeline2.log | grep -v "XCC\|medians\|by wave\|by block\|absolute"
