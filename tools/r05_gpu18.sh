#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -3
python tools/fuzz_attention.py 160 7 2>&1 | tail -2
timeout 600 tools/floor/attn_unit > gpurun_out/r05_attn_timeline3.log 2>&1
grep -A9 "^cfg" gpurun_out/r05_attn_timeline3.log | grep -v "XCC\|medians\|by wave\|by block\|absolute" | grep "^cfg\|first table\|stored\|pages done"
for c in llama2-7b-gptq starcoder-15b; do python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['ms_per_step_blocks'], d['graph_ms_per_step'])"; done
