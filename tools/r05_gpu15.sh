#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -x -k "rope or dense or cfg2" 2>&1 | tail -4 | tee gpurun_out/r05_gpu15_tests.log
python bench.py --config tinyllama-1.1b --steps 40 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cfg2', d['ms_per_step'], d['ms_per_step_blocks'], d['graph_ms_per_step'])" | tee gpurun_out/r05_cfg2_tn1.log
bash tools/profile_config.sh r05b_cfg2 --config tinyllama-1.1b > /dev/null 2>&1
head -14 gpurun_out/r05b_cfg2_kernel_stats.txt 2>/dev/null | cut -c1-200
ls gpurun_out | grep r05b
