#!/bin/bash
# round 5, GPU session 6: library with the 8-byte k-part exchange, argument prefetch (wide GEMM, attention), hoisted block-table
# read: regression tests of the touched kernels, then bench + per-kernel profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_fragments_gpu.py tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r05_gpu6_tests.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r05a_bench_cfg3.json
python -c "
import json; d=json.load(open('gpurun_out/r05a_bench_cfg3.json')); print(d['ms_per_step'], d['ms_per_step_blocks'], d['roofline']['avg_launch_us'], d['roofline_gemm']['avg_launch_us'])"
bash tools/profile_round.sh r05a > /dev/null 2>&1
head -14 gpurun_out/r05a_kernel_stats.csv | cut -c1-200
