#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x --durations=25 2>&1 | tail -45 > gpurun_out/r05_gpu_tests_full.log
tail -45 gpurun_out/r05_gpu_tests_full.log
