#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_fullwidth_gpu.py::test_cfg4_tp8_shard_shapes_match_oracle tests/test_fullwidth_gpu.py::test_cfg5_tp4_shard_shapes_match_oracle "tests/test_fulldepth_gpu.py" -q -m gpu -x --durations=5 2>&1 | tail -12 | tee gpurun_out/r05_gpu14_tests.log
for v in 128 32; do
  echo "== cfg2 TGIS_ROPE_MIN_BLOCKS=$v"
  TGIS_ROPE_MIN_BLOCKS=$v python bench.py --config tinyllama-1.1b --steps 40 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'], d['graph_ms_per_step'])"
done | tee gpurun_out/r05_cfg2_rope.log
