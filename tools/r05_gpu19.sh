#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_fragments_gpu.py -q -m gpu -x -k "dense" 2>&1 | tail -15
python -m pytest tests/test_fullwidth_gpu.py tests/test_fulldepth_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "cfg2 or tinyllama or None" 2>&1 | tail -5
for v in 1 0; do
echo "== cfg2 TGIS_DENSE_FRAGMENTS=$v"
TGIS_DENSE_FRAGMENTS=$v python bench.py --config tinyllama-1.1b --steps 40 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cfg2', d['ms_per_step'], d['ms_per_step_blocks'], d['graph_ms_per_step'], d['config'].get('logits_finite'))"
done
bash tools/profile_config.sh r05c_cfg2 --config tinyllama-1.1b > /dev/null 2>&1
grep -v "Cijk\|at::native\|prefill\|prepare\|act_mul\|Lb0ELi256" gpurun_out/r05c_cfg2_kernel_stats.txt | head -12 | cut -c1-200
