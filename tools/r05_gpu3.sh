#!/bin/bash
# round 5, GPU session 3: lean arithmetic in the wide skeleton (wide.hip r05c + timeline), attention waves per block incl. 3,
# quick regression of the library (fragments, ops, abi) with the argument prefetch / kbias plumbing
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WIDE_SUITE=r05c timeout 600 tools/floor/wide 32 > gpurun_out/r05_wide_c.log 2>&1
WIDE_SUITE=r05t WIDE_T3=1 WIDE_TRACE=1 timeout 300 tools/floor/wide 32 > gpurun_out/r05_wide_t3.log 2>&1
cat gpurun_out/r05_wide_c.log
( for nw in 2 3 4; do for ctx in 1024 1009; do
    echo -n "NW=$nw ctx=$ctx "; TGIS_ATTN_NW=$nw python -c "
import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'text-generation-inference_amd')
import microbench as mb
mb.bench_attn(32,32,32,128,$ctx,ns=1,sets=6)
"; done; done ) > gpurun_out/r05_attn_nw.log 2>&1
cat gpurun_out/r05_attn_nw.log
python -m pytest tests/test_fragments_gpu.py tests/test_ops_gpu.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r05_gpu3_tests.log
for kb in 0 1 2; do echo "== TGIS_GPTQ_WIDE_KBIAS=$kb"; TGIS_GPTQ_WIDE_KBIAS=$kb python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --blocks 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'], d['graph_ms_per_step'])"; done | tee gpurun_out/r05_kbias_bench.log
