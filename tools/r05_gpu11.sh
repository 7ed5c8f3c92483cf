#!/bin/bash
# round 5, GPU session 11: TP shard shapes after CT = 1 / unsplit finished plans / split SiLU / attention split rule
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd $REPO
python -m pytest tests/test_fragments_gpu.py tests/test_tp_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r05_gpu11_tests.log
( echo "== cfg4 TP=8 rank step"; timeout 900 python tools/tp_segments_rccl1.py --steps 16 --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048 2>&1 | grep "one graph"
  echo "== cfg4 TP=8 rank step, TGIS_GPTQ_WIDE_SILU_SPLIT=0"; TGIS_GPTQ_WIDE_SILU_SPLIT=0 timeout 900 python tools/tp_segments_rccl1.py --steps 16 --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048 2>&1 | grep "one graph"
  echo "== cfg3 TP=8 / 4 / 2 rank step"
  for tp in 8 4 2; do timeout 600 python tools/tp_segments_rccl1.py --steps 16 --tp $tp 2>&1 | grep "one graph"; done ) 2>&1 | tee gpurun_out/r05_tp_rank_steps2.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tp4
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tp4 -- python $REPO/tools/tp_segments_rccl1.py --steps 8 --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048 > /tmp/p4.log 2>&1
python $REPO/tools/kernel_breakdown.py /tmp/prof_tp4 500 | grep -v "Cijk\|at::native\|rocclr\|norm_kernelIDF16_Lb1ELb0ELi256" | head -12 | cut -c1-190 | tee $REPO/gpurun_out/r05_tp8_cfg4_kernels2.txt
