#!/bin/bash
# round 5, GPU session 8: per-kernel breakdown of one rank's step at TP = 8 shard shapes (cfg4, cfg3)
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tp4 /tmp/prof_tp3
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tp4 -- python $REPO/tools/tp_segments_rccl1.py --steps 8 --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048 > /tmp/p4.log 2>&1
python $REPO/tools/kernel_breakdown.py /tmp/prof_tp4 500 > $REPO/gpurun_out/r05_tp8_cfg4_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tp3 -- python $REPO/tools/tp_segments_rccl1.py --steps 8 --tp 8 > /tmp/p3.log 2>&1
python $REPO/tools/kernel_breakdown.py /tmp/prof_tp3 500 > $REPO/gpurun_out/r05_tp8_cfg3_kernels.txt
head -20 $REPO/gpurun_out/r05_tp8_cfg4_kernels.txt | cut -c1-200; echo; head -20 $REPO/gpurun_out/r05_tp8_cfg3_kernels.txt | cut -c1-200
