#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd $REPO
( for pl in none 2,2 4,4 4,2 2,4 3,3; do
  echo "== cfg4 TP=8 rank step, TGIS_GPTQ_WIDE_SILU_PLAN=$pl"
  if [ $pl = none ]; then unset TGIS_GPTQ_WIDE_SILU_PLAN; else export TGIS_GPTQ_WIDE_SILU_PLAN=$pl; fi
  timeout 900 python tools/tp_segments_rccl1.py --steps 16 --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048 2>&1 | grep "one graph"
done ) 2>&1 | tee gpurun_out/r05_tp8_silu_plans.log
