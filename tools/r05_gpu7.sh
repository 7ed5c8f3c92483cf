#!/bin/bash
# round 5, GPU session 7: CT = 1 / unsplit "finished" plans of the wide GEMM on tensor-parallel shard shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_fragments_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r05_gpu7_tests.log
( for ct1 in 0 1; do
  echo "== TGIS_GPTQ_WIDE_CT1=$ct1  cfg4 TP=8 rank step"
  TGIS_GPTQ_WIDE_CT1=$ct1 timeout 900 python tools/tp_segments_rccl1.py --steps 16 --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048 2>&1 | grep "one graph"
  echo "== TGIS_GPTQ_WIDE_CT1=$ct1  cfg3 TP=8 / 4 / 2 rank step"
  for tp in 8 4 2; do TGIS_GPTQ_WIDE_CT1=$ct1 timeout 600 python tools/tp_segments_rccl1.py --steps 16 --tp $tp 2>&1 | grep "one graph"; done
done ) 2>&1 | tee gpurun_out/r05_tp_rank_steps.log
