#!/bin/bash
# NOTE (round 5): needs the ablation builds of the round-4 tree (experiments/csrc/r04_ablations/, commit 1892754).
cd "$(dirname "$0")/.."
code='
import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
mb.bench_gptq(32, int(sys.argv[1]), int(sys.argv[2]))
'
export TGIS_GPTQ_NOREDUCE=1
for v in "" NOSTAGE "NOSTAGE,NODEQ,NOMFMA"; do
  if [ -n "$v" ]; then export TGIS_HIP_LIB=$PWD/text-generation-inference_amd/lib/abl_$v.so; else unset TGIS_HIP_LIB; fi
  for rc in 0 1; do
    if [ $rc = 1 ]; then export TGIS_GPTQ_RC1=1; else unset TGIS_GPTQ_RC1; fi
    echo "== ${v:-baseline} RC1=$rc"
    TGIS_GPTQ_PLAN=4096,1,4,3 python -c "$code" 4096 22016 2>&1 | grep gptq
    TGIS_GPTQ_PLAN=2048,2,4,3 python -c "$code" 4096 12288 2>&1 | grep gptq
  done
done
