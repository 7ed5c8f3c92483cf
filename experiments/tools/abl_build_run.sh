#!/bin/bash
# NOTE (round 5): the -DABL_* switches live in experiments/csrc/r04_ablations/ now; this script documents how the round-4
# tree (commit 1892754) was built and timed for profiles/r0[1-3]_*ablations*.log.
# Build ablation variants of libtgis_hip.so (-DABL_*) here, then time the GEMM shapes on the GPU box:
#   tools/abl_build_run.sh build      (CPU container)
#   tools/abl_build_run.sh run        (GPU box)
cd "$(dirname "$0")/.."
VARIANTS="NOSTAGE NODEQ NOMFMA NOLDSREAD NODEQ,NOMFMA NOSTAGE,NODEQ,NOMFMA"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    flags=$(echo $v | sed 's/,/ -DABL_/g; s/^/-DABL_/')
    (cd text-generation-inference_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -o lib/abl_$v.so csrc/*.hip) &
  done
  wait
  ls -la text-generation-inference_amd/lib/
else
  for v in "" $VARIANTS; do
    if [ -n "$v" ]; then export TGIS_HIP_LIB=$PWD/text-generation-inference_amd/lib/abl_$v.so; fi
    echo "== ${v:-baseline}"
    python tools/microbench.py 2>&1 | grep gptq
  done
fi
