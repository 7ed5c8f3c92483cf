#!/bin/bash
# NOTE (round 5): -DABL_NOSYNCWAIT lives in experiments/csrc/r04_ablations/gptq_gemm_body.h now (round-4 tree, commit 1892754).
# Upper bound of what a slacker x-chunk hand-off could gain: build with the group wait compiled out (wrong results) and
# time the cfg3 GEMMs against the product library.
#   (here)      cd text-generation-inference_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DABL_NOSYNCWAIT -o lib/abl_nosync.so csrc/*.hip
#   (GPU box)   bash tools/abl_nosync.sh
cd "$(dirname "$0")/.."
for l in libtgis_hip abl_nosync; do
  echo "== $l"
  TGIS_GPTQ_NOREDUCE=1 TGIS_HIP_LIB=$PWD/text-generation-inference_amd/lib/$l.so timeout 300 python tools/microbench.py 2>&1 | grep gptq_gemm
done
