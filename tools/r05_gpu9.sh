#!/bin/bash
# round 5, GPU session 9: plan variants on TP = 8 shard shapes
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf /tmp/prof_$name
  env "${envs[@]}" rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python $REPO/tools/tp_segments_rccl1.py --steps 8 "$@" > /tmp/$name.log 2>&1
  echo "== $name: ${envs[*]} $*"; grep "one graph" /tmp/$name.log
  python $REPO/tools/kernel_breakdown.py /tmp/prof_$name 500 | grep -v "Cijk\|at::native\|rocclr\|norm_kernelIDF16_Lb1ELb0ELi256" | head -12 | cut -c1-190
}
( run c4_plan11 TGIS_GPTQ_WIDE_PLAN=1,1 -- --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048
  run c3_min64 TGIS_ROPE_MIN_BLOCKS=64 -- --tp 8
  run c3_min32 TGIS_ROPE_MIN_BLOCKS=32 -- --tp 8
  run c3tp4_min64 TGIS_ROPE_MIN_BLOCKS=64 -- --tp 4
  run c4_min32 TGIS_ROPE_MIN_BLOCKS=32 -- --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048
) > $REPO/gpurun_out/r05_tp8_variants.log 2>&1
cat $REPO/gpurun_out/r05_tp8_variants.log
