#!/bin/bash
# round 5, GPU session 10: decode attention at the TP = 8 shard shapes (cfg4: B 64, 8 q heads on 1 kv head, ctx 2048; cfg3: B 32, 4 heads MHA,
# ctx 1024) over waves per block x key splits
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( for nw in 8 4 2; do for ns in 1 2 4 8; do
    echo -n "cfg4/tp8 NW=$nw "; TGIS_ATTN_NW=$nw python -c "
import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'text-generation-inference_amd')
import microbench as mb
mb.bench_attn(64,8,1,128,2048,ns=$ns,sets=8)
" 2>&1 | grep attn_decode; done; done
  for nw in 8 4 2; do for ns in 1 2 4; do
    echo -n "cfg3/tp8 NW=$nw "; TGIS_ATTN_NW=$nw python -c "
import sys
sys.path.insert(0,'tools'); sys.path.insert(0,'text-generation-inference_amd')
import microbench as mb
mb.bench_attn(32,4,4,128,1024,ns=$ns,sets=8)
" 2>&1 | grep attn_decode; done; done ) > gpurun_out/r05_attn_tp8.log 2>&1
cat gpurun_out/r05_attn_tp8.log
