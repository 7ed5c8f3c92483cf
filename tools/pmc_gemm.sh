#!/bin/bash
# PMC passes over the GPTQ GEMM (one shape): tools/pmc_gemm.sh 4096x22016   (run on the GPU box; counters only)
set -e
cd /tmp && export TMPDIR=/tmp
SHAPE=${1:-4096x22016}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$SHAPE
mkdir -p $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_WAIT_IFETCH SQ_WAVES_LT_64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/tools/prof_gemm.py $SHAPE > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gptq_gemm_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f"{k:32s} {v / n:16.1f}  (avg over {n} dispatches)")
PY
