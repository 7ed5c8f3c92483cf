#!/bin/bash
# Counter passes over a short decode run of the headline workload (counters only, one set per pass), per-kernel averages of
# the decode launches -> gpurun_out/<tag>_pmc_step.md.   gpurun -- tools/pmc_step.sh r03
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_step
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
def short(n, grid):
    for k in ("attn_paged_kernel", "gptq_wide_kernel", "gptq_gemm_kernel", "norm_kernel", "dense_gemm_kernel", "argmax"):
        if k in n:
            if k in ("gptq_gemm_kernel", "gptq_wide_kernel"):
                t = n[n.index("<") + 1:n.index(">")].replace(" ", "")
                return f"{k}<{t}> grid {grid}"
            if k == "attn_paged_kernel" or k == "norm_kernel":
                return f"{k} grid {grid}"
            return k
    return None
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        g = r.get("Grid_Size", r.get("Grid_Size_X", ""))
        s = short(r["Kernel_Name"], g)
        if s:
            a = agg[s][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("$REPO/gpurun_out/${TAG}_pmc_step.md", "w") as fh:
    fh.write("# per-kernel counter averages over the launches of a short \`bench.py\` run (prefill launches of a kernel share its row when the grid matches; rocprofv3 --pmc, one counter set per pass)\n\n")
    for kname, cs in sorted(agg.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
        n = max(v[1] for v in cs.values())
        if n < 16:
            continue
        fh.write(f"## {kname}  ({n} dispatches)\n\n| counter | average per dispatch |\n|---|---|\n")
        for c, (v, k) in sorted(cs.items()):
            fh.write(f"| {c} | {v / k:.1f} |\n")
        d = {c: v / k for c, (v, k) in cs.items()}
        fh.write("\n")
        if d.get("SQ_WAVE_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            # both are sums over waves / SIMDs of the chip: the share of wave-resident time in which the wave's SIMD had an MFMA in flight
            fh.write(f"MFMA busy cycles / wave cycles: {d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_WAVE_CYCLES']:.3f};  ")
        if d.get("SQ_WAVE_CYCLES") and d.get("SQ_WAIT_ANY") is not None:
            fh.write(f"waiting / wave cycles: {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.3f};  ")
        if d.get("SQ_LDS_IDX_ACTIVE") and d.get("SQ_LDS_BANK_CONFLICT") is not None:
            fh.write(f"LDS bank-conflict cycles / LDS active cycles: {d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.3f};  ")
        if d.get("SQ_INSTS_VALU") and d.get("SQ_INSTS_MFMA"):
            fh.write(f"VALU : MFMA instructions = {d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA']:.1f} : 1")
        fh.write("\n\n")
PY
rm -rf $OUT
