#!/bin/bash
# Per-kernel breakdown of one bench.py config on the GPU box: tools/profile_config.sh <tag> <bench.py args...>
#   rocprofv3 --kernel-trace of a short run -> gpurun_out/<tag>_kernel_stats.txt (grouped by kernel name + grid)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$TAG -- python $REPO/bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof_$TAG.log 2>&1 || tail -3 $OUT/prof_$TAG.log
python - <<PY
import csv, glob, collections, re
rows = collections.defaultdict(list)
for f in glob.glob("$OUT/prof_$TAG/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in rows.values()) or 1
with open("$OUT/${TAG}_kernel_stats.txt", "w") as fh:
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:28]:
        name = re.sub(r"\(anonymous namespace\)::", "", k[0])[:120]
        line = f"{name} grid {k[1]}x{k[2]}x{k[3]} wg {k[4]} calls {len(v)} avg {sum(v)/len(v)/1000:.1f} us  {100.0*sum(v)/total:.1f}%"
        fh.write(line + "\n"); print(line)
PY
