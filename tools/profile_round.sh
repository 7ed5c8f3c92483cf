#!/bin/bash
# Round profile on the GPU box (gpurun -- tools/profile_round.sh r01):
#   1. rocprofv3 --kernel-trace --stats over a short bench.py run  -> gpurun_out/<tag>_kernel_stats.csv
#   2. two counter passes (FETCH_SIZE, WRITE_SIZE; counters only)  -> gpurun_out/<tag>_attn_traffic.json, <tag>_gemm_traffic.json
# Copy both into profiles/ afterwards (gpurun_out/ is scratch).
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1 || tail -3 $OUT/prof_stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- $CMD > $OUT/prof_fetch.log 2>&1 || tail -3 $OUT/prof_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- $CMD > $OUT/prof_write.log 2>&1 || tail -3 $OUT/prof_write.log
python - <<PY
import csv, glob, json, collections, os
out = "$OUT"; tag = "$TAG"
# 1. per-kernel stats from the kernel trace (grouped by name + grid so that GEMM shapes stay apart)
rows = collections.defaultdict(list)
for f in glob.glob(out + "/prof_stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in rows.values()) or 1
with open(f"{out}/{tag}_kernel_stats.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid", "workgroup", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"])
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k[0][:160], "x".join(x for x in k[1:4] if x), k[4], len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / total, 2)])
# 2. HBM traffic of the attention kernel, per launch
def counter(dirname, name):
    vals = []
    for f in glob.glob(out + f"/{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_paged_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name and int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) >= 0:
                vals.append(float(r["Counter_Value"]))
    return vals
fetch, write = counter("prof_fetch", "FETCH_SIZE"), counter("prof_write", "WRITE_SIZE")
# decode launches only: the prefill launch of each pass is the largest-grid outlier, decode launches dominate the count
def decode_avg(v):
    if not v: return None, 0
    v = sorted(v); med = v[len(v) // 2]
    d = [x for x in v if 0.5 * med <= x <= 1.5 * med]
    return sum(d) / len(d), len(d)
fa, fn = decode_avg(fetch); wa, wn = decode_avg(write)
res = {"kernel": "attn_paged_kernel (decode)", "command": "$CMD".replace("$REPO/", ""),
       "FETCH_SIZE_KB_per_launch_raw": fa, "WRITE_SIZE_KB_per_launch_raw": wa, "launches": fn,
       # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes for 16 B/lane streaming reads
       "fetch_bytes_per_launch_corrected": None if fa is None else fa * 1024 * 2,
       "write_bytes_per_launch_uncalibrated": None if wa is None else wa * 1024}
json.dump(res, open(f"{out}/{tag}_attn_traffic.json", "w"), indent=1)
print(json.dumps(res))
# 3. the same for the int4 streaming GEMM: average over all of its decode launches (the four shapes of a layer)
def gemm_counter(dirname, name):
    vals = []
    for f in glob.glob(out + f"/{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("gptq_gemm_kernel<" in r["Kernel_Name"] or "gptq_wide_kernel<" in r["Kernel_Name"]) and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    return vals
gf, gw = gemm_counter("prof_fetch", "FETCH_SIZE"), gemm_counter("prof_write", "WRITE_SIZE")
if gf:
    resg = {"kernel": "gptq_wide_kernel / gptq_gemm_kernel (all decode launches: qkv + rotary, o_proj, gate_up + SiLU, down)", "launches": len(gf),
            "FETCH_SIZE_KB_per_launch_raw": sum(gf) / len(gf),
            "WRITE_SIZE_KB_per_launch_raw": (sum(gw) / len(gw)) if gw else None,
            "fetch_bytes_per_launch_corrected": sum(gf) / len(gf) * 1024 * 2,
            "write_bytes_per_launch_uncalibrated": (sum(gw) / len(gw) * 1024) if gw else None}
    json.dump(resg, open(f"{out}/{tag}_gemm_traffic.json", "w"), indent=1)
    print(json.dumps(resg))
PY
head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-220
