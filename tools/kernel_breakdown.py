"""Per-kernel summary of a rocprofv3 --kernel-trace --output-format csv run: name x grid x workgroup -> calls, avg us, share.
    python tools/kernel_breakdown.py <dir with *_kernel_trace.csv> [min_calls]"""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:110], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""),
               r.get("Workgroup_Size_X", ""))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
total = sum(sum(v) for v in rows.values()) or 1
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < min_calls:
        continue
    print(f"{k[0]} grid {k[1]}x{k[2]}x{k[3]} wg {k[4]} calls {len(v)} avg {sum(v) / len(v) / 1e3:.1f} us  {100.0 * sum(v) / total:.1f}%")
