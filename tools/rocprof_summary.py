"""Summarise a rocprofv3 results .db (rocpd sqlite) into a per-kernel CSV: calls, total/avg/min/max ns, % time.
usage: python tools/rocprof_summary.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    con = sqlite3.connect(db)
    rows = list(con.execute(
        "select name || ' grid=' || grid_x || 'x' || grid_y || 'x' || grid_z || ' wg=' || workgroup_x, count(*), "
        "sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        "group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(end-start) desc"))
    total = sum(r[2] for r in rows) or 1
    hdr = ["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"]
    lines = [[r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / total, 2)] for r in rows]
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerow(hdr)
    w.writerows(lines)


if __name__ == "__main__":
    main()
