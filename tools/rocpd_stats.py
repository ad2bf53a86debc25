"""Summarise a rocprofv3 result (rocpd sqlite .db from `rocprofv3 --kernel-trace --stats`) as a per-kernel
table: calls, avg / min / max duration (ns), total, share.  Kernels are split by grid size so the five
Llama layer shapes show up separately.  usage: python tools/rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, count(*), "
        "avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
        "group by name, grid_x, workgroup_x order by sum(end-start) desc").fetchall()
    total = sum(r[-1] for r in rows) or 1
    hdr = ["kernel", "grid_x", "block_x", "lds_bytes", "vgpr", "agpr", "sgpr", "calls", "avg_ns", "min_ns", "max_ns",
           "total_ns", "pct"]
    table = [[r[0]] + list(r[1:8]) + [round(r[8], 1), r[9], r[10], r[11], round(100.0 * r[11] / total, 2)] for r in rows]
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(hdr)
            w.writerows(table)
    for t in table[:40]:
        print(f"{t[0][:90]:90s} grid={t[1]:>8} blk={t[2]:>4} vgpr={t[4]:>3} calls={t[7]:>6} avg={t[8]:>10.1f}ns "
              f"min={t[9]:>8} max={t[10]:>8} {t[12]:5.1f}%")


if __name__ == "__main__":
    main()
