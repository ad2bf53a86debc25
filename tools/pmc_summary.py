"""Summarise every rocpd .db under a directory into one text file (per kernel name + grid, per counter: average value per
counter instance and the kernel's average duration), then delete the .db files (gpurun_out/ is capped at 64 MiB).
usage: python tools/pmc_summary.py <dir> <out.txt> [name filter]"""
import glob
import os
import sqlite3
import sys

d, out = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else "awq::"
lines = []
for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = c.execute(
            "select p.name, k.grid_x, k.workgroup_x, p.counter_name, avg(p.counter_value), avg(k.end - k.start), count(*) "
            "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where p.name like ? "
            "group by p.name, k.grid_x, k.workgroup_x, p.counter_name", ("%" + flt + "%",)).fetchall()
    except Exception as e:  # noqa
        rows = []
        lines.append(f"# {db}: {e}")
    for r in rows:
        lines.append(f"{os.path.basename(os.path.dirname(db))} {r[0].split('(')[0].replace('void ', '')} grid={r[1]} block={r[2]} "
                     f"{r[3]} avg={r[4]:.1f} kernel_ns={r[5]:.0f} rows={r[6]}")
    c.close()
    os.remove(db)
open(out, "w").write("\n".join(lines) + "\n")
print(f"{len(lines)} lines -> {out}")
