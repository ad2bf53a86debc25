"""print the start / end of the last N kernels of a rocprofv3 kernel trace (rocpd .db), relative to the first of them (us)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select name, grid_x, start, end{', ' + qcol if qcol else ''} from kernels where name like '%gemv_dma%' order by start desc limit {n}").fetchall()[::-1]
t0 = rows[0][2]
for r in rows:
    print(f"{(r[2] - t0) / 1e3:9.2f} -> {(r[3] - t0) / 1e3:9.2f}  ({(r[3] - r[2]) / 1e3:7.2f} us)  grid {r[1]:>7}  q {r[4] if qcol else '-'}  {r[0][:50]}")
