"""Per-kernel averages of a rocprofv3 --pmc pass (rocpd sqlite .db).  usage:
    python tools/rocpd_pmc.py fetch.db write.db out.json
FETCH_SIZE / WRITE_SIZE are in KiB.  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads -> doubled; WRITE_SIZE taken as is
(uncalibrated, and tiny for these kernels)."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select p.name, k.grid_x, k.workgroup_x, count(*), avg(p.counter_value), sum(p.counter_value) from pmc_events p "
        "join kernels k on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by p.name, k.grid_x, k.workgroup_x",
        (counter,)).fetchall()
    return {(r[0].split("(")[0], r[1], r[2]): dict(calls=r[3], avg=r[4], total=r[5]) for r in rows}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE") if len(sys.argv) > 2 and sys.argv[2] != "-" else {}
    out = []
    for key, f in sorted(fetch.items(), key=lambda kv: -kv[1]["total"]):
        if "awq::" not in key[0]:
            continue
        w = write.get(key, dict(avg=0.0))
        out.append(dict(kernel=key[0].replace("void ", ""), grid_x=key[1], block_x=key[2], calls=f["calls"],
                        fetch_size_kib_avg=round(f["avg"], 2), write_size_kib_avg=round(w["avg"], 2),
                        hbm_bytes_per_launch_corrected=int(2 * f["avg"] * 1024 + w["avg"] * 1024)))
    for o in out:
        print(o)
    if len(sys.argv) > 3:
        json.dump(dict(note="FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes; "
                            "separate --pmc passes of `bench.py --steps 3 --warmup 1 --no-graph`", kernels=out),
                  open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
