#!/bin/bash
# The decode step AS THE BENCH TIMES IT -- one hipGraph replay per step -- under `rocprofv3 --kernel-trace --stats` (the eager profile of
# round 5 did not reproduce the line: 1.09x).  Only the M = 1 leg runs; the script prints the sum of the kernel averages per step next to the ms_per_step
# the same process reported.   usage (GPU box): tools/gpu_decode_graph_profile.sh [out dir]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=${1:-gpurun_out/dg}; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o dec -- python bench.py --steps 40 --warmup 5 --no-prefill --no-batched-decode --no-dropin --no-extra-configs --no-cpu-baseline 2>$O/bench_graph.err | grep '"metric"' | tail -1 ) > $O/bench_graph.json
python tools/rocpd_stats.py $O/prof/dec_results.db $O/decode_graph_kernel_stats.csv > $O/decode_graph_kernel_stats.txt
python - "$O" <<'PY'
import csv, json, sys
o = sys.argv[1]
line = json.load(open(o + "/bench_graph.json"))
rows = [r for r in csv.DictReader(open(o + "/decode_graph_kernel_stats.csv")) if "gemv_dma_kernel" in r["kernel"] or "gemv_cdna4_kernel" in r["kernel"] or "skinny_cdna4_kernel" in r["kernel"]]  # (round 6: qkv decodes on the skinny kernel)
layers = line["config"]["layers"]
# per_kernel_decode() replays each launch kind on its own as well (its graphs hold the same kernels): the averages below are over ALL launches of a kernel row
tot = sum(float(r["avg_ns"]) for r in rows) * layers * 1e-6
out = {"ms_per_step_reported": line["ms_per_step"], "sum_kernel_avg_ms_per_step": round(tot, 4), "ratio": round(tot / line["ms_per_step"], 4),
       "rows": [(r["kernel"][:60], int(r["grid_x"]), int(r["calls"]), float(r["avg_ns"])) for r in rows], "mode": "hipGraph replay under rocprofv3 --kernel-trace"}
print(json.dumps(out))
open(o + "/decode_graph_consistency.json", "w").write(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete
