#!/bin/bash
# decode-path variants of bench.py, back to back on one box (no prefill / cpu legs): which kernel / stacking / side buffer wins
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02e; mkdir -p $O
run() { tag=$1; shift; ( timeout 300 python bench.py --no-prefill --no-cpu-baseline --no-dropin "$@" 2>&1 | tail -1 ) > $O/bench_$tag.json; python - "$O/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {d['value']:8.1f} tok/s  frac {d['roofline']['frac']:.4f}  {d['roofline']['avg_launch_us']:.3f} us/launch  {d['config']['launches_per_token']} launches")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
}
run dma_half_interleaved
run dma_half_stacked --mlp stacked
run dma_packed_interleaved --sz packed
run ring_stacked --mlp stacked --sz packed --tune gemv_dma=0
run dma_half_interleaved_2
run ring_stacked_2 --mlp stacked --sz packed --tune gemv_dma=0
