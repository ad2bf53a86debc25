#!/bin/bash
set -u
# needs the probe build: AWQ_PROBES=1 python -c 'from llm_awq_amd import build; build.build_lib(force=True)' before gpurun
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
for blocks in 2048 4096 8192; do
( timeout 300 python bench.py --layout v2 --unfused-mlp --no-prefill --no-cpu-baseline --tune gemv_probe=2 --tune gemv_probe_blocks=$blocks 2>&1 | tail -1 | cut -c1-1400 ) > $O/floor_linear_$blocks.log
done
( timeout 300 python bench.py --layout v2 --no-prefill --no-cpu-baseline --tune gemv_probe=2 --tune gemv_probe_blocks=8192 2>&1 | tail -1 | cut -c1-1400 ) > $O/floor_linear_fused.log
( timeout 300 python bench.py --layout v2 --no-prefill --no-cpu-baseline --tune gemv_probe=3 2>&1 | tail -1 | cut -c1-1400 ) > $O/floor_null.log
grep -o '"value": [0-9.]*\|"achieved": [0-9.]*\|"avg_launch_us": [0-9.]*' $O/floor_*.log
