#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_cdna4.py -m gpu -x -q -k "knobs or fused" 2>&1 | tail -5 ) > $O/pytest_cdna4.log
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench2 -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --prefill-iters 1 2>&1 | tail -3 ) > $O/rocprof_bench.log
( timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ubench -o ub -- tools/ubench/gemv_ubench 1 2>&1 | tail -3 ) > $O/rocprof_ubench.log
tail -3 $O/pytest_cdna4.log
