#!/bin/bash
# full GPU parity suite + smoke + bench (no profiling)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log
cat $O/smoke.log
( timeout 300 tools/ubench/gemm_ubench 103 2>&1 | cut -c1-100 ) > $O/gemm_ubench.log
cat $O/gemm_ubench.log
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -E "metric|Error|error|Traceback" | tail -3 ) > $O/bench.log
cut -c1-1800 $O/bench.log
