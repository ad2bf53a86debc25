#!/bin/bash
# quick iteration: cdna4 parity tests, GEMV sweep (defaults), bench.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_cdna4.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_cdna4.log
( timeout 300 python tools/gemv_sweep.py --defaults-only --m 1 4 8 2>&1 | grep -v "^/opt" | tail -60 ) > $O/gemv_sweep.log
( timeout 600 python bench.py 2>&1 | tail -3 ) > $O/bench.log
tail -4 $O/pytest_cdna4.log; cat $O/bench.log
