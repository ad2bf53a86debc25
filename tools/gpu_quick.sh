#!/bin/bash
# quick iteration: selected parity tests, bench.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_cdna4.py tests/test_w3.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_quick.log
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -2 ) > $O/bench.log
( timeout 300 python tools/gemvc_sweep.py 1 2>&1 | grep "waves= 0" ) > $O/gemvc_defaults.log
tail -4 $O/pytest_quick.log; cat $O/bench.log | cut -c1-1500; cat $O/gemvc_defaults.log
