#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_cdna4.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_quick.log
( timeout 300 python tools/skinny_sweep.py 2>&1 | grep -v "^/opt" ) > $O/skinny_sweep.log
tail -4 $O/pytest_quick.log; cat $O/skinny_sweep.log
