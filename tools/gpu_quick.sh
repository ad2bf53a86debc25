#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_moe.py tests/test_gpu_cdna4.py tests/test_w3.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_quick.log
( timeout 300 python bench.py --no-cpu-baseline --no-prefill 2>&1 | tail -1 | cut -c1-900 ) > $O/bench.log
tail -4 $O/pytest_quick.log; cat $O/bench.log
