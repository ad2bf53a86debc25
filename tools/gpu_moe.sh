#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_moe.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -6 ) > $O/pytest_moe.log
cat $O/pytest_moe.log
( timeout 900 python tools/w3_moe_sweep.py 2>&1 | grep -v amdgpu.ids ) > $O/w3_moe_sweep.log
cat $O/w3_moe_sweep.log
