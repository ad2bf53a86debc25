#!/bin/bash
# the driver's GPU parity command, with the slowest tests listed
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=30 2>&1 | grep -v amdgpu.ids | tail -45 ) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
